"""ggx_ref64.py — a float64 numpy restatement of the reference's BSDF leaf math, written from the HLSL text alone
(/root/reference/Pathtracer/include/GGX_v6.hlsl:1-224, BRDF_v6.hlsl:7-70, Lambertian_v6.hlsl:2-64, Common_v6.hlsl:1-3),
NOT from oracle/rt_oracle.c: it is the independent witness the oracle's and the kernels' leaf math is pinned
against in tests/test_ggx_pins.py.  Straight-line float64 with numpy's own sqrt / cos / sin, no fused operations, the
reference's constants: PI = 3.1415f (as the float32 value), EPSILON = 1e-6f.

Material record = the reference's MaterialOptimized (Common_v6.hlsl:62-74): Kd, Ks, roughness (Pr), metallic (Pm) rounded to
binary16 (DXC -enable-16bit-types: `half` is a true 16-bit float), and the full-precision float LUT[16] of `materials[mID]`.
"""
import numpy as np

PI = float(np.float32(3.1415))            # Common_v6.hlsl:1
EPSILON = float(np.float32(0.000001))     # Common_v6.hlsl:3
LUT_SIZE_THETA = 16


def half(x):
    """binary16 round trip of MaterialOptimized's members"""
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float64)


class Mat:
    def __init__(self, Kd, Ks, roughness, metallic, lut16):
        self.Kd = half(Kd); self.Ks = half(Ks)
        self.Pr = float(half(roughness)); self.Pm = float(half(metallic))
        self.LUT = np.asarray(lut16, np.float32).astype(np.float64)


def normalize(v):
    v = np.asarray(v, np.float64)
    return v / np.sqrt((v * v).sum(-1, keepdims=True))


def dot(a, b):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64)).sum(-1)


def saturate(x):
    return np.clip(x, 0.0, 1.0)


def ess_lut(m, NdotV):                                         # GGX_v6.hlsl:1-23
    NdotV = saturate(NdotV)
    f = NdotV * (LUT_SIZE_THETA - 1)
    i0 = np.floor(f).astype(int)
    i1 = np.minimum(i0 + 1, LUT_SIZE_THETA - 1)
    w = f - i0
    v0, v1 = m.LUT[i0], m.LUT[i1]
    return v0 + w * (v1 - v0)                                  # lerp(v0, v1, w)


def schlick(F0, cosTheta):                                     # GGX_v6.hlsl:26-29
    c = np.asarray(cosTheta, np.float64)[..., None]
    return saturate(F0 + (1.0 - F0) * np.abs(1.0 - c) ** 5)


def d_ggx(NdotH, roughness):                                   # GGX_v6.hlsl:31-40
    alpha = roughness * roughness
    a2 = alpha * alpha
    den = NdotH * NdotH * (a2 - 1.0) + 1.0
    return a2 / (PI * den * den)


def g2_smith(NdotV, NdotL, alpha):                             # GGX_v6.hlsl:43-52
    a2 = alpha * alpha
    dA = NdotV * np.sqrt(a2 + (1.0 - a2) * NdotL * NdotL)
    dB = NdotL * np.sqrt(a2 + (1.0 - a2) * NdotV * NdotV)
    return 2.0 * NdotL * NdotV / (dA + dB)


def g1_smith(NdotV, alpha):                                    # GGX_v6.hlsl:55-61
    a2 = alpha * alpha
    return 2.0 * NdotV / (np.sqrt(a2 + (1.0 - a2) * NdotV * NdotV) + NdotV)


def ggx_eval(m, normal, L, V):                                 # EvaluateBRDF_GGX, GGX_v6.hlsl:174-206 (L = -incoming, V = outgoing)
    N, V, L = normalize(normal), normalize(V), normalize(L)
    H = normalize(V + L)
    NdotV, NdotL, NdotH, VdotH = dot(N, V), dot(N, L), dot(N, H), dot(V, H)
    F = schlick(m.Ks, VdotH)
    D = d_ggx(NdotH, m.Pr)
    G = g2_smith(NdotV, NdotL, m.Pr * m.Pr)
    den = 4.0 * NdotV * NdotL
    with np.errstate(divide="ignore", invalid="ignore"):
        spec = F * (D * G / den)[..., None]
        Ess = ess_lut(m, NdotV)
        kms = (1.0 - Ess) / Ess
        out = spec * (1.0 + m.Ks * np.asarray(kms)[..., None])
    bad = (den < EPSILON) | ~np.isfinite(out).all(-1)
    return np.where(np.asarray(bad)[..., None], 0.0, out)


def ggx_pdf(m, normal, L, V):                                  # BRDF_PDF_GGX, GGX_v6.hlsl:209-224
    N, V, L = normalize(normal), normalize(V), normalize(L)
    H = normalize(V + L)
    NdotH, NdotV = dot(N, H), dot(N, V)
    alpha = m.Pr * m.Pr
    return g1_smith(NdotV, alpha) * d_ggx(NdotH, m.Pr) / (NdotV * 4.0)


def strategy_probs(m, outgoing, normal):                       # CalculateStrategyProbabilities, BRDF_v6.hlsl:50-70 -> (p_d, p_s)
    fr = schlick(m.Ks, dot(normal, outgoing))
    p_s = np.minimum(1.0, fr.sum(-1) / 3.0 + m.Pm)
    return 1.0 - p_s, p_s


def select_strategy(m, outgoing, normal, r):                   # SelectSamplingStrategy, BRDF_v6.hlsl:7-48 (r = the random number it draws)
    _, p_s = strategy_probs(m, outgoing, normal)
    return np.where(r <= p_s, 0 if m.Pr < 0.04 else 1, 0)


def lambert_eval(m):                                           # EvaluateBRDF_Lambertian, Lambertian_v6.hlsl:54-58
    return m.Kd[:3] / PI


def lambert_pdf(normal, L):                                    # BRDF_PDF_Lambertian, Lambertian_v6.hlsl:61-64
    return np.maximum(dot(normal, L), EPSILON) / PI


def mixture(m, normal, L, V):
    """F = p_d f_lambert + p_s f_ggx, P = p_d pdf_lambert + p_s pdf_ggx (Sampler_v6.hlsl:443-457, 600-612) -> F, P, p_d, p_s"""
    pd, ps = strategy_probs(m, V, normal)
    F = np.asarray(pd)[..., None] * lambert_eval(m) + np.asarray(ps)[..., None] * ggx_eval(m, normal, L, V)
    P = pd * lambert_pdf(normal, L) + ps * ggx_pdf(m, normal, L, V)
    return F, P, pd, ps


def coordinate_system(N):                                      # GGX_v6.hlsl:65-76
    N = np.asarray(N, np.float64)
    T = normalize(np.cross([0.0, 0.0, 1.0], N)) if abs(N[2]) < float(np.float32(0.999)) else normalize(np.cross([1.0, 0.0, 0.0], N))
    return T, np.cross(N, T)


def sample_ggx(m, outgoing, normal, U1, U2):                   # SampleBRDF_GGX, GGX_v6.hlsl:93-169 (U1, U2 = its two RandomFloat draws)
    alpha = m.Pr * m.Pr
    N, V = normalize(normal), normalize(outgoing)
    T1, T2 = coordinate_system(N)
    vx, vy, vz = dot(T1, V), dot(T2, V), dot(N, V)
    Ve = normalize(np.array([alpha * vx, alpha * vy, vz]))
    lensq = Ve[0] * Ve[0] + Ve[1] * Ve[1]
    T1h = np.array([-Ve[1], Ve[0], 0.0]) / np.sqrt(lensq) if lensq > 0.0 else np.array([1.0, 0.0, 0.0])
    T2h = np.cross(Ve, T1h)
    r = np.sqrt(U1)
    phi = 2.0 * PI * U2
    t1, t2 = r * np.cos(phi), r * np.sin(phi)
    s = 0.5 * (1.0 + Ve[2])
    t2 = (1.0 - s) * np.sqrt(saturate(1.0 - t1 * t1)) + s * t2
    Nh = t1 * T1h + t2 * T2h + np.sqrt(saturate(1.0 - t1 * t1 - t2 * t2)) * Ve
    Ne = normalize(np.array([alpha * Nh[0], alpha * Nh[1], max(0.0, Nh[2])]))
    H = Ne[0] * T1 + Ne[1] * T2 + Ne[2] * N
    I = -V
    smp = I - 2.0 * dot(H, I) * H                              # reflect(-V, H)
    if dot(smp, normal) < 0.0:
        smp = -smp
    return smp, H


def sample_lambert(normal, u1, u2):                            # SampleBRDF_Lambertian / RandomUnitVectorInHemisphere, Lambertian_v6.hlsl:2-38
    n = np.asarray(normal, np.float64)
    r = np.sqrt(u1)
    theta = 2.0 * 3.14159265358979 * u2
    x, y = r * np.cos(theta), r * np.sin(theta)
    z = np.sqrt(max(0.0, 1.0 - x * x - y * y))
    up = np.array([0.0, 0.0, 1.0]) if abs(n[2]) < float(np.float32(0.999)) else np.array([1.0, 0.0, 0.0])
    right = normalize(np.cross(up, n))
    fwd = np.cross(n, right)
    s = normalize(x * right + y * fwd + z * n)
    return -s if dot(s, n) < 0.0 else s


def tea(v0, v1):
    """RandomFloat, Common_v6.hlsl:119-138: 4 TEA rounds; -> (float(v0) / 2^32 as float32, new state)"""
    M, s = 0xFFFFFFFF, 0
    for _ in range(4):
        s = (s + 0x9E3779B9) & M
        v0 = (v0 + (((((v1 << 4) & M) + 0xA341316C) & M) ^ ((v1 + s) & M) ^ (((v1 >> 5) + 0xC8013EA4) & M))) & M
        v1 = (v1 + (((((v0 << 4) & M) + 0xAD90777D) & M) ^ ((v0 + s) & M) ^ (((v0 >> 5) + 0x7E95761E) & M))) & M
    return float(np.float32(v0) / np.float32(4294967296.0)), v0, v1


# ---- the host's Ess LUT generator, restated from ObjLoader.h:140-387 as a deterministic QUADRATURE ----------------------
def ess_generator_quadrature(roughness, idx, n=600):
    """What GenerateEssLUT / ComputeEss (ObjLoader.h:294-387) estimate for LUT entry `idx`, with the Monte-Carlo mean over (u1, u2)
    replaced by an n x n midpoint rule: -> (mean, standard deviation of ONE sample).  The host's SampleGGX (ObjLoader.h:176-252) is the
    older VNDF variant WITHOUT the warp of t2: a cosine-weighted direction around the stretched view vector, negative Nh.z clamped;
    the weight is NdotL * [G2 / (4 NdotV NdotL)] / [G1 / (4 NdotV)] with F = 1 and D cancelled (:256-289, :311-326)."""
    EPS = float(np.float32(0.04))
    cosT = EPS + idx / (LUT_SIZE_THETA - 1) * (1.0 - EPS)                     # :360
    sinT = np.sqrt(max(EPS, 1.0 - cosT * cosT))                              # :363
    N = np.array([0.0, 0.0, 1.0]); V = normalize(np.array([sinT, 0.0, cosT]))
    alpha = roughness * roughness
    T1, T2 = coordinate_system(N)
    Vh = normalize(np.array([dot(T1, V), dot(T2, V), dot(N, V)]))
    Vs = normalize(np.array([alpha * Vh[0], alpha * Vh[1], Vh[2]]))
    lensq = Vs[0] ** 2 + Vs[1] ** 2
    if lensq > 0.0:
        T1h = normalize(np.array([-Vs[1], Vs[0], 0.0]) / np.sqrt(lensq)); T2h = np.cross(Vs, T1h)
    else:
        T1h, T2h = np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0])
    u = (np.arange(n) + 0.5) / n
    U1, U2 = np.meshgrid(u, u, indexing="ij")
    r, phi = np.sqrt(U1), 2.0 * np.pi * U2
    x, y = r * np.cos(phi), r * np.sin(phi)
    z = np.sqrt(np.maximum(0.0, 1.0 - x * x - y * y))
    Nhs = normalize(x[..., None] * T1h + y[..., None] * T2h + z[..., None] * Vs)
    Nh = normalize(np.stack([alpha * Nhs[..., 0], alpha * Nhs[..., 1], np.maximum(0.0, Nhs[..., 2])], -1))
    H = normalize(Nh[..., 0:1] * T1 + Nh[..., 1:2] * T2 + Nh[..., 2:3] * N)
    L = normalize(-V - 2.0 * dot(H, -V)[..., None] * H)                      # reflect(-V, H)
    NdotLraw = dot(N, L)
    NdotL, NdotV = np.maximum(NdotLraw, 0.0), max(dot(N, V), 0.0)
    brdf = g2_smith(NdotV, np.maximum(NdotL, 1e-30), alpha) / np.maximum(4.0 * NdotV * NdotL, 1e-7)
    pdf = max(g1_smith(NdotV, alpha) / max(NdotV * 4.0, 1e-7), 1e-7)
    wgt = np.where(NdotLraw > 0.0, np.abs(NdotLraw) * brdf / pdf, 0.0)
    return float(wgt.mean()), float(wgt.std())


def directional_albedo_single_scatter(roughness, cosv, nt=1500, nphi=192):
    """int f_ss(L) (N.L) dL of the GGX lobe with F = 1 and NO multiscatter term, by quadrature over the half vector (dL = 4 (V.H) dH)"""
    m = Mat((0.5, 0.5, 0.5), (1.0, 1.0, 1.0), roughness, 1.0, np.ones(16))   # LUT = 1: kms = 0
    t = (np.arange(nt) + 0.5) / nt
    th, dth = 0.5 * np.pi * t ** 3, 0.5 * np.pi * 3.0 * t ** 2 / nt
    ph = (np.arange(nphi) + 0.5) * (2.0 * np.pi / nphi)
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    H = np.stack([np.sin(TH) * np.cos(PH), np.sin(TH) * np.sin(PH), np.cos(TH)], -1).reshape(-1, 3)
    w = (np.sin(TH) * dth[:, None] * (2.0 * np.pi / nphi)).reshape(-1)
    N = np.array([0.0, 0.0, 1.0]); V = np.array([np.sqrt(max(0.0, 1.0 - cosv * cosv)), 0.0, cosv])
    VH = H @ V
    ok = VH > 0.0
    L = 2.0 * VH[ok, None] * H[ok] - V
    up = L[:, 2] > 0.0
    f = ggx_eval(m, N, L[up], V)[:, 0]
    return float((f * L[up, 2] * 4.0 * VH[ok][up] * w[ok][up]).sum())


# ---- strategy 3 (EXTENSION; the reference has a stub only): rough dielectric transmission, Walter et al., "Microfacet Models for Refraction
# through Rough Surfaces" (EGSR 2007), eqs. 16, 17, 21 — restated from the paper, with this renderer's conventions: GGX D and Smith G of
# GGX_v6.hlsl, Schlick Fresnel with F0 = Ks, visible-normal sampling (pdf of h = G1(wo) |wo.h| D(h) / |wo.n|), no 1/eta^2 radiance scaling.
#   n on wo's side, eta_p = n_t / n_i
def btdf_eval(m, n, L, V, eta_p):
    """-> (f rgb, pdf) for wo = V (incident side) and wi = L (far side)"""
    N, V, L = normalize(n), normalize(V), normalize(L)
    NdotV, NdotL = float(dot(N, V)), float(dot(N, L))
    if not (NdotV > 0.0 and NdotL < 0.0):
        return np.zeros(3), 0.0
    ht = -(V + eta_p * L)                                  # eq. 16: h_t = -(eta_i i + eta_o o), here divided by eta_i
    ht = ht / np.sqrt(float(dot(ht, ht)))
    if float(dot(ht, N)) < 0.0:
        ht = -ht
    VH, LH = float(dot(V, ht)), float(dot(L, ht))
    if not (VH > 0.0 and LH < 0.0):
        return np.zeros(3), 0.0
    den = (VH + eta_p * LH) ** 2                           # (eta_i (i.h) + eta_o (o.h))^2 / eta_i^2
    alpha = m.Pr * m.Pr
    D = d_ggx(float(dot(N, ht)), m.Pr)
    G = g2_smith(NdotV, -NdotL, alpha)
    F = schlick(m.Ks, VH)
    # eq. 21: |i.h||o.h| / (|i.n||o.n|) * eta_o^2 (1 - F) G D / (eta_i (i.h) + eta_o (o.h))^2, with eta_o / eta_i = eta_p
    f = (1.0 - F) * (abs(VH) * abs(LH) / (abs(NdotV) * abs(NdotL)) * eta_p ** 2 * G * D / den)
    jac = eta_p ** 2 * abs(LH) / den                       # eq. 17: |dh / do|
    pdf = g1_smith(NdotV, alpha) * abs(VH) * D / abs(NdotV) * jac
    return f, pdf


def refract(V, H, eta_p):
    """Snell about the microfacet normal H (V.H > 0): the transmitted unit direction, or None on total internal reflection"""
    eta = 1.0 / eta_p
    c = float(dot(V, H))
    s2 = eta * eta * (1.0 - c * c)
    if s2 >= 1.0:
        return None
    return normalize((eta * c - np.sqrt(1.0 - s2)) * np.asarray(H) - eta * np.asarray(V))
