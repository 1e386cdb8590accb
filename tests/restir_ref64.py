"""restir_ref64.py — a float64 restatement of the reference's SPATIAL pass for one pixel, written from the HLSL text alone
(/root/reference/Pathtracer/include/RayGen_v6_pass3.hlsl:46-372, MIS_v6.hlsl:2-59, MIS_GI_v6.hlsl:2-75, Sampler_v6.hlsl:1-22,48-68,86-188,
Common_v6.hlsl:203-350, Reservoir_v6.hlsl:2-80), NOT from oracle/rt_oracle.c or csrc/rtx_restir.hpp: the independent witness for the candidate search and its reject
predicates, the generalized pairwise-MIS weights (canonical / non-canonical, DI and GI), the reconnection Jacobian, the reservoir merges and the final shade, which
until round 3 were pinned only by "GPU == oracle".  Straight-line float64 with numpy's own sqrt / cos / sin; the BSDF is tests/ggx_ref64.py (itself restated from
GGX_v6.hlsl / BRDF_v6.hlsl / Lambertian_v6.hlsl).  Visibility comes through a callback (the witness has no ray tracer; tests hand it the oracle's any-hit query).

Buffers are the reference's byte layouts in MapPixelID order: Reservoir_DI / Reservoir_GI 40 B, SampleData 60 B (Reservoir_v6.hlsl:2-30)."""
import numpy as np
import ggx_ref64 as R

S_BIAS = float(np.float32(0.00002))
EPS = float(np.float32(0.000001))
SPATIAL_CANDIDATES, SPATIAL_TRIES, SPATIAL_RADIUS, M_CAP, W_SUM_T, J_T = 3, 9, 20, 128, 5.0, 5.0     # Common_v6.hlsl:14-25


def map_pixel_id(w, x, y):                                            # Common_v6.hlsl:173-198
    tcx = (w + 3) >> 2
    return ((y >> 2) * tcx + (x >> 2)) * 16 + (y & 3) * 4 + (x & 3)


def load_res(buf, slot):
    f = buf[slot].view(np.float32); h = buf[slot].view(np.float16); u = buf[slot].view(np.uint16)
    return dict(x2=f[0:3].astype(np.float64), w_sum=float(f[3]), n2=f[4:7].astype(np.float64), W=float(f[7]), L2=h[16:19].astype(np.float64), M=int(u[19]))


def load_sd(buf, slot):
    f = buf[slot].view(np.float32); h = buf[slot].view(np.float16); u16 = buf[slot].view(np.uint16); u32 = buf[slot].view(np.uint32)
    return dict(x1=f[0:3].astype(np.float64), mID=int(u16[6]), L1=h[7:10].astype(np.float64), n1=f[5:8].astype(np.float64), o=f[8:11].astype(np.float64), objID=int(u32[11]))


def length(v):
    return float(np.sqrt((np.asarray(v, np.float64) ** 2).sum()))


def normalize(v):
    v = np.asarray(v, np.float64)
    n = np.sqrt((v * v).sum())
    return v / n if n > 0 else v * 0.0


def reconnect_di(m, x1, n1, x2, n2, L, o):                            # ReconnectDI, Sampler_v6.hlsl:106-131
    d = x2 - x1
    dist = length(d)
    c1 = max(0.0, float(np.dot(n1, normalize(d))))
    if np.dot(n2, normalize(-d)) < 0.0:
        n2 = -n2
    c2 = max(0.0, float(np.dot(n2, normalize(-d))))
    with np.errstate(all="ignore"):
        F, _, _, _ = R.mixture(m, n1, normalize(d), normalize(o))
        return np.asarray(F, np.float64) * L * c1 * c2 / (dist * dist)


def reconnect_gi(m, x1, n1, x2, L, o):                                # ReconnectGI, Sampler_v6.hlsl:134-161
    d = x2 - x1
    c1 = abs(float(np.dot(n1, normalize(d))))
    with np.errstate(all="ignore"):
        F, _, _, _ = R.mixture(m, n1, normalize(d), normalize(o))
        fr = np.asarray(F, np.float64) * c1 * L
    return fr if np.isfinite(fr).all() else np.zeros(3)


def jacobian(sd_r, sd_q, x2q, n2q):                                   # Jacobian_Reconnection, Sampler_v6.hlsl:48-68
    vq, vr = x2q - sd_q["x1"], x2q - sd_r["x1"]
    with np.errstate(all="ignore"):
        cq = abs(float(np.dot(normalize(-vq), normalize(n2q)))); cr = abs(float(np.dot(normalize(-vr), normalize(n2q))))
        return np.float64(cq) / np.float64(cr) * (np.float64(np.dot(vr, vr)) / np.float64(np.dot(vq, vq)))


def reject_distance(x1, x2, cam, thr):                                # Common_v6.hlsl:340-347
    d1, d2 = length(x1 - cam), length(x2 - cam)
    with np.errstate(all="ignore"):
        return bool(np.float64(abs(d1 - d2)) / np.float64(max(d1, d2)) > thr)


def reject_jacobian(J, thr):                                          # Common_v6.hlsl:291-295
    return bool(J > thr or J < 1.0 / thr or np.isnan(J) or np.isinf(J))


class Rng:
    def __init__(self, x, y, sample, frame_seed):                     # RayGen_v6_pass3.hlsl:63-79
        M = 0xFFFFFFFF
        self.s0 = ((y * 73856093) & M) ^ ((x * 19349663) & M) ^ ((sample * 83492791) & M) ^ ((frame_seed * 293803) & M)
        self.s1 = ((x * 37623481) & M) ^ ((y * 51964263) & M) ^ ((sample * 68250729) & M) ^ ((frame_seed * 423977) & M)

    def next(self):
        v, self.s0, self.s1 = R.tea(self.s0, self.s1)
        return v


def random_pixel(radius, w, h, x, y, rng):                            # GetRandomPixelCircleWeighted, Common_v6.hlsl:203-243 (spatial_exponent 1)
    while True:
        u = rng.next()
        r = float(radius) * u
        ang = rng.next() * 6.2831853
        nx, ny = x + int(np.cos(ang) * r), y + int(np.sin(ang) * r)
        while nx < 0 or nx >= w:
            nx = -nx if nx < 0 else 2 * w - nx - 2
        while ny < 0 or ny >= h:
            ny = -ny if ny < 0 else 2 * h - ny - 2
        if not (nx == x and ny == y):
            return nx, ny


def spatial_pass_pixel(x, y, w, h, frame_seed, cam_pos, mats, cur_di, cur_gi, cur_sd, visible):
    """-> None when the pixel does not run the spatial pass (light seen directly / miss), else dict(di=..., gi=..., radiance=..., n_di, n_gi, rays).
    mats[mID] = ggx_ref64.Mat;  visible(x1, n1, x2) -> 1.0 / 0.0 = VisibilityCheck (Sampler_v6.hlsl:86-104)"""
    slot = map_pixel_id(w, x, y)
    sd = load_sd(cur_sd, slot)
    if not (sd["L1"] == 0.0).all() or sd["mID"] == 0xFFFE or sd["mID"] >= len(mats):
        return None
    m = mats[sd["mID"]]
    rng = Rng(x, y, 3, frame_seed)
    rays = 0

    def p_hat(s, r, vis):                                             # GetP_Hat, Sampler_v6.hlsl:163-171
        nonlocal rays
        f = length(reconnect_di(m, s["x1"], s["n1"], r["x2"], r["n2"], r["L2"], s["o"]))
        if vis:
            rays += 1
            return f * visible(s["x1"], s["n1"], r["x2"])
        return f

    def p_hat_gi(s, r, vis):                                          # GetP_Hat_GI, Sampler_v6.hlsl:173-181
        nonlocal rays
        f = reconnect_gi(m, s["x1"], s["n1"], r["x2"], r["L2"], s["o"])
        if vis:
            rays += 1
            return f * visible(s["x1"], s["n1"], r["x2"])
        return f

    rcur, gcur = load_res(cur_di, slot), load_res(cur_gi, slot)
    M_sum_di, M_sum_gi = float(min(M_CAP, rcur["M"])), float(min(M_CAP, gcur["M"]))
    cand_di, cand_gi = [], []
    for _ in range(SPATIAL_TRIES):                                    # pass3:106-135
        if len(cand_di) >= SPATIAL_CANDIDATES:
            break
        pr = map_pixel_id(w, *random_pixel(SPATIAL_RADIUS, w, h, x, y, rng))
        sn, rn = load_sd(cur_sd, pr), load_res(cur_di, pr)
        ok = (not np.dot(sd["n1"], sn["n1"]) < 0.9 and not reject_distance(sd["x1"], sn["x1"], cam_pos, 0.1)
              and length(rn["n2"]) > 0 and length(rn["L2"]) > 0 and rn["w_sum"] > 0 and rn["M"] > 0
              and length(sn["L1"]) == 0.0 and sn["mID"] != 0xFFFE and sn["mID"] == sd["mID"])
        if ok:
            cand_di.append(pr); M_sum_di += min(M_CAP, rn["M"])
    for _ in range(SPATIAL_TRIES):                                    # pass3:146-186
        if len(cand_gi) >= SPATIAL_CANDIDATES:
            break
        pr = map_pixel_id(w, *random_pixel(SPATIAL_RADIUS, w, h, x, y, rng))
        sn, gn = load_sd(cur_sd, pr), load_res(cur_gi, pr)
        with np.errstate(all="ignore"):
            ok = (m.Pr > 0.3 and not reject_distance(sd["x1"], sn["x1"], cam_pos, 0.1) and not np.dot(normalize(gn["x2"] - sd["x1"]), sd["n1"]) < 0.0
                  and not gn["w_sum"] > W_SUM_T and gn["w_sum"] > 0 and gn["M"] > 0 and not reject_jacobian(jacobian(sn, sd, gn["x2"], gn["n2"]), J_T)
                  and length(sn["L1"]) == 0.0 and sn["mID"] != 0xFFFE and sn["mID"] == sd["mID"])
        if ok:
            cand_gi.append(pr); M_sum_gi += min(M_CAP, gn["M"])
    can, can_gi = dict(rcur), dict(gcur)
    # GenPairwiseMIS_canonical, MIS_v6.hlsl:2-37
    cM = float(min(M_CAP, can["M"])); cMmax = M_sum_di - cM
    p_c = p_hat(sd, can, False)
    mi_c = cM / M_sum_di
    for pr in cand_di:
        sn, rn = load_sd(cur_sd, pr), load_res(cur_di, pr)
        den = cM * p_c + cMmax * p_hat(sn, can, True)
        if den > 0.0:
            mi_c += (min(M_CAP, rn["M"]) / M_sum_di) * (cM * p_c / den)
    w_c = mi_c * p_hat(sd, can, False) * can["W"]
    # GenPairwiseMIS_canonical_GI, MIS_GI_v6.hlsl:2-41
    gM = float(min(M_CAP, can_gi["M"])); gMmax = M_sum_gi - gM
    pg_c = length(p_hat_gi(sd, can_gi, False))
    mi_c_gi = gM / M_sum_gi
    for pr in cand_gi:
        sn, gn = load_sd(cur_sd, pr), load_res(cur_gi, pr)
        with np.errstate(all="ignore"):
            p_from = length(p_hat_gi(sn, can_gi, True)) * jacobian(sd, sn, can_gi["x2"], can_gi["n2"])
            den = gM * pg_c + gMmax * p_from
            if den > 0.0:
                mi_c_gi += (min(M_CAP, gn["M"]) / M_sum_gi) * (gM * pg_c / den)
    mi_c_gi = min(max(mi_c_gi, 0.0), 1.0)
    w_c_gi = mi_c_gi * length(p_hat_gi(sd, can_gi, False)) * can_gi["W"]
    rcur["M"], rcur["w_sum"] = int(min(M_CAP, can["M"])), w_c
    gcur["M"], gcur["w_sum"] = int(min(M_CAP, can_gi["M"])), w_c_gi
    picks_di, picks_gi = [], []
    for pr in cand_di:                                                # pass3:247-283, GenPairwiseMIS_noncanonical MIS_v6.hlsl:40-59
        sn, rn = load_sd(cur_sd, pr), load_res(cur_di, pr)
        num = (M_sum_di - cM) * p_hat(sn, can, False)
        den = num + cM * p_hat(sd, can, False)
        mi_s = (min(M_CAP, rn["M"]) / M_sum_di) * (num / den) if den > 0.0 else 0.0
        w_s = mi_s * p_hat(sd, rn, False) * rn["W"]
        rcur["w_sum"] += w_s; rcur["M"] = (rcur["M"] + min(M_CAP, rn["M"])) & 0xFFFF
        with np.errstate(all="ignore"):
            take = rng.next() < np.float64(w_s) / np.float64(rcur["w_sum"])
        picks_di.append(bool(take))
        if take:
            rcur["x2"], rcur["n2"], rcur["L2"] = rn["x2"], rn["n2"], rn["L2"]
    for pr in cand_gi:                                                # pass3:286-334, GenPairwiseMIS_noncanonical_GI MIS_GI_v6.hlsl:44-75
        sn, gn = load_sd(cur_sd, pr), load_res(cur_gi, pr)
        with np.errstate(all="ignore"):
            num = (M_sum_gi - gM) * length(p_hat_gi(sn, can_gi, False)) * jacobian(sd, sn, can_gi["x2"], can_gi["n2"])
            den = num + gM * length(p_hat_gi(sd, can_gi, False))
            mi_s = min(max((min(M_CAP, gn["M"]) / M_sum_gi) * (num / den), 0.0), 1.0) if den > 0.0 else 0.0
            j_gi = jacobian(sn, sd, gn["x2"], gn["n2"])
            w_s = mi_s * length(p_hat_gi(sd, gn, True)) * gn["W"] * j_gi
        if j_gi != 0.0:
            gcur["w_sum"] += w_s; gcur["M"] = (gcur["M"] + min(M_CAP, gn["M"])) & 0xFFFF
            with np.errstate(all="ignore"):
                take = rng.next() < np.float64(w_s) / np.float64(gcur["w_sum"])
            picks_gi.append(bool(take))
            if take:
                gcur["x2"], gcur["n2"], gcur["L2"] = gn["x2"], gn["n2"], gn["L2"]
    ph = p_hat(sd, rcur, True)                                        # pass3:336-372
    rcur["W"] = rcur["w_sum"] / ph if ph > EPS else 0.0
    acc = reconnect_di(m, sd["x1"], sd["n1"], rcur["x2"], rcur["n2"], rcur["L2"], sd["o"]) * rcur["W"]
    f_fin = p_hat_gi(sd, gcur, False)
    pg = length(f_fin)
    gcur["W"] = gcur["w_sum"] / pg if pg > EPS else 0.0
    acc = acc + f_fin * gcur["W"]
    return dict(di=rcur, gi=gcur, radiance=acc, n_di=len(cand_di), n_gi=len(cand_gi), rays=rays, cand_di=cand_di, cand_gi=cand_gi, picks_di=picks_di, picks_gi=picks_gi,
                weights=dict(mi_c=mi_c, mi_c_gi=mi_c_gi))


# ---- the TEMPORAL pass (RayGen_v6_pass2.hlsl:46-204, MIS_v6.hlsl:61-81, MIS_GI_v6.hlsl:78-110, Sampler_v6.hlsl:738-785) -----------------------------------------
TEMPORAL_M_CAP = 16


def temporal_pass_pixel(x, y, w, h, frame_seed, cam_pos, prev_view, prev_proj, inst, mats, cur_di, cur_gi, cur_sd, last_di, last_gi, last_sd, visible):
    """cur_* = this frame's pass-1 output, last_* = the history; inst[objID] = (objectToWorldInverse, prevObjectToWorld) as 4x4 float64 column-vector matrices;
    prev_view / prev_proj likewise.  -> None when the pixel leaves at once, else dict(di, gi, acc_di, acc_gi, pixel)"""
    slot = map_pixel_id(w, x, y)
    sd = load_sd(cur_sd, slot)
    if not (sd["L1"] == 0.0).all() or sd["mID"] == 0xFFFE or sd["mID"] >= len(mats):
        return None
    m = mats[sd["mID"]]
    rng = Rng(x, y, 2, frame_seed)
    rc, gc = load_res(cur_di, slot), load_res(cur_gi, slot)
    o2w_inv, prev_o2w = inst[sd["objID"] if sd["objID"] < len(inst) else 0]
    clip = prev_proj @ (prev_view @ (prev_o2w @ (o2w_inv @ np.append(sd["x1"], 1.0))))           # GetLastFramePixelCoordinates_Float
    if clip[3] <= 0.0:
        px, py = -1, -1
    else:
        uv = clip[:2] / clip[3] * 0.5 + 0.5
        px, py = int(np.rint(uv[0] * w)), int(np.rint((1.0 - uv[1]) * h))
    inside = 0 <= px < w and 0 <= py < h
    zero_r = dict(x2=np.zeros(3), w_sum=0.0, n2=np.zeros(3), W=0.0, L2=np.zeros(3), M=0)
    if inside:
        ts = map_pixel_id(w, px, py)
        rl, gl, sl = load_res(last_di, ts), load_res(last_gi, ts), load_sd(last_sd, ts)
    else:                                                             # (the reference reads out of bounds here: zeros)
        rl, gl, sl = dict(zero_r), dict(zero_r), dict(x1=np.zeros(3), mID=0, L1=np.zeros(3), n1=np.zeros(3), o=np.zeros(3), objID=0)
    base = px != -1 and py != -1 and length(sl["L1"]) == 0.0 and not reject_distance(sd["x1"], sl["x1"], cam_pos, 0.1) and sl["mID"] == sd["mID"]
    acc_di = base and length(rl["n2"]) > 0 and length(rl["L2"]) > 0 and rl["w_sum"] > 0 and rl["M"] > 0 and bool((rl["x2"] != 0.0).all())
    acc_gi = base and not gl["w_sum"] > W_SUM_T and gl["w_sum"] > 0 and gl["M"] > 0

    def pairwise(cM, lM):                                            # GenPairwiseMIS_{canonical,noncanonical}_temporal
        M_sum = cM + lM
        mi_c = cM / M_sum
        den = cM + (M_sum - cM)
        if den > 0.0:
            mi_c += (lM / M_sum) * (cM / den)
        num = M_sum - cM
        den2 = num + cM
        mi_t = (lM / M_sum) * num / den2 if den2 > 0.0 else 0.0
        return mi_c, mi_t
    if acc_di:
        cM, lM = float(min(TEMPORAL_M_CAP, rc["M"])), float(min(TEMPORAL_M_CAP, rl["M"]))
        with np.errstate(all="ignore"):
            mi_c, mi_t = pairwise(cM, lM)
            if length(rl["n2"]) == 0.0:
                mi_c, mi_t = 1.0, 0.0
            w_c = mi_c * length(reconnect_di(m, sd["x1"], sd["n1"], rc["x2"], rc["n2"], rc["L2"], sd["o"])) * rc["W"]
            w_t = mi_t * length(reconnect_di(m, sd["x1"], sd["n1"], rl["x2"], rl["n2"], rl["L2"], sd["o"])) * visible(sd["x1"], sd["n1"], rl["x2"]) * rl["W"]
            rc["M"], rc["w_sum"] = int(cM), w_c
            rc["w_sum"] += w_t; rc["M"] = (rc["M"] + int(lM)) & 0xFFFF
            if rng.next() < np.float64(w_t) / np.float64(rc["w_sum"]):
                rc["x2"], rc["n2"], rc["L2"] = rl["x2"], rl["n2"], rl["L2"]
            ph = length(reconnect_di(m, sd["x1"], sd["n1"], rc["x2"], rc["n2"], rc["L2"], sd["o"]))
            rc["W"] = rc["w_sum"] / ph if ph > EPS else 0.0
    if acc_gi:
        cM, lM = float(min(TEMPORAL_M_CAP, gc["M"])), float(min(TEMPORAL_M_CAP, gl["M"]))
        with np.errstate(all="ignore"):
            mi_c, mi_t = pairwise(cM, lM)
            w_c = mi_c * length(reconnect_gi(m, sd["x1"], sd["n1"], gc["x2"], gc["L2"], sd["o"])) * gc["W"]
            w_t = mi_t * length(reconnect_gi(m, sd["x1"], sd["n1"], gl["x2"], gl["L2"], sd["o"]) * visible(sd["x1"], sd["n1"], gl["x2"])) * gl["W"]
            gc["M"], gc["w_sum"] = int(cM), w_c
            gc["w_sum"] += w_t; gc["M"] = (gc["M"] + int(lM)) & 0xFFFF
            if rng.next() < np.float64(w_t) / np.float64(gc["w_sum"]):
                gc["x2"], gc["n2"], gc["L2"] = gl["x2"], gl["n2"], gl["L2"]
            pg = length(reconnect_gi(m, sd["x1"], sd["n1"], gc["x2"], gc["L2"], sd["o"]))
            gc["W"] = gc["w_sum"] / pg if pg > EPS else 0.0
    return dict(di=rc, gi=gc, acc_di=bool(acc_di), acc_gi=bool(acc_gi), pixel=(px, py))


# ---- PASS 1 (RayGen_v6_pass1.hlsl:48-190): SampleRIS (Sampler_v6.hlsl:653-736) with SampleLightNEE (:276-395) and SampleLightBSDF (:199-274), GetP_Hat / GetW
# (:163-188), SamplePathSimple (Path_Sampler_v6.hlsl:3-285) with SampleLightNEE_GI (Sampler_v6.hlsl:509-647) and SampleLightBSDF_GI (:397-506), UpdateReservoir /
# UpdateReservoir_GI (Reservoir_v6.hlsl:33-80) — restated in float64 from the HLSL text, round 4.  The camera ray is NOT restated (it is pinned bit for bit elsewhere): the
# pixel's primary hit (x1, n1, o, mID) is taken from the SampleData record the pass wrote.  Rays are answered by callbacks (tests hand in the oracle's closest-hit / any-hit
# queries); random numbers are the reference's TEA stream, so every draw happens in the reference's order or the pixel diverges visibly.
NEE_SAMPLES_DI, BSDF_SAMPLES_DI = 4, 1                                # Common_v6.hlsl:8-12 (nee_samples_DI = nee_samples)


def half3(v):
    return np.asarray(v, np.float64).astype(np.float32).astype(np.float16).astype(np.float64)


class Light:
    """LightTriangle (Renderer.h:113-124) as rtx_get_lights / orc_get_lights hand it out: 20 floats"""
    def __init__(self, rec20):
        f = np.asarray(rec20, np.float32); u = f.view(np.uint32)
        self.x, self.y, self.z = f[0:3].astype(np.float64), f[4:7].astype(np.float64), f[8:11].astype(np.float64)
        self.cdf, self.weight, self.emission = float(f[3]), float(f[11]), f[12:15].astype(np.float64)
        self.inst, self.count, self.total_weight = int(u[7]), int(u[15]), float(f[16])


def _lobes(m, normal, L, V_eval, V_pdf):
    """p_d, p_s (CalculateStrategyProbabilities with V_eval), brdf0 / brdf1 (EvaluateBRDF 0 / 1) and pdf0 / pdf1 (BRDF_PDF 0 / 1, with V_pdf) for the direction L = -incidence"""
    with np.errstate(all="ignore"):
        pd, ps = R.strategy_probs(m, V_eval, normal)
        b0, b1 = R.lambert_eval(m), R.ggx_eval(m, normal, L, V_eval)
        q0, q1 = R.lambert_pdf(normal, L), R.ggx_pdf(m, normal, L, V_pdf)
    return float(pd), float(ps), np.asarray(b0, np.float64), np.asarray(b1, np.float64), float(q0), float(q1)


def _safe(s, v):                                                      # SafeMultiply, Common_v6.hlsl:151-160
    with np.errstate(all="ignore"):
        r = np.float64(s) * np.asarray(v, np.float64)
    return r if np.isfinite(r).all() else np.zeros_like(np.atleast_1d(r), dtype=np.float64).reshape(np.shape(r))


def _select_light(lights, r):                                         # the binary search of SampleLightNEE(_GI)
    left, right, sel = 0, lights[0].count - 1, 0
    while left <= right:
        mid = left + (right - left) // 2
        if r < lights[mid].cdf:
            sel, right = mid, mid - 1
        else:
            left = mid + 1
    return lights[sel]


def _light_point(lt, o2w, rng):
    M = o2w[lt.inst]
    xv, yv, zv = (M @ np.append(lt.x, 1.0))[:3], (M @ np.append(lt.y, 1.0))[:3], (M @ np.append(lt.z, 1.0))[:3]
    xi1, xi2 = rng.next(), rng.next()
    if xi1 + xi2 > 1.0:
        xi1, xi2 = 1.0 - xi1, 1.0 - xi2
    u, v, w = 1.0 - xi1 - xi2, xi1, xi2
    return xv, yv, zv, u * xv + v * yv + w * zv


def _sample_brdf(m, strategy, outgoing, normal, rng):                 # SampleBRDF, BRDF_v6.hlsl:72-87: both lobes draw two numbers
    u1, u2 = rng.next(), rng.next()
    with np.errstate(all="ignore"):
        return R.sample_lambert(normal, u1, u2) if strategy == 0 else R.sample_ggx(m, outgoing, normal, u1, u2)[0]


def pass1_pixel(x, y, w, h, frame_seed, sd, mats, ke_full, lights, o2w, closest, occluded, visible, nee_samples=4, bounces=3):
    """sd = the pixel's SampleData (load_sd); mats[mID] = ggx_ref64.Mat, ke_full[mID] = materials[mID].Ke (float32 values); lights = [Light]; o2w[inst] = objectToWorld as a
    4x4 column-vector matrix; closest(origin, direction) -> dict(hit, pos, normal, mID) for a ray with TMin = s_bias (the reference's TraceRay + ClosestHit), occluded(origin,
    direction, tmin, tmax) -> bool (the shadow ray type), visible(x1, n1, x2) -> 1.0 / 0.0 (VisibilityCheck).
    -> None for a pixel that samples nothing (light seen directly, miss), else dict(di, gi, debug)"""
    if sd["mID"] == 0xFFFE or sd["mID"] >= len(mats) or length(ke_full[sd["mID"]]) > 0.0:
        return None
    m = mats[sd["mID"]]
    rng = Rng(x, y, 1, frame_seed)
    x1, n1, o = sd["x1"], sd["n1"], sd["o"]                            # payload.hitPosition, payload.hitNormal (unit), -direction
    di = dict(x2=np.zeros(3), w_sum=0.0, n2=np.zeros(3), W=0.0, L2=np.zeros(3), M=0)
    gi = dict(x2=np.zeros(3), w_sum=0.0, n2=np.zeros(3), W=0.0, L2=np.zeros(3), M=0)      # (xn, nn, E3 under the DI field names)
    M1, M2 = nee_samples, BSDF_SAMPLES_DI

    def update(res, wi, a, b, c):                                     # UpdateReservoir(_GI): the draw happens whatever wi is
        res["w_sum"] += wi
        with np.errstate(all="ignore"):
            take = rng.next() < np.float64(wi) / np.float64(res["w_sum"])
        if take:
            res["x2"], res["n2"], res["L2"] = np.asarray(a, np.float64), np.asarray(b, np.float64), half3(c)
        return bool(take)

    # ---------------- SampleRIS ----------------
    strategy = int(R.select_strategy(m, o, n1, rng.next()))
    for _ in range(M1):                                               # SampleLightNEE, useVisibility = false
        lt = _select_light(lights, rng.next())
        xv, yv, zv, sp = _light_point(lt, o2w, rng)
        L = sp - x1
        dist2 = float(np.dot(L, L))
        Ln = normalize(L)
        cr = np.cross(yv - xv, zv - xv)
        nl = normalize(cr)
        if np.dot(nl, -Ln) < 0.0:
            nl = -nl
        area = abs(length(cr) * 0.5)
        pdf_l = lt.weight / max(area, EPS)
        cos_x, cos_y = float(np.dot(n1, Ln)), float(np.dot(nl, -Ln))
        G = max(cos_y * cos_x / dist2, EPS)
        pd, ps, b0, b1, q0, q1 = _lobes(m, n1, Ln, normalize(o), normalize(o))
        brdf = _safe(pd, b0) + _safe(ps, b1)
        with np.errstate(all="ignore"):
            P = float(_safe(pd, q0 * cos_y / dist2)) + float(_safe(ps, q1 * cos_y / dist2))
            p_hat = length(lt.emission * brdf * G)
            pdf_light = max(EPS, pdf_l)
            mi = pdf_light / (M1 * pdf_light + M2 * P)
            wi = mi * p_hat / pdf_light
        if p_hat > 0.0:
            update(di, wi, sp, nl, lt.emission)
    for _ in range(M2):                                               # SampleLightBSDF
        smp = _sample_brdf(m, strategy, o, n1, rng)
        hit = closest(x1, smp)
        p_hat, pdf_light, P = 0.0, 0.0, 0.0
        if hit["hit"]:
            ke = np.asarray(ke_full[hit["mID"]], np.float64)
            if ke.sum() > EPS:
                L = hit["pos"] - x1
                dist = length(L); dist2 = dist * dist
                cos_t = float(np.dot(hit["normal"], -smp))
                pdf_light = (ke.sum() / 3.0) / lights[0].total_weight
                pd, ps, b0, b1, q0, q1 = _lobes(m, n1, smp, normalize(o), o)
                brdf = _safe(pd, b0) + _safe(ps, b1)
                with np.errstate(all="ignore"):
                    P = float(_safe(pd, q0 * cos_t / dist2)) + float(_safe(ps, q1 * cos_t / dist2))
                    p_hat = length(brdf * ke * float(np.dot(n1, smp)) * cos_t / dist2)
            if p_hat > 0.0:
                with np.errstate(all="ignore"):
                    mi = P / (M1 * pdf_light + M2 * P)
                    wi = mi * p_hat / P
                update(di, wi, hit["pos"], hit["normal"], ke)
    di["M"] = 1
    # ---------------- the reservoir's visibility and W (pass1:147-151) ----------------
    f_g = length(reconnect_di(m, x1, n1, di["x2"], di["n2"], di["L2"], o))
    with np.errstate(all="ignore"):
        p_hat = f_g * (visible(x1, n1, di["x2"]) if length(di["x2"] - x1) > 0.0 else 1.0)
    di["W"] = di["w_sum"] / p_hat if p_hat > EPS else 0.0
    # ---------------- SamplePathSimple ----------------
    acc_L = _path_simple(m, gi, x1, n1, o, mats, ke_full, lights, o2w, closest, occluded, rng, update, nee_samples, bounces)
    with np.errstate(all="ignore"):
        debug = acc_L + reconnect_di(m, x1, n1, di["x2"], di["n2"], di["L2"], o) * di["W"]
        f_c = length(reconnect_gi(m, x1, n1, gi["x2"], gi["L2"], o))
    gi["W"] = gi["w_sum"] / f_c if f_c > EPS else 0.0
    gi["M"] = 1
    return dict(di=di, gi=gi, debug=debug)


def _mat_opt(mats, ke_full, mID):
    return mats[mID], half3(ke_full[mID])


def _path_simple(m0, res, p0, n0, o0, mats, ke_full, lights, o2w, closest, occluded, rng, update, nee_samples, bounces):
    acc_f, acc_f_rec, acc_pdf = np.ones(3), np.ones(3), 1.0
    x1s, x2s = np.zeros(3), np.zeros(3)
    acc_L = np.zeros(3)
    origin, normal, outgoing, m = np.asarray(p0, np.float64), np.asarray(n0, np.float64), normalize(o0), m0
    # 1) the first BSDF bounce
    strategy = int(R.select_strategy(m, outgoing, normal, rng.next()))
    smp = _sample_brdf(m, strategy, outgoing, normal, rng)
    hit = closest(origin, smp)
    if not hit["hit"] or length(ke_full[hit["mID"]]) > 0.0:             # a light (or nothing: sanitised) straight away: no sample
        return acc_L
    inc = normalize(-smp)
    pd, ps, b0, b1, q0, q1 = _lobes(m, normal, -inc, outgoing, outgoing)
    F = _safe(pd, b0) + _safe(ps, b1)
    P = float(_safe(pd, q0)) + float(_safe(ps, q1))
    acc_pdf *= P
    acc_f = acc_f * (F * float(np.dot(normal, smp)))
    outgoing = inc
    m, ke_h = _mat_opt(mats, ke_full, hit["mID"])
    normal, origin = hit["normal"], hit["pos"]
    # 2) the reconnection vertex
    xn, nn = origin.copy(), normalize(normal)
    # 3) bounces with unshadowed NEE + one BSDF sample each
    for i in range(bounces):
        rng.next()                                                    # SelectSamplingStrategy: its result is overwritten before use, its draw is not (:118)
        for _ in range(nee_samples):                                  # SampleLightNEE_GI, useVisibility = false
            lt = _select_light(lights, rng.next())
            xv, yv, zv, sp = _light_point(lt, o2w, rng)
            L = sp - origin
            dist2 = float(np.dot(L, L))
            Ln = normalize(L)
            cr = np.cross(yv - xv, zv - xv)
            nl = normalize(cr)
            if np.dot(nl, -Ln) < 0.0:
                nl = -nl
            area = abs(length(cr) * 0.5)
            pdf_l = lt.weight / max(area, EPS)
            cos_x = abs(float(np.dot(normal, Ln))); cos_x = 0.0 if cos_x < EPS else cos_x
            cos_y = abs(float(np.dot(nl, -Ln))); cos_y = 0.0 if cos_y < EPS else cos_y
            pd, ps, b0, b1, q0, q1 = _lobes(m, normal, Ln, normalize(outgoing), normalize(outgoing))
            brdf = _safe(pd, b0) + _safe(ps, b1)
            Pb = float(_safe(pd, q0)) + float(_safe(ps, q1))
            pdf_light = 1.0                                           # the caller's initial value survives when cos_theta_y == 0 (:610-611)
            with np.errstate(all="ignore"):
                if cos_y > 0.0:
                    pdf_light = max(EPS, pdf_l) * dist2 / cos_y
                a_pdf = acc_pdf * pdf_light
                a_l = acc_f * (brdf * cos_x)
                thr = brdf * cos_x
                contribution = lt.emission * a_l / a_pdf if a_pdf > 0.0 else np.zeros(3)
                mi = pdf_light / (nee_samples * pdf_light + Pb)
                E_rec = acc_f_rec * mi * lt.emission * thr
                E_path = mi * contribution
                wi = length(E_path)
                acc_L = acc_L + mi * contribution
            if not np.isfinite(wi):
                wi = 0.0
            if update(res, wi, xn, normalize(nn), E_rec):
                x1s, x2s = origin + S_BIAS * normalize(normal), sp
        strategy = int(R.select_strategy(m, outgoing, normal, rng.next()))
        smp = _sample_brdf(m, strategy, outgoing, normal, rng)        # SampleLightBSDF_GI
        hit = closest(origin, smp)
        if not hit["hit"]:                                            # (sanitised: a miss ends the estimator here)
            break
        m2, ke2 = _mat_opt(mats, ke_full, hit["mID"])
        pd, ps, b0, b1, q0, q1 = _lobes(m, normal, smp, normalize(outgoing), outgoing)
        brdf = _safe(pd, b0) + _safe(ps, b1)
        Pb = float(_safe(pd, q0)) + float(_safe(ps, q1))
        NdotL = float(np.dot(normal, smp))
        pdf_light, contribution, emission = 1.0, np.zeros(3), np.zeros(3)
        with np.errstate(all="ignore"):
            acc_pdf *= Pb
            acc_f = acc_f * (brdf * NdotL)
            thr = brdf * NdotL
            if length(ke2) > 0.0:
                L = hit["pos"] - origin
                dist = length(L)
                cos_t = float(np.dot(hit["normal"], -smp))
                pdf_light = ((ke2.sum() / 3.0) / lights[0].total_weight) * (dist * dist) / cos_t
                emission = ke2
                contribution = ke2 * acc_f / acc_pdf
            acc_f_rec = acc_f_rec * thr
            if length(contribution) > 0.0:
                mi = Pb / (nee_samples * pdf_light + Pb)
                E_rec = acc_f_rec * mi * emission
                E_path = mi * contribution
                wi = length(E_path)
                acc_L = acc_L + E_path
                if not np.isfinite(wi):
                    wi = 0.0
                update(res, wi, xn, normalize(nn), E_rec)
                break
        origin, m, outgoing, normal = hit["pos"], m2, -smp, hit["normal"]
    if nee_samples > 0 and length(x2s - x1s) > EPS:                   # the reservoir's own shadow ray (:266-281)
        d = x2s - x1s
        if occluded(x1s, normalize(d), 0.5 * S_BIAS, max(S_BIAS, length(d) - S_BIAS * 5.0)):
            res["w_sum"] *= 0.0
    return acc_L
