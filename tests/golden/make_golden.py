#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/.  Run in the BUILD container only
(it reads /root/reference through oracle/_ref/ref_probe and copies the two MIT-licensed scene files).

  ref_tinyobj_{garage,monke}.npz  raw parse of the reference's vendored tinyobjloader v2.0.0 (pins our OBJ/MTL reader)
  ref_glm_lookat.json             glm 0.9.8.5 lookAt matrices (pins Manipulator::setLookat / getMatrix)
  oracle_golden.npz               known-answer vectors produced by oracle/rt_oracle.c (SURVEY.md §8c list) —
                                  the reference itself holds no golden vectors, so these pin the oracle against
                                  regressions and travel to the GPU box as the expected outputs
"""
import json
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

REF = "/root/reference/Pathtracer"
PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_probe")

LOOKATS = [(-1.5, 1.5, 3.5, 0, 1, 0, 0, 1, 0),                                   # Renderer.cpp:46-48
           (278 / 555, 273 / 555, -475 / 555, 278 / 555, 273 / 555, 0, 0, 1, 0),  # Cornell camera
           (3.25, -2.5, 7.125, -1.0, 0.5, 0.25, 0.1, 0.9, 0.2)]
_rng = np.random.default_rng(31)
for _ in range(29):                                                                  # random eye / centre / up (not normalised, some nearly along the view)
    e, c = _rng.normal(size=3) * 10.0 ** _rng.integers(-1, 3), _rng.normal(size=3)
    u = _rng.normal(size=3) if _rng.random() < 0.7 else (c - e) * 0.98 + _rng.normal(size=3) * 0.2
    LOOKATS.append(tuple(float(x) for x in np.concatenate([e, c, u])))


def ref_fixtures():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    for name in ("garage", "monke"):
        for ext in ("obj", "mtl"):
            shutil.copy(f"{REF}/include/{name}.{ext}", HERE)
        js = json.loads(subprocess.check_output([PROBE, "obj", f"{REF}/include/{name}.obj", f"{REF}/include/"]), parse_int=float)   # keeps "-0" as -0.0
        vi = np.concatenate([np.array(s["vertex_index"], np.int32) for s in js["shapes"]])
        ni = np.concatenate([np.array(s["normal_index"], np.int32) for s in js["shapes"]])
        mid = np.concatenate([np.array(s["material_ids"], np.int32) for s in js["shapes"]])
        nfv = np.concatenate([np.array(s["num_face_vertices"], np.int32) for s in js["shapes"]])
        mats = np.array([m["diffuse"] + m["specular"] + m["emission"] + [m["dissolve"], m["roughness"], m["metallic"], m["sheen"], m["clearcoat_thickness"], m["ior"]]
                         for m in js["materials"]], np.float32)
        np.savez_compressed(os.path.join(HERE, f"ref_tinyobj_{name}.npz"), vertices=np.array(js["vertices"], np.float32),
                            normals=np.array(js["normals"], np.float32), vertex_index=vi, normal_index=ni, material_ids=mid,
                            num_face_vertices=nfv, materials=mats, material_names=np.array([m["name"] for m in js["materials"]]))
    out = []
    for la in LOOKATS:
        la32 = [float(np.float32(v)) for v in la]
        m = subprocess.check_output([PROBE, "lookat"] + [repr(v) for v in la32]).split()
        out.append({"args": la32, "matrix": [float(v) for v in m]})
    json.dump(out, open(os.path.join(HERE, "ref_glm_lookat.json"), "w"), indent=1)


def fixed_rays(n, seed, lo, hi, tmax):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    r = np.zeros((n, 8), np.float32)
    r[:, 0:3], r[:, 3], r[:, 4:7], r[:, 7] = o, 1e-4, d, tmax
    return r


def oracle_fixtures():
    rt = graft.load_package()
    orc = graft.load_oracle()
    g = {}
    # 1. TEA RNG + seeds
    for i, (x, y, s, t) in enumerate([(0, 0, 1, 0), (1919, 1079, 1, 12345), (7, 3, 2, 0xFFFFFFFF)]):
        seed = orc.seed_init(x, y, s, t)
        vals, end = orc.tea(seed, 8)
        g[f"tea{i}_seed"], g[f"tea{i}_vals"], g[f"tea{i}_end"] = np.array(seed, np.uint32), vals, np.array(end, np.uint32)
    # 2. Cornell: lights, primary rays, hits, surfaces, images
    sc = rt.Scene.cornell()
    o = orc.Oracle().load(sc, 16 / 9)
    g["cornell_lights"] = o.lights()
    p = rt.Params(width=1920, height=1080)
    rays = o.primary_rays(p, 1)
    pick = [0, 960 + 540 * 1920, 1919 + 1079 * 1920]
    g["cornell_primary_pick"] = rays[pick]
    r = np.concatenate([o.primary_rays(rt.Params(width=32, height=16), 1), fixed_rays(512, 11, -0.2, 1.2, 1e4)])
    g["cornell_rays"] = r
    g["cornell_hits"] = o.trace_closest(r, 0)
    g["cornell_surface"] = o.surface(r, g["cornell_hits"])
    sr = fixed_rays(1024, 12, 0.05, 0.95, 0.6)
    g["cornell_shadow_rays"], g["cornell_shadow_occ"] = sr, o.trace_any(sr, 0)
    for tag, kw in (("c1", dict(spp=1, max_bounces=4)), ("c2", dict(spp=4, max_bounces=8))):
        o.set_camera(*sc.view_proj(64 / 36))
        acc, cnt = o.render(rt.Params(width=64, height=36, nee_samples=1, flags=1, **kw))
        g[f"cornell_{tag}_accum"], g[f"cornell_{tag}_rays"] = acc, np.array(cnt, np.uint64)
    # 3. BSDF table: 64 (n, wo, wi, seed) tuples on three materials, both modes
    rng = np.random.default_rng(21)
    def unit(k):
        v = rng.normal(size=(k, 3)); return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    n_, wo, wi = unit(64), unit(64), unit(64)
    flip = (n_ * wo).sum(1) < 0; wo[flip] = -wo[flip]
    seeds = rng.integers(0, 2**32, size=(64, 2), dtype=np.uint64).astype(np.uint32).view(np.float32)
    g["bsdf_in_eval"], g["bsdf_in_sample"] = np.concatenate([n_, wo, wi], 1), np.concatenate([n_, wo, seeds], 1)
    # 4. garage + monke (the reference's own startup scene): GGX materials, instances, smooth normals
    gs = rt.Scene.from_obj([os.path.join(HERE, "garage.obj"), os.path.join(HERE, "monke.obj")], HERE + "/")
    og = orc.Oracle().load(gs, 16 / 9)
    g["garage_lights_head"] = og.lights()[:8]
    g["garage_num_lights"] = np.array([len(og.lights())], np.uint32)
    gr = np.concatenate([og.primary_rays(rt.Params(width=32, height=18), 1), fixed_rays(448, 13, -3, 3, 1e4)])
    g["garage_rays"], g["garage_hits"] = gr, og.trace_closest(gr, 0)
    g["garage_surface"] = og.surface(gr, g["garage_hits"])
    for flags in (1, 0):
        for mat in (1, 2, 5):
            g[f"bsdf_eval_f{flags}_m{mat}"] = og.bsdf_eval(mat, flags, g["bsdf_in_eval"])
            g[f"bsdf_sample_f{flags}_m{mat}"] = og.bsdf_sample(mat, flags, g["bsdf_in_sample"])
    og.set_camera(*gs.view_proj(64 / 36))
    acc, cnt = og.render(rt.Params(width=64, height=36, spp=2, max_bounces=6, nee_samples=2, flags=0))
    g["garage_accum"], g["garage_rays_count"] = acc, np.array(cnt, np.uint64)
    # 5. the reference's own pipeline: pass-1 estimator and two ReSTIR frames (pass 1 + temporal + spatial)
    o.set_camera(*sc.view_proj(48 / 28)); o.set_camera(*sc.view_proj(48 / 28))
    p1 = rt.Params(width=48, height=28, spp=1, max_bounces=3, nee_samples=4, flags=0, frame_seed=3)
    acc, (di, gi, sd), cnt = o.render_v6_pass1(p1)
    g["pass1_accum"], g["pass1_di"], g["pass1_gi"], g["pass1_sd"], g["pass1_rays"] = acc, di, gi, sd, np.array(cnt, np.uint64)
    acc, st, cnt = o.restir_frames(rt.Params(width=48, height=28, spp=2, max_bounces=3, nee_samples=4, flags=0, frame_seed=3))
    g["restir_accum"], g["restir_last_di"], g["restir_last_gi"], g["restir_last_sd"], g["restir_rays"] = acc, st[3], st[4], st[5], np.array(cnt, np.uint64)
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **g)
    print({k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    if os.path.isdir(REF):
        ref_fixtures()
    oracle_fixtures()
