#!/usr/bin/env python3
"""Same-box, same-process alternating A/B of rtx options on one workload: the scene is built once, the option values alternate round by round.
usage: python tools/ab_frame.py <sponza|bistro|garage|cornell> <pt|restir> <opt>=<a>,<b>[,...] [rounds=3] [frames=3] [timing=0] [same=0] [fixed <opt>=<v> ...]
prints per setting: ms per frame (wall, all rounds), per-kernel-class ms (last round, only with timing=1: launches serialised; the default timing=0 leaves the frame as a caller gets it) and the image checksum"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import __graft_entry__ as graft
rt = graft.load_package()
gd = os.path.join(ROOT, "tests", "golden")
ctors = {"cornell": rt.Scene.cornell, "sponza": rt.Scene.sponza_class, "bistro": rt.Scene.bistro_class,
         "garage": lambda: rt.Scene.from_obj([os.path.join(gd, "garage.obj"), os.path.join(gd, "monke.obj")], gd + "/")}
kind, mode = sys.argv[1], sys.argv[2]
kv = [a for a in sys.argv[3:] if "=" in a]
opt, vals = kv[0].split("="); opt = int(opt); vals = [int(v) for v in vals.split(",")]
named = {a.split("=")[0]: int(a.split("=")[1]) for a in kv[1:] if not a.split("=")[0].isdigit()}
fixed = [(int(a.split("=")[0]), int(a.split("=")[1])) for a in kv[1:] if a.split("=")[0].isdigit()]
rounds, frames, timing = named.get("rounds", 3), named.get("frames", 3), named.get("timing", 0)
W, H = 1920, 1080
sc = ctors[kind]()
same = named.get("same", 0)                     # same=1: ONE context, the option is switched between rounds (run-time options only; removes context-to-context placement effects, +-1.5 %)
ctxs = {}
for v in vals:                                   # one context per setting (options that change the commit need their own scene upload)
    if same and ctxs:
        ctxs[v] = next(iter(ctxs.values()))
        continue
    c = rt.Context(0)
    for k, x in fixed:
        c.set_option(k, x)
    c.set_option(opt, v)
    c.upload(sc, W / H)
    ctxs[v] = c
if mode == "restir":
    p = rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=0)
else:
    p = rt.Params(width=W, height=H, spp=16 if kind != "cornell" else 64, max_bounces=8, nee_samples=1, rr_start=3, sample_base=1,
                  flags=1 if kind in ("sponza", "cornell") else (4 if kind == "bistro" else 0))


def run(c, seed0):
    c.clear(W, H)
    if mode == "restir":
        c.restir_reset()
    t0 = time.perf_counter()
    for f in range(frames):
        q = p.copy(frame_seed=seed0 + f)
        (c.render_restir if mode == "restir" else c.render)(q)
    return (time.perf_counter() - t0) * 1e3 / frames


res = {v: [] for v in vals}
for v in vals:
    if same:
        ctxs[v].set_option(opt, v)               # (one context per value: set before its upload — options that change the commit must not be touched again)
    run(ctxs[v], 1)                              # warm-up (allocations)
for r in range(rounds):
    for v in vals:
        if same:
            ctxs[v].set_option(opt, v)
            run(ctxs[v], 1)                      # one untimed pass after a switch (launch sizes are predicted from the previous call)
        res[v].append(run(ctxs[v], 1))
for v in vals:
    c = ctxs[v]
    if same:
        c.set_option(opt, v); run(c, 1)
    sha = hashlib.sha1(c.read_accum().tobytes()).hexdigest()[:12]
    km = ""
    if timing:
        c.set_option(rt.OPT_KERNEL_TIMING, 1); run(c, 1); st = c.stats()
        km = "  " + ", ".join(f"{rt.KERNEL_NAMES[i]} {st.kernel_ms[i]:.2f}" for i in rt.KERNEL_NAMES if st.kernel_launches[i] > 0)
    st = c.stats()
    print(f"{kind} {mode} opt {opt}={v}: " + " ".join(f"{t:.3f}" for t in res[v]) + f" ms/frame  (min {min(res[v]):.3f})  rays {st.rays_primary + st.rays_extension + st.rays_shadow}  sha {sha}{km}", flush=True)
    if not same:
        c.close()
