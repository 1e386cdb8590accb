#!/usr/bin/env python3
"""How full the machine is over ONE launch of the persistent closest-hit kernel (tooling build: make -C royaltracer-dx_amd VARIANT=wclk VARFLAGS="-DRTX_WAVE_CLOCK -DRTX_TRACE_WAVES=8" (the cap keeps the stamped kernel at the 64 VGPRs / 8 waves per SIMD of the product build)): every wave records its start and end
(s_memrealtime); the tool renders 1080p 16 spp with TWO bounces, so the recorded launch is bounce 1 (31 M incoherent rays), and prints the number of live waves over time.
usage: python tools/wave_timeline.py [sponza|bistro|cornell] [option_id=value ...]     (cornell: the fused tiny-scene kernel of bounces 1-7 of the headline frame)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RTX_LIB_PATH"] = os.environ.get("RTX_LIB_PATH") or os.path.join(ROOT, "royaltracer-dx_amd", "librtx_hip_wclk.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import __graft_entry__ as graft
rt = graft.load_package()
kind = next((a for a in sys.argv[1:] if a in ("sponza", "bistro", "cornell")), "sponza")
sc = {"sponza": rt.Scene.sponza_class, "bistro": rt.Scene.bistro_class, "cornell": rt.Scene.cornell}[kind]()
W, H = 1920, 1080
c = rt.Context(0)
for a in sys.argv[1:]:
    if "=" in a:
        c.set_option(int(a.split("=")[0]), int(a.split("=")[1]))
c.upload(sc, W / H); c.clear(W, H)
p = rt.Params(width=W, height=H, spp=16, max_bounces=2, nee_samples=1, flags=1 if kind == "sponza" else 4)
if kind == "cornell":            # the headline frame: the stamped launch is the fused kernel of bounces 1-7 (up to 10 240 workgroups = 40 960 waves)
    p = rt.Params(width=W, height=H, spp=64, max_bounces=8, nee_samples=1, rr_start=3, flags=1)
rt.lib.rtx_debug_wave_times.argtypes = [C.POINTER(C.c_ulonglong), C.c_uint, C.c_int]
c.render(p)
assert rt.lib.rtx_debug_wave_times(None, 0, 1) == 0
c.render(p)
N = 65536
out = (C.c_ulonglong * (2 * N))()
assert rt.lib.rtx_debug_wave_times(out, N, 0) == 0
a = np.frombuffer(out, dtype=np.uint64).reshape(N, 2).astype(np.int64)
sh = a[32768:][(a[32768:, 1] > 0) & (a[32768:, 0] > 0)] if kind != "cornell" else a[:0]      # general path: the any-hit launch of bounce 0 (overlaps the stamped closest-hit launch)
a = a[:32768] if kind != "cornell" else a
a = a[a[:, 1] > 0]
t0, t1 = a[:, 0].min(), a[:, 1].max()
span = (t1 - t0) / 100.0                                    # microseconds (100 MHz)
life = (a[:, 1] - a[:, 0]) / 100.0
ev = np.concatenate([np.stack([a[:, 0], np.ones(len(a), np.int64)], 1), np.stack([a[:, 1], -np.ones(len(a), np.int64)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
live = np.cumsum(ev[:, 1]); tt = (ev[:, 0] - t0) / 100.0
peak = live.max()
area = float(np.sum(live[:-1] * np.diff(tt)))
print(f"{kind}: {len(a)} waves, launch span {span:.0f} us, wave lifetime median {np.median(life):.0f} us (p5 {np.percentile(life, 5):.0f}, p95 {np.percentile(life, 95):.0f}), peak live waves {peak}")
print(f"  wave-time / (span x peak) = {area / (span * peak):.3f}   (1 = the machine holds its peak number of waves from the first to the last microsecond)")
grid = np.linspace(0, span, 21)
idx = np.searchsorted(tt, grid, side="right") - 1
print("  live waves at 0 %, 5 %, ... 100 % of the span: " + " ".join(str(int(live[max(i, 0)])) for i in idx))
below = tt[np.where(live < 0.9 * peak)[0]]
tail = below[below > 0.5 * span]
print(f"  first time after mid-launch with fewer than 90 % of the peak: {tail.min():.0f} us ({tail.min() / span:.1%} of the span)" if len(tail) else "  never below 90 % after mid-launch")
if len(sh):
    s0, s1 = sh[:, 0].min(), sh[:, 1].max()
    both = np.concatenate([a, sh])
    ev2 = np.concatenate([np.stack([both[:, 0], np.ones(len(both), np.int64)], 1), np.stack([both[:, 1], -np.ones(len(both), np.int64)], 1)])
    ev2 = ev2[np.argsort(ev2[:, 0], kind="stable")]
    live2 = np.cumsum(ev2[:, 1]); b0, b1 = both[:, 0].min(), both[:, 1].max(); tt2 = (ev2[:, 0] - b0) / 100.0
    sp2 = (b1 - b0) / 100.0
    idx2 = np.searchsorted(tt2, np.linspace(0, sp2, 21), side="right") - 1
    print(f"  any-hit launch of bounce 0: {len(sh)} waves, from {(s0 - t0) / 100.0:.0f} to {(s1 - t0) / 100.0:.0f} us on the closest-hit launch's clock")
    print(f"  BOTH launches: span {sp2:.0f} us, wave-time / (span x {live2.max()}) = {float(np.sum(live2[:-1] * np.diff(tt2))) / (sp2 * live2.max()):.3f}; live waves at 0 %, 5 %, ...: " + " ".join(str(int(live2[max(i, 0)])) for i in idx2))
c.close()
