"""Scene generation and commit (BVH build) times of the two large procedural scenes on the GPU box: python tools/build_time.py"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import __graft_entry__ as g
rt = g.load_package()
for kind in ('sponza', 'bistro'):
    t0 = time.time(); sc = rt.Scene.sponza_class() if kind == 'sponza' else rt.Scene.bistro_class(); t1 = time.time()
    c = rt.Context(0); t2 = time.time(); c.upload(sc, 16 / 9); t3 = time.time()
    print(f"{kind}: generate {t1 - t0:.2f} s, upload+commit (BVH build) {t3 - t2:.2f} s, triangles {sc.num_triangles}")
    c.close()
