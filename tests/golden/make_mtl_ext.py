#!/usr/bin/env python3
"""SURVEY 8(f3) "full MTL PBR extension coverage": random .mtl files with EVERY statement tinyobjloader v2.0.0 reads (classic + PBR extension +
all map_* statements with their options) next to what the reference's vendored tinyobj parses from them (oracle/_ref/ref_probe).  Run in
the BUILD container only (needs /root/reference); writes tests/golden/mtlext/*.obj|.mtl and mtlext/ref.json, which pin the extended
MTL reader (royaltracer-dx_amd/host/ObjLoader.cpp: MaterialExt + texture ids)."""
import json
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "mtlext")
PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_probe")
N = 10
MAPS = ["map_Ka", "map_Kd", "map_Ks", "map_Ke", "map_Ns", "map_bump", "map_Bump", "bump", "map_d", "disp", "refl", "map_Pr", "map_Pm", "map_Ps", "norm"]
OPTS = ["-blendu on", "-blendv off", "-clamp on", "-boost 2.5", "-bm 0.35", "-o 0.1 0.2 0.3", "-o 0.5", "-s 2 2 2", "-s 1.5 0.5", "-t 0.1 0.1 0.1", "-mm 0.1 0.9", "-texres 512",
        "-imfchan r", "-type sphere", "-colorspace sRGB"]


def gen(k, rng):
    sep = lambda: rng.choice([" ", "  ", "\t"])
    M = []
    names = ["pbr%d_%d" % (k, i) for i in range(int(rng.integers(1, 5)))]
    for nm in names:
        M.append("newmtl " + nm)
        for key, cnt in (("Ka", 3), ("Kd", 3), ("Ks", 3), ("Ke", 3), ("Tf" if rng.random() < 0.5 else "Kt", 3)):
            if rng.random() < 0.75:
                M.append(key + sep() + sep().join("%g" % rng.uniform(0, 1) for _ in range(cnt)))
        for key in ("Ns", "Ni", "d", "Pr", "Pm", "Ps", "Pc", "Pcr", "aniso", "anisor"):
            if rng.random() < 0.7:
                M.append(key + sep() + "%g" % rng.uniform(0, 2))
        if rng.random() < 0.7:
            M.append("illum" + sep() + str(int(rng.integers(0, 11))))
        for key in rng.permutation(MAPS):
            if rng.random() < 0.45:
                opts = " ".join(rng.choice(OPTS, int(rng.integers(0, 4)), replace=False))
                fname = rng.choice(["tex_%d.png" % int(rng.integers(0, 6)), "dir/sub tex %d.jpg" % int(rng.integers(0, 3)), "T%d.exr" % k])
                M.append(key + sep() + (opts + " " if opts else "") + fname + rng.choice(["", " ", "\t"]))
        M.append("")
    L = ["mtllib mx%02d.mtl" % k, "v 0 0 0", "v 1 0 0", "v 0 1 0", "v 1 1 0"]
    for nm in names:
        L += ["usemtl " + nm, "f 1 2 3", "f 2 4 3"]
    return "\n".join(L) + "\n", "\n".join(M) + "\n"


def main():
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    ref = {}
    for k in range(N):
        rng = np.random.default_rng(9100 + k)
        obj, mtl = gen(k, rng)
        open(os.path.join(OUT, "mx%02d.obj" % k), "w").write(obj)
        open(os.path.join(OUT, "mx%02d.mtl" % k), "w").write(mtl)
        js = json.loads(subprocess.check_output([PROBE, "obj", os.path.join(OUT, "mx%02d.obj" % k), OUT + "/"], stderr=subprocess.DEVNULL), strict=False)
        ref["mx%02d" % k] = js["materials"]
    json.dump(ref, open(os.path.join(OUT, "ref.json"), "w"), indent=0)
    print("wrote", N, "files,", sum(len(v) for v in ref.values()), "materials,", sum(1 for v in ref.values() for m in v for t in m["tex"] if t), "texture statements")


if __name__ == "__main__":
    main()
