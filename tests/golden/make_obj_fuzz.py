#!/usr/bin/env python3
"""Random small OBJ / MTL files + what the reference's vendored tinyobjloader v2.0.0 parses from them (oracle/_ref/ref_probe).
Run in the BUILD container only (needs /root/reference); writes tests/golden/objfuzz/*.obj|.mtl and objfuzz/ref.npz, which pin
ObjLoader::loadObjFile (royaltracer-dx_amd/host/ObjLoader.cpp) beyond the two bundled scenes: index forms, negative indices, quads
(shorter-diagonal split), n-gons, groups / objects, comments, CRLF, tabs, unknown materials, Tr vs d, PBR keys, missing .mtl."""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "objfuzz")
PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_probe")
N = 24


def fnum(rng, x):
    return rng.choice(["%g" % x, "%.6f" % x, "%e" % x, ("+" if x >= 0 and rng.random() < 0.3 else "") + "%g" % x])


def gen(k, rng):
    eol = "\r\n" if k % 5 == 3 else "\n"
    sep = lambda: rng.choice([" ", "  ", "\t", " \t "])
    L, M = [], []
    nmat = int(rng.integers(0, 4))
    names = ["m%d_%s" % (i, rng.choice(["a", "B", "red.001", "x-y"])) for i in range(nmat)]
    have_mtl = nmat > 0 and k % 7 != 6
    if nmat:
        L.append("mtllib" + sep() + ("fz%02d.mtl" % k if have_mtl else "missing_%02d.mtl" % k))
    for nm in names:
        M.append("newmtl " + nm + rng.choice(["", " ", "\t"]))
        for key, cnt in (("Kd", 3), ("Ks", 3), ("Ke", 3)):
            if key == "Kd" and k % 8 == 5 and nm == names[0]:
                M.append("map_Kd first_%d.png" % k)                 # diffuse texture before any Kd in the file: tinyobj defaults Kd to 0.6
                continue
            if rng.random() < 0.8:
                M.append(rng.choice(["", "  ", "\t"]) + key + sep() + sep().join(fnum(rng, rng.uniform(0, 1) * (8 if key == "Ke" and rng.random() < .3 else 1)) for _ in range(cnt)))
        if rng.random() < 0.5: M.append("d" + sep() + fnum(rng, rng.uniform(0.1, 1)))
        if rng.random() < 0.4: M.append("Tr" + sep() + fnum(rng, rng.uniform(0, 0.9)))
        if rng.random() < 0.3: M.append("d" + sep() + fnum(rng, rng.uniform(0.1, 1)))           # d after Tr / repeated d
        for key in ("Pr", "Pm", "Ps", "Pc", "Ni", "Ns", "illum"):
            if rng.random() < 0.5: M.append(key + sep() + (str(int(rng.integers(0, 8))) if key == "illum" else fnum(rng, rng.uniform(0, 1.5))))
        if rng.random() < 0.3: M.append("map_Kd tex_%d.png" % k)
        if rng.random() < 0.3: M.append("# comment in mtl")
        M.append("")
    nv, nn, nt = 0, 0, 0
    def add_verts(c):
        nonlocal nv
        for _ in range(c):
            p = rng.normal(size=3) * 2
            ex = ""
            if rng.random() < 0.1: ex = sep() + fnum(rng, 1.0)                                   # w
            elif rng.random() < 0.1: ex = sep() + sep().join(fnum(rng, rng.uniform(0, 1)) for _ in range(3))   # vertex colour
            L.append("v" + sep() + sep().join(fnum(rng, x) for x in p) + ex)
            nv += 1
    def add_normals(c):
        nonlocal nn
        for _ in range(c):
            p = rng.normal(size=3); p /= np.linalg.norm(p)
            L.append("vn" + sep() + sep().join(fnum(rng, x) for x in p)); nn += 1
    def add_tex(c):
        nonlocal nt
        for _ in range(c):
            L.append("vt" + sep() + sep().join(fnum(rng, x) for x in rng.uniform(0, 1, 2))); nt += 1
    if rng.random() < 0.3: L.append("# header comment")
    for blk in range(int(rng.integers(1, 4))):
        if rng.random() < 0.6: L.append(rng.choice(["g", "o"]) + sep() + "part%d" % blk + rng.choice(["", " extra"]))
        add_verts(int(rng.integers(3, 9)))
        if rng.random() < 0.7: add_normals(int(rng.integers(1, 5)))
        if rng.random() < 0.5: add_tex(int(rng.integers(1, 4)))
        if rng.random() < 0.3: L.append("s" + sep() + rng.choice(["1", "off", "0"]))
        for f in range(int(rng.integers(1, 7))):
            if names and rng.random() < 0.5:
                L.append("usemtl" + sep() + (rng.choice(names) if rng.random() < 0.85 else "no_such_material") + rng.choice(["", " "]))
            cnt = int(rng.choice([3, 3, 3, 4, 4, 5, 6]))
            cnt = min(cnt, nv)
            vs = rng.choice(nv, cnt, replace=False)
            form = int(rng.integers(0, 4)) if nn else int(rng.choice([0, 2]) if nt else 0)       # 0: v  1: v//vn  2: v/vt  3: v/vt/vn
            if form in (2, 3) and not nt: form = 1 if form == 3 else 0
            neg = rng.random() < 0.3
            toks = []
            for vi in vs:
                a = str(int(vi) - nv) if neg else str(int(vi) + 1)
                ni = int(rng.integers(0, nn)) if nn else 0; ti = int(rng.integers(0, nt)) if nt else 0
                b = str(ni - nn) if neg else str(ni + 1); t = str(ti - nt) if neg else str(ti + 1)
                toks.append([a, a + "//" + b, a + "/" + t, a + "/" + t + "/" + b][form])
            L.append("f" + sep() + sep().join(toks) + rng.choice(["", " ", "\t"]))
        if rng.random() < 0.2: L.append("")
        if rng.random() < 0.2: L.append("   # indented comment")
    return eol.join(L) + eol, eol.join(M) + eol if have_mtl else None


def main():
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    ref = {}
    for k in range(N):
        rng = np.random.default_rng(4200 + k)
        obj, mtl = gen(k, rng)
        open(os.path.join(OUT, "fz%02d.obj" % k), "w", newline="").write(obj)
        if mtl is not None:
            open(os.path.join(OUT, "fz%02d.mtl" % k), "w", newline="").write(mtl)
        js = json.loads(subprocess.check_output([PROBE, "obj", os.path.join(OUT, "fz%02d.obj" % k), OUT + "/"], stderr=subprocess.DEVNULL), parse_int=float, strict=False)
        cat = lambda key, dt: np.concatenate([np.array(s[key], dt) for s in js["shapes"]]) if js["shapes"] else np.zeros(0, dt)
        ref["f%02d_vertices" % k] = np.array(js["vertices"], np.float32); ref["f%02d_normals" % k] = np.array(js["normals"], np.float32)
        ref["f%02d_vertex_index" % k] = cat("vertex_index", np.int32); ref["f%02d_normal_index" % k] = cat("normal_index", np.int32)
        ref["f%02d_material_ids" % k] = cat("material_ids", np.int32); ref["f%02d_num_face_vertices" % k] = cat("num_face_vertices", np.int32)
        ref["f%02d_materials" % k] = np.array([m["diffuse"] + m["specular"] + m["emission"] + [m["dissolve"], m["roughness"], m["metallic"], m["sheen"], m["clearcoat_thickness"], m["ior"]]
                                               for m in js["materials"]], np.float32).reshape(-1, 15)
    np.savez_compressed(os.path.join(OUT, "ref.npz"), **ref)
    print("wrote", N, "files; total triangles", sum(len(ref["f%02d_num_face_vertices" % k]) for k in range(N)))


if __name__ == "__main__":
    main()
