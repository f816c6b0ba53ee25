"""TEST INFRASTRUCTURE ONLY.  Golden vectors for scene ingestion.

Writes the OBJ / MTL fixtures under tests/golden/obj/ (text authored here, not taken from anywhere)
and, for each, what the REAL reference makes of it: Scene::loadFromObj (src/scene.cpp:159-298)
through oracle/_ref/libyart_ref.so -> tests/golden/g5_obj_<name>.npz.  Runs only in the authoring
container.  Every face of every fixture has a material in force: without one the reference indexes
its Bsdf list with -1 (src/scene.cpp:261-266), which is undefined, so there is nothing to pin.

    python oracle/make_golden_obj.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
OBJ = os.path.join(OUT, "obj")

SYNTAX_MTL = """# materials for syntax.obj
newmtl white
Ka 0.1 0.1 0.1
Kd 0.75 0.7500001 7.5e-1
Ks 0 0 0
illum 2
Ns 10.5
Ni 1.0

newmtl \tmetal
Kd 0.2 0.3 0.4
Ks 0.9 0.8 0.7
Ns 96.078431
illum 5
map_Kd some texture.png

newmtl glass
Ni 1.52
Kd 1 1 1
illum 7
Tr 0.25
d 0.5
newmtl white
Kd 0.123 0.456 0.789
illum 2
\tnewmtl odd
 illum 4
Kd 1e-1 +2.5E-1 -0
Ns 1e
Ni .5
"""

SYNTAX_OBJ = """# ingestion syntax fixture
mtllib missing_file.mtl
mtllib syntax.mtl
\t# a comment after a tab
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.5 0.5 1.25e0
v\t-1.5\t+2.25\t-3.125E-1
v 007.5 1e2 1.0E+1
v 0.1 0.2 0.3
v 0.30000000000000004 0.1234567890123456789 123456789.123456789
v 3.14159274101257324 2.7182817459106445 1.41421353816986084
v -0 +0 0.0e5
v .5 1e 2.5x
v 1e-40 1e38 3.5e38
v 4 4
vn 0 0 1
vn 0 1 0
vn 1 0 0
vn 0.57735 0.57735 0.57735
vt 0 0
vt 1 0
vt 1 1

usemtl white
g quad with//normals
f 1//1 2//1 3//1 4//1
f 1//1 3//1 5//4
usemtl metal
f 1/1/2 2/2/2 6/3/2
f -14/1 -13/2 -9/3
f 1 2 7
usemtl glass
o pentagon
f 1 2 3 5 4
f 8 9 10
f 11 12 13
g
usemtl odd
f 1 3 2
f 14 1 2
f 2\t3\t4
usemtl white
g two words  trailing
f 5/3 4/2 3/1 2 1
f 1 2
usemtl metal
f 3 2 1\r
f 10//3 9//3 8//3\r
"""


def write_fixtures():
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, "syntax.mtl"), "w", newline="") as f:
        f.write(SYNTAX_MTL)
    with open(os.path.join(OBJ, "syntax.obj"), "w", newline="") as f:
        f.write(SYNTAX_OBJ)

    # terrain + fan: shared vertices (generated normals sum 1..120 face normals per vertex, i.e. through
    # several growth steps of the reference's unordered_set), coordinates printed with 1..17 digits
    rng = np.random.default_rng(5)
    lines = ["mtllib terrain.mtl", "usemtl ground", "o terrain"]
    n = 36
    h = rng.random((n + 1, n + 1))
    nv = 0
    for j in range(n + 1):
        for i in range(n + 1):
            digits = int(rng.integers(1, 18))
            x, y, z = i / n * 4 - 2, h[j, i] * 0.5, j / n * 4 - 2
            lines.append("v %.*f %.*g %.*e" % (digits, x, digits, y, digits, z))
            nv += 1
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i + 1
            b, c, d = a + 1, a + n + 2, a + n + 1
            if (i + j) % 3 == 0:
                lines.append("f %d %d %d %d" % (a, b, c, d))          # a quad: fanned
            else:
                lines.append("f %d %d %d" % (a, b, c))
                lines.append("f %d %d %d" % (a - nv - 1, c - nv - 1, d - nv - 1))   # relative numbers
    lines += ["usemtl rim", "g fan"]
    k = 120
    lines.append("v 0 1 0")
    for i in range(k):
        ang = 2 * np.pi * i / k
        lines.append("v %.9g %.9g %.9g" % (np.cos(ang) * (1 + 0.1 * rng.random()), 0.8 + 0.1 * rng.random(), np.sin(ang)))
    c0 = nv + 1
    for i in range(k):
        lines.append("f %d %d %d" % (c0, c0 + 1 + i, c0 + 1 + (i + 1) % k))
    # a third mesh that brings its own normals for every vertex
    lines += ["usemtl ground", "g with_normals", "vn 0 1 0", "vn 0.6 0.8 0"]
    lines += ["f 1//1 2//1 %d//2" % (n + 2), "f 2//1 %d//2 %d//2" % (n + 3, n + 2)]
    with open(os.path.join(OBJ, "terrain.obj"), "w", newline="") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(OBJ, "terrain.mtl"), "w", newline="") as f:
        f.write("newmtl ground\nKd 0.4 0.5 0.3\nillum 2\nnewmtl rim\nKd 0.9 0.9 0.9\nKs 0.5 0.5 0.5\nNs 30\nillum 5\n")


def run(name):
    folder = OBJ + os.sep
    d = orc.reference_load_obj(folder, name + ".obj")
    f = orc.reference_load_obj(folder, name + ".obj", so=orc.REF_FMA_SO)
    d["verts_fma"], d["normals_fma"] = f["verts"], f["normals"]
    path = os.path.join(OUT, "g5_obj_%s.npz" % name)
    np.savez_compressed(path, **d)
    sys.stdout.flush()
    print("%-12s %d meshes %d verts %d tris %d materials; fma build: verts equal %s, normals equal %.4f  (%d KB)"
          % (name, len(d["meshes"]), len(d["verts"]), len(d["tris"]), len(d["materials"]),
             np.array_equal(d["verts"].view(np.uint32), f["verts"].view(np.uint32)),
             (d["normals"].view(np.uint32) == f["normals"].view(np.uint32)).mean(), os.path.getsize(path) // 1024))


if __name__ == "__main__":
    orc.build_reference()
    write_fixtures()
    for name in ("syntax", "terrain"):
        run(name)
