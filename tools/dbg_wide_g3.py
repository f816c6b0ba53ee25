import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yart_amd import api
g = dict(np.load("tests/golden/g3_edges.npz"))
a = api.Accel(g["verts"], g["tris"], g["spheres"])
got = api.hits_to_numpy(a.intersect_closest(torch.from_numpy(g["rays"]).cuda(), 0))
bad = np.nonzero(got["prim"] != g["prim"])[0]
print(len(bad), "mismatches; t bits equal:", (got["t"][bad].view(np.uint32) == g["t"][bad].view(np.uint32)).sum())
for i in bad[:40]:
    print(i, g["rays"][i], "got", got[i], "want", g["prim"][i], g["t"][i], g["u"][i], g["v"][i])
