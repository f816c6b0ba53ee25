import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import load_golden
from yart_amd import api
g=load_golden(sys.argv[1] if len(sys.argv)>1 else 'g3_edges.npz')
a=api.Accel(g['verts'],g['tris'],g['spheres'])
for flags in (api.YART_REFERENCE_RULE|api.YART_STATIC_LAUNCH, 0):
    got=api.hits_to_numpy(a.intersect_closest(torch.from_numpy(g['rays']).cuda(), flags))
    bad=np.nonzero((got['prim']!=g['prim'])|(got['t'].view(np.uint32)!=g['t'].view(np.uint32)))[0]
    print('flags',flags,'bad',len(bad))
    for i in bad[:12]:
        print(i,g['rays'][i],'got',got[i],'want',(g['t'][i],g['u'][i],g['v'][i],g['prim'][i]))
