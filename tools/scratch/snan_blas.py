import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from voidin_amd import abi, synth
from voidin_amd.runtime import Context
from oracle import ref as oracle_mod
ctx = Context()
for which, (row, col, val) in enumerate([(55, 2, 0xFFA00001), (55, 2, 0xFFA00000), (55, 2, 0x7FA00001), (55, 2, 0xFF800001), (55, 2, 0xFFE00001), (55, 2, 0xFFC00001), (7, 0, 0xFFA00001), (300, 1, 0xFFA00001)]):
    v, i = synth.triangle_soup(900, seed=79)
    v = v.copy(); v.view(np.uint32)[row, col] = val
    want, widx = oracle_mod.bvh_build(v, i)
    got, gidx = ctx.bvh_build(v, i)
    same = got.tobytes() == want.tobytes() and np.array_equal(gidx, widx)
    if not same:
        import collections
        print('  triangles with the vertex:', np.nonzero((i.reshape(-1,3) == row).any(axis=1))[0][:12])
    print("case", which, hex(val), "same:", same, "n", len(got), len(want))
    if not same:
        m = min(len(got), len(want)); got = got[:m]; want = want[:m]
        for f in got.dtype.names:
            a, b = got[f], want[f]
            ne = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).reshape(len(a), -1).any(axis=1))[0]
            if len(ne):
                print("  field", f, "differs in", len(ne), "nodes; first", ne[:8])
                for k in ne[:3]:
                    print("    node", k, "got", a[k], np.atleast_1d(a[k]).view(np.uint32), "want", b[k], np.atleast_1d(b[k]).view(np.uint32), "count", got["count"][k], want["count"][k])
        print("  idx differ:", int((gidx != widx).sum()))
        tri_with_nan = np.nonzero(np.isnan(v[i.reshape(-1, 3)]).any(axis=(1, 2)))[0]
        print("  triangles touching the NaN vertex:", tri_with_nan[:10])
