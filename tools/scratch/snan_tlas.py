import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from voidin_amd import abi, synth
from voidin_amd.runtime import Context
from oracle import ref as oracle_mod
ctx = Context()
meshes = synth.mesh_infos()
for which, (row, col, val) in enumerate([(17, 12, 0x7FA00000), (200, 0, 0xFFA00000), (333, 9, 0x7F800001), (17, 12, 0x7FC00000), (200, 0, 0xFFC00000)]):
    inst = synth.instances(500, seed=synth.SEED_BASE + 61, extent=80.0)
    t = inst["transform"].view(np.uint32)
    t[row, col] = val
    got, want = ctx.tlas_build(inst, meshes), oracle_mod.tlas_build(inst, meshes)
    same = got.tobytes() == want.tobytes()
    print("case", which, hex(val), "same bytes:", same)
    if not same:
        for f in got.dtype.names:
            a, b = got[f], want[f]
            ne = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).reshape(len(a), -1).any(axis=1))[0]
            if len(ne):
                print("  field", f, "differs in", len(ne), "nodes; first", ne[:5])
                for k in ne[:3]:
                    print("    node", k, "got", a[k], a[k].view(np.uint32), "want", b[k], b[k].view(np.uint32))
