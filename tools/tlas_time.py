"""Time vd_tlas_build[_wide]_dev at the bench sizes (bench.py's scenes), optionally against the oracle.
    python tools/tlas_time.py [--check] [--spec 1,2,0] [--profile] [sizes...]
env: VD_TLAS_INDEX=0 -> the r1 path, VD_TLAS_PHASE2, VD_TLAS_REFRESH.  --spec: VD_OPT_TLAS_SPEC values to A/B (0 no speculation,
1 helper waves in the chain's workgroup, 2 a helper workgroup on another CU of the same XCC); bytes compared with the first."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from voidin_amd import abi, synth  # noqa: E402
from voidin_amd.runtime import Context  # noqa: E402

check = "--check" in sys.argv
specs = [None]
if "--spec" in sys.argv:
    k = sys.argv.index("--spec")
    specs = [int(v) for v in sys.argv[k + 1].split(",")]
    del sys.argv[k:k + 2]
sizes = [int(x) for x in sys.argv[1:] if x.isdigit()] or [8192, 16384, 32768, 65536]
ctx = Context(0)
meshes = synth.mesh_infos()
d_m = ctx.upload(meshes)
for n in sizes:
    wide = n > 32768
    inst = synth.instances(n, seed=synth.SEED_BASE + (7 if n == 65536 else 6), extent=400.0 if n == 65536 else 300.0)
    d_i = ctx.upload(inst)
    d_t = ctx.empty((2 * n + 1) * (48 if wide else 32))
    first = None
    for spec in specs:
        if spec is not None:
            ctx.set_option("tlas.spec", spec)
        ts = []
        for rep in range(3):
            ctx.set_option("tlas.profile", 1 if ("--profile" in sys.argv and rep == 2) else 0)
            torch.cuda.synchronize(); t = time.perf_counter()
            ctx.tlas_build_dev(d_i, n, d_m, len(meshes), d_t, wide=wide)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        msg = f"n={n} wide={wide} spec={spec} build ms: " + " ".join(f"{x * 1e3:.1f}" for x in ts)
        b = d_t.cpu().numpy().tobytes()
        if first is None:
            first = b
        elif spec is not None:
            msg += f"  same bytes as spec={specs[0]}: {b == first}"
        if check:
            from oracle import ref
            want = ref.tlas_build(inst, meshes, wide=wide)
            got = d_t.cpu().numpy()[: (2 * n + 1) * (48 if wide else 32)].view(abi.TLAS_NODE_WIDE if wide else abi.TLAS_NODE)
            msg += f"  bit-exact vs oracle: {got.tobytes() == want.tobytes()}"
        print(msg, flush=True)
