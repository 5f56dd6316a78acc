"""Queue TLAS refits back to back (what a frame loop does) and report the per-call time; under
`rocprofv3 --kernel-trace` the trace shows where the time between the two kernels of a refit goes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import synth  # noqa: E402
from voidin_amd.runtime import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = Context(0)
meshes = synth.mesh_infos()
inst = synth.instances(n, seed=synth.SEED_BASE + 6, extent=300.0)
d_m, d_i = ctx.upload(meshes), ctx.upload(inst)
wide = n > 32768
d_t = ctx.empty((2 * n + 1) * (48 if wide else 32))
torch.cuda.synchronize(); t0 = time.perf_counter()
ctx.tlas_build_dev(d_i, n, d_m, len(meshes), d_t, wide=wide)
torch.cuda.synchronize()
print(f"n={n}: build {(time.perf_counter() - t0) * 1e3:.1f} ms (VD_TLAS_GROUPS={os.environ.get('VD_TLAS_GROUPS', 'default')})")
for _ in range(3):
    ctx.tlas_refit_dev(d_i, n, d_m, len(meshes), d_t, wide=wide)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.tlas_refit_dev(d_i, n, d_m, len(meshes), d_t, wide=wide)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"n={n}: {reps} queued refits: host enqueue {t_host / reps * 1e6:.1f} us/call, end to end {t_all / reps * 1e6:.1f} us/call")
