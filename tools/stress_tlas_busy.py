"""The multi-workgroup TLAS build while another stream keeps the GPU full of cull work: the chain's workgroups must
become co-resident (or time out and fall back) - either way the result is the same bytes and nothing hangs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import synth  # noqa: E402
from voidin_amd.runtime import Context  # noqa: E402

n = 20000
build = Context(0)
busy = Context(0, use_torch_stream=False)          # its own stream
meshes = synth.mesh_infos()
inst = synth.instances(n, seed=synth.SEED_BASE + 15, extent=500.0)
d_m, d_i = build.upload(meshes), build.upload(inst)
d_t = build.empty((2 * n + 1) * 32)
build.tlas_build_dev(d_i, n, d_m, len(meshes), d_t)
torch.cuda.synchronize()
want = d_t.cpu().numpy().tobytes()
cam = synth.camera_uniform()
big = synth.instances(4_000_000, seed=synth.SEED_BASE + 3, with_inverse=False)
d_big = build.upload(big)
d_out, d_cnt = build.empty(len(big) * 20), torch.zeros(4, dtype=torch.int32, device="cuda")
bad = 0
t0 = time.perf_counter()
for r in range(10):
    for _ in range(3000):                          # ~0.35 s of queued cull work on the other stream
        busy.cull_compact_dev(cam, d_m, len(meshes), d_big, len(big), d_out, d_cnt)
    d_t.zero_()
    t1 = time.perf_counter()
    build.tlas_build_dev(d_i, n, d_m, len(meshes), d_t)   # synchronises its own stream
    dt = time.perf_counter() - t1
    busy.synchronize()
    torch.cuda.synchronize()
    ok = d_t.cpu().numpy().tobytes() == want
    bad += not ok
    print(f"rep {r}: build under load {dt * 1e3:.0f} ms, {'same' if ok else 'DIFFERENT'}", flush=True)
print(f"{bad} different; total {time.perf_counter() - t0:.1f} s")
sys.exit(1 if bad else 0)
