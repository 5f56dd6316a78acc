"""The north-star exchange behind the C ABI on ONE GPU: a one-rank RCCL communicator (vd_dist_create, world = 1) runs
vd_dist_step_full_dev = cull_mask -> ncclAllGather on the ctx stream -> expand_mask, and vd_dist_step_draws_dev (the
literal 20-byte exchange); both must equal vd_cull_compact_dev and the oracle bit for bit.  Runs in a child process
under a timeout: a communicator that cannot initialise must fail the test, not hang the box.  (N > 1 ranks over RCCL
need one GPU per rank: the driver's 8-GPU run; the multi-rank logic is covered on gloo in test_dist_gloo.py /
test_gpu_dist.py.)"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

PRELUDE = r'''
import sys
import numpy as np
import torch
sys.path.insert(0, {root!r})
from oracle import ref
from voidin_amd import abi, synth
from voidin_amd import dist as vdist
from voidin_amd.runtime import Context

torch.cuda.set_device(0)
ctx = Context(0)
cam, meshes = synth.camera_uniform(), synth.mesh_infos()
'''

CHILD = PRELUDE + r'''
for n, kw in ((70_001, dict(scale_range=(0.02, 0.6), extent=600.0)), (abi.CULL_SPLIT_MIN + 12_345, dict(scale_range=(0.25, 4.0)))):
    inst = synth.instances(n, seed=synth.SEED_BASE + 91, with_inverse=False, **kw)
    inst["mesh"][::97] = 0xFFFFFFF0                       # ids beyond the table: the unsigned clamp is part of the contract
    want, wn = ref.compact(ref.cull_emit(cam, meshes, inst, threads=8))
    d_m, d_i = ctx.upload(meshes), ctx.upload(inst)
    rv = vdist.RcclVisibility(ctx, n, d_m, len(meshes), d_i)            # world = 1: no process group needed
    assert rv.world == 1 and rv.info.shard_size == n and rv.info.rccl_version > 20000, rv.info.rccl_version
    d_out = ctx.empty(n * 20)
    d_cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
    for name, fn in (("full", rv.step), ("draws", rv.step_draws)):
        d_out.fill_(0xEE); d_cnt.zero_()
        for _ in range(3):
            fn(cam, d_out, d_cnt)
        ctx.synchronize()
        cnt = int(d_cnt[0].item())
        assert cnt == wn, (name, cnt, wn)
        assert d_out[: cnt * 20].cpu().numpy().tobytes() == want[:wn].tobytes(), name
    # the legs one by one (what bench.py times as step_breakdown)
    d_out.fill_(0); d_cnt.zero_()
    rv.cull_to_mask(cam); rv.allgather_masks(); rv.expand_all(d_out, d_cnt)
    ctx.synchronize()
    assert int(d_cnt[0].item()) == wn and d_out[: wn * 20].cpu().numpy().tobytes() == want[:wn].tobytes()
    print("ok", n, wn, rv.info.rccl_version, rv.info.rccl_library.decode())
    rv.close()
print("RCCL_WORLD1_OK")
'''

GRAPH_CHILD = PRELUDE + r'''
# a step captured into a HIP graph and replayed: the collective sits on the ctx stream like the kernels around it
n = 300_000
inst = synth.instances(n, seed=synth.SEED_BASE + 92, with_inverse=False)
want, wn = ref.compact(ref.cull_emit(cam, meshes, inst, threads=8))
d_m, d_i = ctx.upload(meshes), ctx.upload(inst)
rv = vdist.RcclVisibility(ctx, n, d_m, len(meshes), d_i)
d_out, d_cnt = ctx.empty(n * 20), torch.zeros(4, dtype=torch.int32, device="cuda")
rv.step(cam, d_out, d_cnt); ctx.synchronize()
g = torch.cuda.CUDAGraph()
main_stream = torch.cuda.current_stream().cuda_stream
try:
    try:
        with torch.cuda.graph(g):
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            rv.step(cam, d_out, d_cnt)
    finally:
        ctx.set_stream(main_stream)
    d_out.fill_(0); d_cnt.zero_()
    g.replay(); torch.cuda.synchronize()
    assert int(d_cnt[0].item()) == wn and d_out[: wn * 20].cpu().numpy().tobytes() == want[:wn].tobytes()
    print("graph replay ok")
except Exception as e:                                   # capture of a collective is RCCL's to allow
    print("graph capture not available:", repr(e)[:200])
    raise SystemExit(3)
print("RCCL_GRAPH_OK")
'''



def test_rccl_world1_step_equals_single_gpu_compaction():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run(["timeout", "840", sys.executable, "-c", CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "RCCL_WORLD1_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_rccl_step_replays_from_a_hip_graph():
    """vd_dist_step_full_dev only enqueues (two kernels-with-memsets and one collective on the ctx stream), so a step can
    be captured once and replayed.  The capture of a collective passed on the driver's box in round 3: a timeout or a wrong
    replayed result FAILS; only RCCL explicitly refusing the capture (the child's exit code 3, with its message) is a skip."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run(["timeout", "700", sys.executable, "-c", GRAPH_CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=800, env=env)
    assert out.returncode != 124, "graph capture / replay of the RCCL step did not finish in 240 s:\n" + out.stdout[-1500:] + out.stderr[-1500:]
    if out.returncode == 3:
        pytest.skip("RCCL refused the graph capture of a collective: " + out.stdout[-300:])
    assert out.returncode == 0 and "RCCL_GRAPH_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
