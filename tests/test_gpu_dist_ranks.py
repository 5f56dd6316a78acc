"""The `world > 1` code of the C-ABI exchange (voidin_amd/csrc/dist.hip) on ONE GPU: 2, 3 and 8 rank processes share
cuda:0 and bind a test-double RCCL (tests/cpp/fake_rccl.cpp -> $VD_RCCL_LIB; real RCCL refuses two ranks per device).
Every rank's vd_dist_step_full_dev / vd_dist_step_draws_dev / vd_dist_step_indices_dev list must equal
vd_cull_compact_dev on the whole scene byte for byte - with an unequal last shard, ranks whose shard is empty
(n_local == 0), a rank with zero survivors, 2-byte mesh ids, at 100 003 and at 10 M instances (BASELINE configs[3]).
The double reports what it executed, so a pass means the all-gathers and the grouped ncclSend / ncclRecv block ran."""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT

FAKE_SRC = os.path.join(ROOT, "tests", "cpp", "fake_rccl.cpp")
FAKE_LIB = os.path.join(ROOT, "tests", "cpp", "libfake_rccl.so")
WORKER = os.path.join(ROOT, "tests", "dist_rank_worker.py")


def build_fake():
    if not os.path.exists(FAKE_LIB) or os.path.getmtime(FAKE_LIB) < os.path.getmtime(FAKE_SRC):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-fPIC", "-shared", FAKE_SRC, "-o", FAKE_LIB],
                       check=True, capture_output=True, timeout=300)
    return FAKE_LIB


def test_fake_rccl_builds_and_exports_what_dist_hip_binds():
    """CPU check: the double compiles and carries the ten symbols load_rccl() resolves (csrc/dist.hip)."""
    lib = build_fake()
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    have = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    src = open(os.path.join(ROOT, "voidin_amd", "csrc", "dist.hip")).read()
    import re
    bound = set(re.findall(r'VD_SYM\(\w+, "(nccl\w+)"\)', src))
    assert len(bound) == 10 and bound <= have, sorted(bound - have)
    # the product never names the double
    for root, _, files in os.walk(os.path.join(ROOT, "voidin_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                assert "libfake_rccl" not in open(os.path.join(root, f), errors="ignore").read(), f
    assert "libfake_rccl" not in open(os.path.join(ROOT, "bench.py")).read()


def _scratch_dir():
    """Where the ranks' shared segments live: /dev/shm when it has room (the 10 M draws exchange stages ~200 MB), else
    the temp dir (file-backed mmap works the same)."""
    base = None
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > (2 << 30):
            base = "/dev/shm"
    except OSError:
        pass
    return tempfile.mkdtemp(prefix="vd_ranks_", dir=base)


def run_ranks(ctx, world, n, n_mesh=16, kw=None, hidden_rank=None, fail_send=None, timeout=600):
    """Start `world` rank processes; returns (results by rank, want_n)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_rank_worker as W
    from voidin_amd import dist as vdist
    from voidin_amd import synth
    kw = kw or dict(scale_range=(0.02, 0.6), extent=600.0)
    job = {"world": world, "n": n, "n_mesh": n_mesh, "seed": synth.SEED_BASE + 61, "kw": kw}
    if hidden_rank is not None:
        job["hidden"] = list(vdist.shard_range(n, hidden_rank, world))
    # the whole scene on this process's context: vd_cull_compact_dev is the thing every rank must reproduce
    inst = W.scene_shard(job, 0, n)
    cam, meshes = synth.camera_uniform(), synth.mesh_infos(n_mesh)
    d_m, d_i = ctx.upload(meshes), ctx.upload(inst)
    d_out, d_cnt = ctx.empty(n * 20), torch.zeros(4, dtype=torch.int32, device="cuda")
    ctx.cull_compact_dev(cam, d_m, n_mesh, d_i, n, d_out, d_cnt)
    ctx.synchronize()
    want_n = int(d_cnt[0].item())
    want = d_out[: want_n * 20].cpu().numpy().tobytes()
    del d_i, d_out, inst
    torch.cuda.empty_cache()
    if hidden_rank is not None:
        lo, hi = job["hidden"]
        base = np.frombuffer(want, dtype=np.uint32).reshape(-1, 5)[:, 4]
        assert hi > lo and not ((base >= lo) & (base < hi)).any(), "the hidden shard has survivors: test set-up"
    d = _scratch_dir()
    try:
        job.update(dir=d, want_n=want_n, want_digest=hashlib.blake2b(want, digest_size=16).hexdigest(), want_raw=n <= 2_000_000)
        if job["want_raw"]:
            with open(os.path.join(d, "want.bin"), "wb") as f:
                f.write(want)
        env = dict(os.environ)
        env.update(VD_RCCL_LIB=build_fake(), VD_FAKE_RCCL_DIR=d, VD_FAKE_RCCL_TIMEOUT_S="120", HSA_ENABLE_IPC_MODE_LEGACY="0",
                   OMP_NUM_THREADS="4")
        procs = []
        for r in range(world):
            e = dict(env)
            j = dict(job, rank=r)
            if fail_send and r == fail_send[0]:
                e["VD_FAKE_RCCL_FAIL_SEND_AT"] = str(fail_send[1])
            if fail_send:
                j["fail_send"] = True
                if r == fail_send[2]:
                    e["VD_FAKE_RCCL_TIMEOUT_S"] = "5"      # the rank left without its message gives up long before the others do
            procs.append(subprocess.Popen(["timeout", str(timeout), sys.executable, WORKER, json.dumps(j)], stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE, text=True, env=e))
        res = {}
        for r, p in enumerate(procs):
            try:
                out, err = p.communicate(timeout=timeout + 30)
            except subprocess.TimeoutExpired:
                p.kill()
                out, err = p.communicate()
            line = [ln for ln in out.splitlines() if ln.startswith("RANK_RESULT ")]
            assert line, f"rank {r} gave no result (rc {p.returncode}):\n{out[-1500:]}\n{err[-1500:]}"
            res[r] = json.loads(line[-1][len("RANK_RESULT "):])
        return res, want_n
    finally:
        shutil.rmtree(d, ignore_errors=True)


def check(res, world, want_n, n):
    from voidin_amd import dist as vdist
    for r in range(world):
        x = res[r]
        assert x.get("ok"), f"rank {r}: {x.get('bad')} {x.get('error', '')}"
        assert x["rccl"][0] == 99901 and "libfake_rccl" in x["rccl"][1], x["rccl"]        # the double was bound, by the library itself
        lo, hi = vdist.shard_range(n, r, world)
        assert x["shard"] == [lo, hi, vdist.shard_size(n, world), hi - lo]
        st = x["fake_stats"]
        # per rank: set_scene's table all-gather + 2 x (mask all-gather) + 4 x (count all-gather) = 7 collectives, and the
        # 4 grouped exchanges of the draws / indices steps; each collective is (world - 1) sends and receives
        assert st["allgathers"] == 7 and st["groups"] == 4, st
        assert st["recvs"] >= 7 * (world - 1) and st["sends"] >= 7 * (world - 1), st
    if want_n:
        total_p2p = sum(res[r]["fake_stats"]["bytes_received"] for r in range(world))
        assert total_p2p >= (world - 1) * want_n * (20 + 4) * 2, "the grouped exchanges carried less than the lists"


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,n_mesh,hidden", [
    (2, 100_003, 16, None),          # unequal last shard
    (3, 100_003, 16, 1),             # the middle rank has zero survivors
    (8, 100_003, 16, 5),             # S = 12 501, last shard 12 496; rank 5 sends nothing
    (8, 5, 16, None),                # S = 1: ranks 5..7 own no instance at all
    (3, 50_001, 300, None),          # 2-byte rows in the replicated instance -> mesh table
    (8, 12, 70_000, None),           # 4-byte rows; ranks 6, 7 empty (S = 2)
])
def test_world_gt_1_steps_equal_single_gpu_compaction(ctx, world, n, n_mesh, hidden):
    res, want_n = run_ranks(ctx, world, n, n_mesh=n_mesh, hidden_rank=hidden)
    check(res, world, want_n, n)
    if n >= 1000:
        assert 0 < want_n < n


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_world_gt_1_at_ten_million_instances(ctx, world):
    """BASELINE configs[3]'s scene (10 M instances, its distribution: 95.6 % visible -> 191 MB of commands) over
    2 and 8 ranks: full, draws and indices steps on every rank == the one-GPU list (count + BLAKE2 of the bytes)."""
    n = 10_000_000
    res, want_n = run_ranks(ctx, world, n, kw=dict(scale_range=(0.25, 4.0)), timeout=900)
    check(res, world, want_n, n)
    assert want_n > n // 2


@pytest.mark.gpu
def test_failed_send_closes_the_group(ctx):
    """csrc/dist.hip exchange_records: rank 1's first ncclSend inside the grouped exchange (to rank 0) fails.  Rank 1
    reports VD_ERR_COMM, still serves rank 2, and its group is closed: the all-gather that follows on the same thread
    completes on every rank.  Rank 0, which never got its message, reports VD_ERR_COMM from the double's bounded wait."""
    world, n = 3, 30_011
    res, _ = run_ranks(ctx, world, n, fail_send=(1, 1, 0))
    for r in range(world):
        assert "error" not in res[r], res[r].get("error")
        assert res[r]["allgather_after_failure_ok"], f"rank {r}: the collective after the failed step did not complete"
    assert res[1]["step_failed_with_comm_error"] and res[1]["fake_stats"]["failed_sends"] == 1
    assert "group closed" in res[1]["error_text"]
    # rank 1's first send goes to rank 0 (peers in rank order): rank 0 times out on that receive, rank 2 is served
    assert res[0]["step_failed_with_comm_error"] and not res[2]["step_failed_with_comm_error"]
