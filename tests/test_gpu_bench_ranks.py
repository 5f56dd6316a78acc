"""`python bench.py --gpus 2` end to end on ONE GPU: the launcher starts two ranks (gloo rendezvous, both on
cuda:0, HIP kernels), rank 0 prints one line with n_gpus = 2, a step breakdown, every gather mode, and the
ordered draw list of the whole scene bit-equal to the N = 1 list (CRC of the list + oracle check inside bench)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=900, env=e)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-1000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("gather", ["full", "indices", "draws", "shard"])
def test_two_ranks_one_gpu_equal_single_gpu(gather):
    common = ["--instances", "3000001", "--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline"]
    one = _bench("--gpus", "1", *common)
    two = _bench("--gpus", "2", "--gather", gather, *common, env={"VOIDIN_DIST_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["scaling"] == "strong" and two["config"]["instances_total"] == 3000001
    assert two["config"]["instances_per_gpu"] == 1500001
    assert "step_breakdown" in two and two["step_breakdown"]["cull_to_mask_ms"] > 0
    assert set(two["extra"]["gather_modes"]) >= {"full", "draws", "indices", "shard"}
    assert one["config"]["verified_bit_exact_vs_oracle"] is True
    assert two["config"]["verified_bit_exact_vs_oracle"] is True
    if gather != "shard":          # shard: every rank keeps its own list (rank 0's is checked against the oracle inside bench)
        assert one["config"]["draw_list_crc32"] == two["config"]["draw_list_crc32"]
    assert two["config"]["parallelism"].startswith("instance-shard x2") and f"gather={gather}" in two["config"]["parallelism"]


def test_weak_scaling_mode_two_ranks():
    two = _bench("--gpus", "2", "--scaling", "weak", "--instances", "1000000", "--steps", "2", "--warmup", "1", "--no-extra",
                 "--no-cpu-baseline", env={"VOIDIN_DIST_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["config"]["instances_total"] == 2000000
    assert two["config"]["verified_bit_exact_vs_oracle"] is True


def test_two_ranks_report_the_second_metric_and_the_weak_run():
    """Without --no-extra an N > 1 line also carries BASELINE's second metric at N GPUs - one BLAS build per rank, replicas
    only, the same nodes on every rank - and the weak-scaled run next to the strong headline."""
    two = _bench("--gpus", "2", "--instances", "400000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--bvh-u", "128", "--bvh-v", "64",
                 env={"VOIDIN_DIST_BACKEND": "gloo"})
    rep = two["extra"]["bvh_build_replicas"]
    assert rep["ranks"] == 2 and rep["tris_per_rank"] == 2 * 128 * 64 and rep["same_nodes_on_every_rank"] is True and rep["value"] > 0
    assert two["extra"]["weak_scaling"]["instances_total"] == 800000
    assert two["value_full"] > 0 and two["value_shard"] > 0


def test_eight_ranks_default_command_finishes_in_time_and_carries_the_scaling_keys():
    """The command the driver's 8-GPU tier runs - `bench.py --gpus 8 --steps K --warmup W` (here with --no-extra) at the DEFAULT
    10 M instances - end to end on one GPU: the launcher, every rank's input generation, the C-ABI exchange (vd_dist_*, RCCL bound
    through $VD_RCCL_LIB = the tests' double, tests/cpp/fake_rccl.cpp; real RCCL refuses eight ranks on one device), all four
    gather modes, and rank 0's verification of the WHOLE gathered list against the oracle.  Wall clock < 120 s, so that an
    8-rank launch on real hardware cannot time out on what surrounds the timed steps; the line must carry what a scaling curve
    needs.  Timings through the double are not scaling numbers and are not asserted."""
    import shutil
    import tempfile
    import time
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_dist_ranks import _scratch_dir, build_fake
    d = _scratch_dir()
    try:
        env = {"VD_RCCL_LIB": build_fake(), "VD_FAKE_RCCL_DIR": d, "VD_FAKE_RCCL_TIMEOUT_S": "120", "VOIDIN_RANKS_SHARE_GPUS": "1",
               "HSA_ENABLE_IPC_MODE_LEGACY": "0", "OMP_NUM_THREADS": "4"}
        t0 = time.time()
        line = _bench("--gpus", "8", "--steps", "20", "--warmup", "5", "--no-extra", "--timeout", "300", env=env)
        wall = time.time() - t0
    finally:
        shutil.rmtree(d, ignore_errors=True)
    assert wall < 120.0, f"{wall:.0f} s"
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["steps"] == 20 and line["warmup"] == 5
    assert line["config"]["instances_total"] == 10_000_000 and line["config"]["instances_per_gpu"] == 1_250_000
    assert line["config"]["verified_bit_exact_vs_oracle"] is True
    assert line["config"]["rccl"]["version"] == 99901 and "libfake_rccl" in line["config"]["rccl"]["library"]      # the double, bound by the library itself
    assert "C ABI vd_dist_*" in line["config"]["parallelism"] and "instance-shard x8" in line["config"]["parallelism"]
    assert line["value"] > 0 and line["value_full"] > 0 and line["value_shard"] > 0
    for k in ("cull_to_mask_ms", "mask_allgather_ms", "expand_all_shards_ms", "mask_bytes_per_rank"):
        assert line["step_breakdown"][k] > 0, k
    assert set(line["extra"]["gather_modes"]) >= {"full", "draws", "indices", "shard"}
    assert line["scaling_ceiling"]["mask_allgather_ms_measured"] == line["step_breakdown"]["mask_allgather_ms"]
    # the ordered list of the whole scene is the one-GPU list (same CRC as the N = 1 line of the same workload)
    one = _bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline")
    assert one["config"]["draw_list_crc32"] == line["config"]["draw_list_crc32"]
