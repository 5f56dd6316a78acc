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
