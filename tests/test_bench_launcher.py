"""bench.py --gpus N starts N ranks by itself: the launcher and the rendezvous run here on CPU
(gloo); the same command with the HIP kernels is tests/test_gpu_bench_ranks.py."""
import inspect
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line expected, got {len(lines)}: {p.stdout[-500:]}"
    return json.loads(lines[0])


def test_gpus_flag_spawns_that_many_ranks():
    for n in (2, 3):
        line = _run("--gpus", str(n), "--launch-check", env={"VOIDIN_DIST_BACKEND": "gloo"})
        assert line["n_gpus"] == n and line["ranks"] == list(range(n))


def test_one_rank_needs_no_process_group():
    assert _run("--launch-check")["n_gpus"] == 1


def test_under_torchrun_the_process_is_a_rank_not_a_launcher():
    # WORLD_SIZE already set (as torch.distributed.run does): bench.py must not spawn again
    line = _run("--gpus", "1", "--launch-check", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert line["n_gpus"] == 1


def test_launcher_touches_neither_torch_nor_hip():
    import bench
    src = inspect.getsource(bench.launch_ranks) + inspect.getsource(bench.main)
    assert "torch" not in src and "hip" not in src.lower().replace("ship", "")


def test_failing_rank_fails_the_launch():
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e["VOIDIN_DIST_BACKEND"] = "no_such_backend"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                       timeout=300, env=e)
    assert p.returncode != 0


def _rank_pids(launcher_pid):
    out = subprocess.run(["ps", "-o", "pid=", "--ppid", str(launcher_pid)], capture_output=True, text=True).stdout.split()
    return [int(x) for x in out]


def test_a_stopped_launcher_takes_its_ranks_with_it():
    """SIGTERM to the launcher (what `timeout 600 python bench.py --gpus 2` sends) must not orphan the ranks,
    and --timeout bounds a launch whose ranks never finish."""
    import signal
    import time
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e["VOIDIN_DIST_BACKEND"] = "gloo"
    e["VOIDIN_LAUNCH_CHECK_HOLD_S"] = "60"            # the ranks rendezvous and then sit there
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"]
    p = subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    kids = []
    for _ in range(100):
        kids = _rank_pids(p.pid)
        if len(kids) == 2:
            break
        time.sleep(0.1)
    assert len(kids) == 2
    time.sleep(1.0)
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) == 128 + signal.SIGTERM
    time.sleep(0.2)
    for k in kids:
        assert not os.path.exists(f"/proc/{k}") or open(f"/proc/{k}/stat").read().split()[2] == "Z", f"rank process {k} survived its launcher"
    t0 = time.time()
    q = subprocess.run(cmd + ["--timeout", "6"], env=e, capture_output=True, text=True, timeout=120)
    assert q.returncode == 124 and time.time() - t0 < 60 and "still running" in q.stderr


def test_tree_shape_counts_every_partitioned_primitive(oracle):
    import bench
    from voidin_amd import synth
    v, i = synth.knot_mesh(24, 8)
    nodes, _ = oracle.bvh_build(v, i)
    # brute force: walk the tree recursively
    def walk(k, d):
        if nodes["count"][k] > 0:
            return int(nodes["count"][k]), 0, d
        l = int(nodes["left_first"][k])
        a, sa, da = walk(l, d + 1)
        b, sb, db = walk(l + 1, d + 1)
        return a + b, sa + sb + a + b, max(da, db)
    total, active, depth = walk(0, 0)
    got = bench.tree_shape(nodes)
    assert total == len(i) // 3
    assert got["sum_active_prims"] == active and got["depth"] == depth
    assert got["interior_nodes"] == int((nodes["count"] == 0).sum()) - 1
