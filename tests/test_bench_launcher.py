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


def test_second_metric_is_promoted_into_the_keys_a_reader_keeps():
    """BASELINE's metric names two numbers; the BVH build, the TLAS refit and the padded (consumer) form of the
    headline are copied from `extra` into `roofline` / `cpu_baseline` (the objects the driver's record keeps)."""
    import bench
    line = {"roofline": {"bound": "hbm"}, "cpu_baseline": {"value": 1.0}}
    extra = {
        "bvh_build": {"n_tris": 8_388_608, "ms": 25.0, "value": 335.5, "bit_exact_vs_oracle": True,
                      "phases_ms": {"ms_phase_a": 16.0, "ms_phase_b": 7.5, "kernel_launches": 900},
                      "roofline": {"emulating_770B_per_prim_level": {"frac": 0.72}, "binned_44B_per_prim_level": {"frac": 0.046}},
                      "cpu_baseline": {"value": 0.108, "cores": 1, "kind": "port", "sample": "the timed mesh"}},
        "tlas": {"refit_queued_ms": 0.09, "build_ms": 185.0, "bit_exact_vs_oracle": True, "cpu_baseline": {"value": 29000.0}},
        "tlas_wide_64k": {"refit_gpu_ms": 0.095, "refit_after_motion_bit_exact_vs_oracle": True},
        "cull_compact_pad_tail": {k: {"ms": 0.28, "M_inst_per_s": 35000.0, "visible_fraction": 0.9, "tail_bytes": 1,
                                      "whole_padded_buffer_bit_exact_vs_oracle": True} for k in ("baseline", "dist_small")},
    }
    bench.promote_second_metric(line, extra)
    b = line["roofline"]["bvh_build"]
    for k in ("n_tris", "ms", "Mprims_per_s", "traffic_bytes", "frac_real", "frac_770B", "frac_44B", "bit_exact_vs_oracle", "kernel_launches"):
        assert k in b, k
    pmc = json.load(open(bench.latest_pmc("bvh")))
    assert b["traffic_bytes"] == int(pmc["total"]["sum_MB"] * 1e6)
    assert abs(b["frac_real"] - b["traffic_bytes"] / 25.0e-3 / 8e12) < 1e-3
    assert line["cpu_baseline"]["bvh_build"] == {"Mprims_per_s": 0.108, "cores": 1, "kind": "port", "sample": "the timed mesh"}
    assert line["roofline"]["tlas_refit_ms"]["32768"] == 0.09 and line["roofline"]["tlas_refit_ms"]["65536_wide"] == 0.095
    assert set(line["roofline"]["pad_tail"]) == {"baseline", "dist_small"}
    # a line without the extras (--no-extra, N > 1) is left alone
    bare = {"roofline": {}, "cpu_baseline": None}
    bench.promote_second_metric(bare, {})
    assert bare == {"roofline": {}, "cpu_baseline": None}
