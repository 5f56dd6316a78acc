"""Parity at BASELINE.json's full sizes, against the oracle, inside the driver-run suite:

  * configs[4] "8M tris SAH BVH build": the 2048 x 2048 knot mesh (8 388 608 triangles) - the mesh bench.py times;
  * TLAS build at exactly 32 768 instances (the reference layout's limit, tlas.rs:71) and at 65 536 in the wide layout;
  * 65 536-instance refit after motion against the oracle's refit on the GPU-built topology.

The three oracle builds (~45 s, ~30 s, ~2-3 min on one core each) start together on host threads when the module is
first used (ctypes releases the GIL) and are joined by the tests that need them, so the module costs the longest of
them, not their sum."""
import concurrent.futures
import time

import numpy as np
import pytest

from conftest import fields_equal, first_difference
from voidin_amd import abi, synth

pytestmark = pytest.mark.gpu

N_TLAS, N_WIDE = 32768, 65536


@pytest.fixture(scope="module")
def jobs(oracle):
    meshes = synth.mesh_infos()
    v, i = synth.knot_mesh(2048, 2048)
    inst = synth.instances(N_TLAS, seed=synth.SEED_BASE + 6, extent=300.0)      # bench.py's TLAS scene
    winst = synth.instances(N_WIDE, seed=synth.SEED_BASE + 7, extent=400.0)     # bench.py's wide scene
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=3)

    def timed(fn, *a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        return r, time.perf_counter() - t
    j = {"meshes": meshes, "mesh": (v, i), "inst": inst, "winst": winst,
         "wide": pool.submit(timed, oracle.tlas_build, winst, meshes, wide=True),     # the longest first
         "blas": pool.submit(timed, oracle.bvh_build, v, i),
         "tlas": pool.submit(timed, oracle.tlas_build, inst, meshes)}
    yield j
    pool.shutdown(wait=True)


def test_blas_8m_triangles_bit_exact(ctx, jobs):
    import torch
    v, i = jobs["mesh"]
    n_tri = len(i) // 3
    assert n_tri == 8_388_608
    d_v, d_i, d_n = ctx.upload(v), ctx.upload(i), ctx.empty(2 * n_tri * 32)
    n_nodes = ctx.bvh_build_dev(d_v, len(v), d_i, n_tri, d_n, 2 * n_tri)
    torch.cuda.synchronize()
    nodes = d_n.cpu().numpy()[: n_nodes * 32].view(abi.BVH_NODE)
    idx = d_i.cpu().numpy().view(np.uint32)[: 3 * n_tri]
    (wn, wi), t_cpu = jobs["blas"].result(timeout=1200)
    assert len(nodes) == len(wn), f"node count {len(nodes)} != {len(wn)}"
    assert nodes.tobytes() == wn.tobytes(), first_difference(nodes, wn)
    assert np.array_equal(idx, wi)
    # size-independent properties of the result: the permutation is a permutation; leaves cover [0, T) once, in order
    tri_in = np.sort(i.reshape(-1, 3).view([("a", "<u4"), ("b", "<u4"), ("c", "<u4")]).reshape(-1), order=("a", "b", "c"))
    tri_out = np.sort(idx.reshape(-1, 3).view([("a", "<u4"), ("b", "<u4"), ("c", "<u4")]).reshape(-1), order=("a", "b", "c"))
    assert np.array_equal(tri_in, tri_out)
    leaves = nodes[nodes["count"] > 0]
    order = np.argsort(leaves["left_first"], kind="stable")
    lf, cnt = leaves["left_first"][order].astype(np.int64), leaves["count"][order].astype(np.int64)
    assert lf[0] == 0 and np.array_equal(lf[1:], (lf + cnt)[:-1]) and lf[-1] + cnt[-1] == n_tri and cnt.max() <= 3
    print(f"oracle BLAS {t_cpu:.1f} s")


def test_tlas_32768_build_bit_exact(ctx, jobs):
    got = ctx.tlas_build(jobs["inst"], jobs["meshes"])
    want, t_cpu = jobs["tlas"].result(timeout=1200)
    assert fields_equal(got, want), first_difference(got, want)
    assert ctx.tlas_refit(jobs["inst"], jobs["meshes"], got).tobytes() == got.tobytes()     # T3: refit(build(x)) == build(x)
    print(f"oracle TLAS {t_cpu:.1f} s")


def test_tlas_wide_65536_refit_after_motion_vs_oracle(ctx, oracle, jobs):
    """GPU build (wide layout) -> 10 % of the instances move (compute_update) -> vd_tlas_refit_wide_dev against the
    oracle's refit of the SAME topology; O(N) on the CPU, so it does not wait for the oracle's build."""
    import torch
    meshes, winst = jobs["meshes"], jobs["winst"]
    d_m, d_i = ctx.upload(meshes), ctx.upload(winst)
    d_w = ctx.empty((2 * N_WIDE + 1) * 48)
    ctx.tlas_build_dev(d_i, N_WIDE, d_m, len(meshes), d_w, wide=True)
    torch.cuda.synchronize()
    topo = d_w.cpu().numpy()[: (2 * N_WIDE + 1) * 48].view(abi.TLAS_NODE_WIDE).copy()
    moved = oracle.compute_update(np.arange(0, N_WIDE, 10, dtype=np.uint32), winst, 1.0, 0.016)
    d_mv = ctx.upload(moved)
    ctx.tlas_refit_dev(d_mv, N_WIDE, d_m, len(meshes), d_w, wide=True)
    torch.cuda.synchronize()
    got = d_w.cpu().numpy()[: (2 * N_WIDE + 1) * 48].view(abi.TLAS_NODE_WIDE)
    want = oracle.tlas_refit(moved, meshes, topo)
    assert got.tobytes() == want.tobytes(), first_difference(got, want)
    assert not np.array_equal(got["min"], topo["min"])                      # something did move
    # union property: every interior box is the union of its children's boxes
    k = np.arange(N_WIDE + 1, 2 * N_WIDE + 1)
    l, r = got["left"][k], got["right"][k]
    assert np.array_equal(got["min"][k], np.minimum(got["min"][l], got["min"][r]))
    assert np.array_equal(got["max"][k], np.maximum(got["max"][l], got["max"][r]))


def test_tlas_wide_65536_build_bit_exact(ctx, jobs):
    got = ctx.tlas_build(jobs["winst"], jobs["meshes"], wide=True)
    want, t_cpu = jobs["wide"].result(timeout=1500)
    assert got.tobytes() == want.tobytes(), first_difference(got, want)
    print(f"oracle wide TLAS {t_cpu:.1f} s")
