"""Run-time half of the fence-free hand-offs: every kernel with an inter-workgroup protocol is run many
times and every result is compared byte for byte with the first one (and the first one with the oracle elsewhere in
the suite).  A race shows up as a run that differs.  tests/test_isa_protocols.py checks the emitted code."""
import numpy as np
import pytest
import torch

from voidin_amd import dist as vdist
from voidin_amd import synth

pytestmark = pytest.mark.gpu


def test_split_cull_scan_hand_off_repeats(ctx):
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    d_m = ctx.upload(meshes)
    for n, iters in ((3_000_001, 150), (10_000_000, 60)):
        inst = synth.instances(n, seed=synth.SEED_BASE + 3, with_inverse=False)
        d_i = ctx.upload(inst)
        d_o, d_c = ctx.empty(n * 20), torch.zeros(4, dtype=torch.int32, device="cuda")
        first = None
        for it in range(iters):
            if it % 16 == 0:
                d_o.zero_()
            ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, n, d_o, d_c)
            c = int(d_c[0].item())
            if first is None:
                first = (c, d_o[: c * 20].clone())
            else:
                assert c == first[0] and torch.equal(d_o[: c * 20], first[1]), f"n={n}: run {it} differs"
        if n == 3_000_001:            # several shards through vd_expand_mask_dev: fast (shard % 4 == 0) and general path
            for shards in (4, 3):
                S, wps = vdist.shard_size(n, shards), vdist.mask_words(vdist.shard_size(n, shards))
                d_mask = torch.zeros(wps * shards, dtype=torch.int64, device="cuda")
                ids = np.zeros(S * shards, np.uint8)
                ids[:n] = inst["mesh"]
                for r in range(shards):
                    lo, hi = vdist.shard_range(n, r, shards)
                    ctx.cull_mask_dev(cam, d_m, len(meshes), ctx.upload(inst[lo:hi]), hi - lo, d_mask[r * wps:])
                d_ids = ctx.upload(ids)
                for it in range(60):
                    ctx.expand_mask_dev(d_mask, n, S, d_ids, d_m, len(meshes), d_o, d_c, id_bytes=1)
                    c = int(d_c[0].item())
                    assert c == first[0] and torch.equal(d_o[: c * 20], first[1]), f"expand shards={shards}: run {it} differs"
        del d_i, d_o


def test_tlas_refit_climb_repeats(ctx):
    meshes = synth.mesh_infos()
    d_m = ctx.upload(meshes)
    for n, wide in ((32768, False), (65536, True)):
        inst = synth.instances(n, seed=synth.SEED_BASE + 6, extent=300.0)
        d_i, d_t = ctx.upload(inst), ctx.empty((2 * n + 1) * (48 if wide else 32))
        ctx.tlas_build_dev(d_i, n, d_m, len(meshes), d_t, wide=wide)
        torch.cuda.synchronize()
        built = d_t.clone()
        for it in range(200):
            ctx.tlas_refit_dev(d_i, n, d_m, len(meshes), d_t, wide=wide)
            assert torch.equal(d_t, built), f"refit n={n}: run {it} differs from the build"


def test_several_workgroup_tlas_chain_repeats(ctx):
    """The 16-workgroup chain (sizes above what the indexed build takes, or VD_OPT_TLAS_INDEX = 0): tagged-word exchange."""
    ctx.set_option("tlas.index", 0)
    try:
        meshes = synth.mesh_infos()
        inst = synth.instances(16461, seed=synth.SEED_BASE + 14, extent=700.0)
        first = ctx.tlas_build(inst, meshes).tobytes()
        for it in range(6):
            assert ctx.tlas_build(inst, meshes).tobytes() == first, f"run {it} differs"
    finally:
        ctx.set_option("tlas.index", None)


def test_indexed_tlas_build_repeats(ctx):
    meshes = synth.mesh_infos()
    inst = synth.instances(20000, seed=synth.SEED_BASE + 14, extent=500.0)
    first = ctx.tlas_build(inst, meshes).tobytes()
    for it in range(4):
        assert ctx.tlas_build(inst, meshes).tobytes() == first, f"run {it} differs"
