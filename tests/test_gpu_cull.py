"""Parity of the HIP cull / emit / compaction path against the CPU oracle, through the C ABI.
Bit-exact (integer/index output): memcmp of the command buffers."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden
from voidin_amd import abi, synth

pytestmark = pytest.mark.gpu

CASES = ["cull_model_wide.npz", "cull_model_small.npz", "cull_jitter_wide.npz", "cull_jitter_small.npz",
         "cull_model_scene.npz", "cull_model_scene_nave.npz", "cull_model_scene_x3.npz"]   # the last three: the reference's own demo scene (make_golden.model_scene_cases)


def run_dev(ctx, cam, meshes, inst, pad_tail=False):
    import torch
    n = len(inst)
    d_m, d_i = ctx.upload(meshes), ctx.upload(inst)
    d_emit = ctx.empty(n * 20)
    d_comp = ctx.empty(n * 20)
    d_comp.fill_(0xAB)
    d_cnt = torch.zeros(4, dtype=torch.int32, device=ctx.torch_device)
    ctx.cull_emit_dev(cam, d_m, len(meshes), d_i, n, d_emit)
    ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, n, d_comp, d_cnt, pad_tail)
    torch.cuda.synchronize()
    emit = d_emit.cpu().numpy()[: n * 20].view(abi.DRAW)
    cnt = int(d_cnt[0].item()) & 0xFFFFFFFF
    comp = d_comp.cpu().numpy()[: n * 20].view(abi.DRAW)
    return emit, comp, cnt


@pytest.mark.parametrize("name", CASES)
def test_golden_fixtures(ctx, ctx_options, name):
    g = golden(name)
    emit, comp, cnt = run_dev(ctx, g["camera"], g["meshes"], g["instances"])
    assert emit.tobytes() == g["draws"].tobytes()
    assert cnt == int(g["count"])
    assert comp[:cnt].tobytes() == g["compact"].tobytes()
    # the split form (what inputs of >= 2 Mi instances run): bit mask + per-instance mesh id table - 1-byte ids up to 256
    # meshes, 2-byte ids beyond (cull_model_scene_x3 has 327) - + scan + expansion
    ctx_options("cull.split_min", 1)
    emit, comp, cnt = run_dev(ctx, g["camera"], g["meshes"], g["instances"])
    ctx_options("cull.split_min", None)
    assert emit.tobytes() == g["draws"].tobytes()
    assert cnt == int(g["count"]) and comp[:cnt].tobytes() == g["compact"].tobytes()
    # host-pointer entry points give the same bytes
    assert ctx.cull_emit(g["camera"], g["meshes"], g["instances"]).tobytes() == g["draws"].tobytes()
    c2, n2 = ctx.cull_compact(g["camera"], g["meshes"], g["instances"])
    assert n2 == cnt and c2[:n2].tobytes() == g["compact"].tobytes()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4097, 100_000])
@pytest.mark.parametrize("dist", ["wide", "small"])
def test_ragged_sizes_bit_exact(ctx, oracle, n, dist):
    # BASELINE config 2 (100k) and every tile-boundary size around 64 / 256 / 2048
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    kw = dict(scale_range=(0.25, 4.0)) if dist == "wide" else dict(scale_range=(0.01, 0.3), extent=400.0)
    inst = synth.instances(n, seed=synth.SEED_BASE + 2, **kw)
    want = oracle.cull_emit(cam, meshes, inst)
    emit, comp, cnt = run_dev(ctx, cam, meshes, inst, pad_tail=True)
    assert emit.tobytes() == want.tobytes()
    wc, wn = oracle.compact(want, pad_tail=True)
    assert cnt == wn and comp.tobytes() == wc.tobytes()


CLOUDS = {"wide": dict(scale_range=(0.25, 4.0)), "small": dict(scale_range=(0.02, 0.6), extent=600.0),
          "mid": dict(scale_range=(0.5, 1.2), extent=1500.0)}


@pytest.mark.parametrize("znear", [0.001, 1.0, 100.0])
@pytest.mark.parametrize("zfar", [50.0, 500.0, 5.0, -10.0])
@pytest.mark.parametrize("cloud", ["wide", "small", "mid"])
def test_finite_far_plane_takes_the_third_return_of_is_visible(ctx, ctx_options, oracle, cloud, zfar, znear):
    """`is_visible`'s third test, `c.z + r > znear && c.z - r > zfar` (shaders/emit_draws.wgsl:28-30), never fires under the
    reference's own camera (zfar = +inf, crates/components/src/camera.rs:22-23,38) - but the 320-byte uniform is an INPUT and a
    caller can hand over a finite far plane.  Emit + compact, fused and split form, memcmp against the oracle; on the clouds
    whose scales let an instance behind the camera pass the two side-plane tests (max scale in ~[0.78, 1)) the far test must
    change the survivor set, i.e. the branch is TAKEN under test."""
    cam_inf, meshes = synth.camera_uniform(), synth.mesh_infos()
    cam = cam_inf.copy()
    cam["zfar"], cam["znear"] = np.float32(zfar), np.float32(znear)
    n = 300_000
    inst = synth.instances(n, seed=synth.SEED_BASE + 2, with_inverse=False, **CLOUDS[cloud])
    want = oracle.cull_emit(cam, meshes, inst, threads=8)
    want_inf = oracle.cull_emit(cam_inf, meshes, inst, threads=8)
    culled_by_far = int(want_inf["instance_count"].sum()) - int(want["instance_count"].sum())
    assert culled_by_far >= 0
    if cloud in ("wide", "mid") and zfar <= 50.0:
        assert culled_by_far > 0, "the far-plane return is not exercised by this case"
    wc, wn = oracle.compact(want, pad_tail=True)
    for split in (False, True):
        if split:
            ctx_options("cull.split_min", 1)
        emit, comp, cnt = run_dev(ctx, cam, meshes, inst, pad_tail=True)
        assert emit.tobytes() == want.tobytes(), f"split={split}"
        assert cnt == wn and comp.tobytes() == wc.tobytes(), f"split={split}"
    ctx_options("cull.split_min", None)


def test_far_plane_on_a_cloud_behind_the_camera(ctx, ctx_options, oracle):
    """Every instance behind the camera, uniform scale 0.85, so that most pass the side planes and the far test alone decides:
    all three returns of is_visible are taken in one launch (emit_draws.wgsl:22-32)."""
    cam_inf, meshes = synth.camera_uniform(eye=(0.0, 0.0, 0.0), pitch_deg=0.0), synth.mesh_infos()
    n = 100_000
    inst = synth.instances(n, seed=synth.SEED_BASE + 70, scale_range=(0.85, 0.85), extent=40.0, with_inverse=False)
    rng = np.random.default_rng(70)
    t = inst["transform"].reshape(n, 16)
    t[:, 12] = rng.uniform(-30, 30, n).astype(np.float32)        # view x
    t[:, 13] = rng.uniform(-30, 30, n).astype(np.float32)
    t[:, 14] = rng.uniform(-100, 900, n).astype(np.float32)      # +z = behind an unrotated camera at the origin (RH view space)
    for zfar in (20.0, 60.0):
        cam = cam_inf.copy()
        cam["zfar"] = np.float32(zfar)
        want, want_inf = oracle.cull_emit(cam, meshes, inst, threads=8), oracle.cull_emit(cam_inf, meshes, inst, threads=8)
        by_far = int(want_inf["instance_count"].sum()) - int(want["instance_count"].sum())
        assert by_far > n // 50, by_far                            # the far test culls thousands ...
        assert 0 < int(want["instance_count"].sum()) < n - by_far  # ... and the side planes and `visible` are taken too
        wc, wn = oracle.compact(want)
        for split in (False, True):
            if split:
                ctx_options("cull.split_min", 1)
            emit, comp, cnt = run_dev(ctx, cam, meshes, inst)
            assert emit.tobytes() == want.tobytes() and cnt == wn and comp[:cnt].tobytes() == wc[:wn].tobytes(), (zfar, split)
        ctx_options("cull.split_min", None)


def test_empty_input(ctx):
    import torch
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    d_m = ctx.upload(meshes)
    d_cnt = torch.full((4,), 7, dtype=torch.int32, device=ctx.torch_device)
    ctx.cull_emit_dev(cam, d_m, len(meshes), None, 0, None)
    ctx.cull_compact_dev(cam, d_m, len(meshes), None, 0, None, d_cnt)
    torch.cuda.synchronize()
    assert int(d_cnt[0].item()) == 0
    out, cnt = ctx.cull_compact(cam, meshes, np.zeros(0, abi.INSTANCE))
    assert cnt == 0 and len(out) == 0


def test_invalid_arguments_return_codes(ctx):
    cam = synth.camera_uniform()
    lib, h = ctx.lib, ctx.h
    assert lib.vd_cull_emit_dev(h, None, None, 0, None, 5, None) == abi.VD_ERR_INVALID_ARG
    assert b"vd_cull_emit" in lib.vd_last_error(h)
    assert lib.vd_cull_compact_dev(h, cam.ctypes.data, None, 0, None, 5, None, None, 0) == abi.VD_ERR_INVALID_ARG


def test_nan_and_out_of_range_mesh(ctx, oracle):
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    inst = synth.instances(300, scale_range=(0.01, 0.3), extent=400.0)
    inst["mesh"][3] = 1000
    inst["mesh"][77] = 0xFFFFFFFF
    inst["transform"][5] = np.nan
    inst["transform"][9][12] = np.inf
    inst["transform"][11] = 0.0
    want = oracle.cull_emit(cam, meshes, inst)
    emit, comp, cnt = run_dev(ctx, cam, meshes, inst)
    assert emit.tobytes() == want.tobytes()
    wc, wn = oracle.compact(want)
    assert cnt == wn and comp[:cnt].tobytes() == wc[:wn].tobytes()


def test_compact_is_filter_of_emit_and_standalone_compaction(ctx):
    """C3 is a pure function of C1's output: fused == vd_compact_draws_dev(emit)."""
    import torch
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    n = 300_001
    inst = synth.instances(n, scale_range=(0.02, 0.6), extent=600.0)
    emit, comp, cnt = run_dev(ctx, cam, meshes, inst)
    keep = emit["instance_count"] == 1
    assert cnt == keep.sum() and comp[:cnt].tobytes() == emit[keep].tobytes()
    d_in = ctx.upload(emit)
    d_out = ctx.empty(n * 20)
    d_cnt = torch.zeros(4, dtype=torch.int32, device=ctx.torch_device)
    ctx.compact_draws_dev(d_in, n, d_out, d_cnt)
    torch.cuda.synchronize()
    assert int(d_cnt[0].item()) == cnt
    assert d_out.cpu().numpy()[: cnt * 20].tobytes() == comp[:cnt].tobytes()


def test_deterministic_across_runs(ctx):
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    inst = synth.instances(500_000, scale_range=(0.02, 0.6), extent=600.0)
    a = run_dev(ctx, cam, meshes, inst)
    b = run_dev(ctx, cam, meshes, inst)
    assert a[2] == b[2] and a[0].tobytes() == b[0].tobytes() and a[1][: a[2]].tobytes() == b[1][: b[2]].tobytes()


def test_full_size_10m_properties(ctx, oracle):
    """BASELINE config 3: 10M instances.  The oracle finishes this in seconds, so the survivor
    set is compared bit-exact, plus the size-independent properties (sorted survivor indices,
    compact == filter(emit), count consistency)."""
    import torch
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    n = 10_000_000
    inst = synth.instances(n, seed=synth.SEED_BASE + 3, with_inverse=False)
    d_m, d_i = ctx.upload(meshes), ctx.upload(inst)
    d_emit, d_comp = ctx.empty(n * 20), ctx.empty(n * 20)
    d_cnt = torch.zeros(4, dtype=torch.int32, device=ctx.torch_device)
    ctx.cull_emit_dev(cam, d_m, len(meshes), d_i, n, d_emit)
    ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, n, d_comp, d_cnt)
    torch.cuda.synchronize()
    cnt = int(d_cnt[0].item())
    emit = d_emit.cpu().numpy()[: n * 20].view(abi.DRAW)
    comp = d_comp.cpu().numpy()[: cnt * 20].view(abi.DRAW)
    keep = emit["instance_count"] == 1
    assert cnt == keep.sum()
    assert (np.diff(comp["base_instance"].astype(np.int64)) > 0).all()
    assert comp.tobytes() == emit[keep].tobytes()
    want = oracle.cull_emit(cam, meshes, inst, threads=8)
    assert emit.tobytes() == want.tobytes()


@pytest.mark.parametrize("id_bytes", [1, 2, 4])
@pytest.mark.parametrize("n,shards", [(1, 1), (100_000, 1), (100_001, 3), (1_000_000, 8), (130, 2),
                                      (262_221, 4),      # shard 65556: multiple of 4, 1025 mask words (groups straddle shards)
                                      (1_000, 5),        # 4 mask words per shard
                                      (13_000_000, 4)])  # > 12 Mi instances: LDS-staged form of the 1-byte-id kernel
def test_mask_wire_format_equals_fused_compaction(ctx, oracle, n, shards, id_bytes):
    """Multi-GPU wire format on one GPU: cull every shard into its bitmask, concatenate the
    shard masks as the all-gather would, expand -> identical to the fused single-pass list."""
    import torch
    from voidin_amd import dist as vdist
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    inst = synth.instances(n, seed=synth.SEED_BASE + 4, scale_range=(0.02, 0.6), extent=600.0, with_inverse=False)
    if n > 2_000_000 and id_bytes != 1:
        pytest.skip("large case covers the 1-byte-id kernel only")
    want, wn = oracle.compact(oracle.cull_emit(cam, meshes, inst, threads=8))
    d_m = ctx.upload(meshes)
    S = vdist.shard_size(n, shards)
    wps = vdist.mask_words(S)
    d_mask_all = torch.zeros(wps * shards, dtype=torch.int64, device="cuda")
    ids = np.zeros(S * shards, {1: np.uint8, 2: np.uint16, 4: np.uint32}[id_bytes])
    ids[:n] = inst["mesh"]
    for r in range(shards):
        lo, hi = vdist.shard_range(n, r, shards)
        if hi > lo:
            d_i = ctx.upload(inst[lo:hi])
            ctx.cull_mask_dev(cam, d_m, len(meshes), d_i, hi - lo, d_mask_all[r * wps:])
    d_ids = ctx.upload(ids)
    d_out = ctx.empty(n * 20)
    d_cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
    ctx.expand_mask_dev(d_mask_all, n, S, d_ids, d_m, len(meshes), d_out, d_cnt, id_bytes=id_bytes)
    torch.cuda.synchronize()
    cnt = int(d_cnt[0].item())
    assert cnt == wn
    assert d_out.cpu().numpy()[: cnt * 20].tobytes() == want[:wn].tobytes()


def test_large_mesh_table(ctx, oracle):
    """700 MeshInfo records: exercises the non-LDS mesh-table path of vd_expand_mask_dev and the
    gathers of the cull kernels."""
    import torch
    from voidin_amd import dist as vdist
    cam = synth.camera_uniform()
    meshes = synth.mesh_infos(700, seed=synth.SEED_BASE + 40)
    n = 150_000
    inst = synth.instances(n, n_mesh=700, seed=synth.SEED_BASE + 41, scale_range=(0.02, 0.6), extent=600.0, with_inverse=False)
    want = oracle.cull_emit(cam, meshes, inst)
    wc, wn = oracle.compact(want)
    emit, comp, cnt = run_dev(ctx, cam, meshes, inst)
    assert emit.tobytes() == want.tobytes() and cnt == wn and comp[:cnt].tobytes() == wc[:wn].tobytes()
    d_m, d_i = ctx.upload(meshes), ctx.upload(inst)
    sv = vdist.ShardedVisibility(ctx, n, d_m, len(meshes), d_i)
    d_out, d_cnt = ctx.empty(n * 20), torch.zeros(4, dtype=torch.int32, device="cuda")
    sv.step(cam, d_out, d_cnt)
    torch.cuda.synchronize()
    assert int(d_cnt[0].item()) == wn and d_out.cpu().numpy()[: wn * 20].tobytes() == wc[:wn].tobytes()


@pytest.mark.parametrize("n_mesh", [1, 257, 600, 66_000])
def test_split_forms_with_every_id_width_and_table_size(ctx, oracle, n_mesh):
    """From abi.CULL_SPLIT_MIN (2 Mi) instances on both vd_cull_emit and vd_cull_compact go through pass 1's bits + ids: 1-byte ids
    (<= 256 meshes), 2-byte ids with the LDS table (<= 512) and without, 4-byte ids (> 65536 meshes)."""
    cam = synth.camera_uniform()
    meshes = synth.mesh_infos(n_mesh, seed=synth.SEED_BASE + 50)
    if n_mesh > 60_000:                                   # base_index would overflow u32 with the default index counts
        meshes["index_count"] = 36
        meshes["base_index"] = np.arange(n_mesh, dtype=np.uint32) * 36
        meshes["vertex_offset"] = np.arange(n_mesh, dtype=np.int32) * 12
    n = abi.CULL_SPLIT_MIN + 12_345
    inst = synth.instances(n, n_mesh=n_mesh, seed=synth.SEED_BASE + 51, scale_range=(0.02, 0.6), extent=600.0, with_inverse=False)
    want = oracle.cull_emit(cam, meshes, inst, threads=8)
    wc, wn = oracle.compact(want)
    emit, comp, cnt = run_dev(ctx, cam, meshes, inst)
    assert emit.tobytes() == want.tobytes()
    assert cnt == wn and comp[:cnt].tobytes() == wc[:wn].tobytes()


def test_id_table_follows_edited_instances(ctx, oracle):
    """Pass 1 keeps the instance->mesh table between calls and rewrites only rows that changed: alternate inputs whose
    mesh ids differ everywhere / in a few rows / not at all (only transforms), and sizes that move the table."""
    import torch
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    n = abi.CULL_SPLIT_MIN + 999
    a = synth.instances(n, seed=synth.SEED_BASE + 60, scale_range=(0.02, 0.6), extent=600.0, with_inverse=False)
    b = a.copy(); b["mesh"] = (b["mesh"] + 5) % len(meshes)                      # every row differs
    c = a.copy(); c["mesh"][[3, 1024, 500_000, n - 1]] = [7, 7, 0, 11]           # four rows differ
    d = synth.instances(n, seed=synth.SEED_BASE + 61, scale_range=(0.02, 0.6), extent=600.0, with_inverse=False)
    d["mesh"] = a["mesh"]                                                        # other transforms, same meshes
    small = a[: abi.CULL_SPLIT_MIN + 1]
    d_m = ctx.upload(meshes)
    d_out, d_cnt = ctx.empty(n * 20), torch.zeros(4, dtype=torch.int32, device="cuda")
    for name, inst in [("a", a), ("b", b), ("a", a), ("c", c), ("d", d), ("small", small), ("a", a)]:
        want, wn = oracle.compact(oracle.cull_emit(cam, meshes, inst, threads=8))
        ctx.cull_compact_dev(cam, d_m, len(meshes), ctx.upload(inst), len(inst), d_out, d_cnt)
        torch.cuda.synchronize()
        assert int(d_cnt[0].item()) == wn, name
        assert d_out.cpu().numpy()[: wn * 20].tobytes() == want[:wn].tobytes(), name


@pytest.mark.parametrize("n", [abi.CULL_SPLIT_MIN - 1, abi.CULL_SPLIT_MIN, abi.CULL_SPLIT_MIN + 77, 3_000_001])
def test_split_and_fused_forms_agree_around_the_switch(ctx, oracle, n):
    """vd_cull_compact runs the fused kernel below abi.CULL_SPLIT_MIN (2 Mi) instances and the split form (bitmask +
    expansion) from there on: both must equal the oracle, including pad_tail and a shard offset."""
    import torch
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    inst = synth.instances(n, seed=synth.SEED_BASE + 13, scale_range=(0.02, 0.6), extent=600.0, with_inverse=False)
    want = oracle.cull_emit(cam, meshes, inst)
    want["base_instance"] += np.uint32(12345)
    wc, wn = oracle.compact(want, pad_tail=True)
    d_m, d_i = ctx.upload(meshes), ctx.upload(inst)
    d_out = ctx.empty(n * 20)
    d_out.fill_(0xCD)
    d_cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
    ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, n, d_out, d_cnt, True, 12345)
    torch.cuda.synchronize()
    assert int(d_cnt[0].item()) == wn
    assert d_out.cpu().numpy()[: n * 20].tobytes() == wc.tobytes()
    # vd_cull_emit switches at the same size (mask pass + emit_from_mask_kernel); ragged n exercises the tail quad
    d_emit = ctx.empty(n * 20 + 64)
    d_emit.fill_(0xEE)
    ctx.cull_emit_dev(cam, d_m, len(meshes), d_i, n, d_emit, 12345)
    torch.cuda.synchronize()
    got = d_emit.cpu().numpy()
    assert got[: n * 20].tobytes() == want.tobytes() and (got[n * 20:] == 0xEE).all()
