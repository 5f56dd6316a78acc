"""The HiZ occlusion extension.  voidin has no occlusion culling, so there is no reference to be
equal to: the CPU twin (oracle/vd_oracle_occlusion.c) is checked against brute-force numpy for what the extension
promises - the pyramid is a min pyramid, and nothing that can be seen is ever culled - and the HIP kernels are
checked bit for bit against the twin."""
import math

import numpy as np
import pytest

from conftest import golden
from voidin_amd import abi, synth


def _cloud(n, seed, cam_z=50.0):
    """Instances in front of a camera at (0, 0, cam_z) looking down -z: translate + non-uniform scale + rotation."""
    meshes = synth.mesh_infos()
    inst = synth.instances(n, seed=seed, extent=300.0, centre=(0.0, 0.0, cam_z - 200.0), scale_range=(0.25, 4.0))
    return meshes, inst


def _all_bits(n):
    m = np.zeros((n + 63) // 64, dtype=np.uint64)
    m[:] = np.uint64(0xFFFFFFFFFFFFFFFF)
    if n % 64:
        m[-1] = np.uint64((1 << (n % 64)) - 1)
    return m


def _bits(mask, n):
    return np.unpackbits(mask.view(np.uint8), bitorder="little")[:n].astype(bool)


def _random_depth(w, h, seed):
    """Blocky depth: big rectangles of near occluders over a far background, some exactly 0 (cleared)."""
    u = synth.uniform01(seed, 0, 64 * 5).reshape(64, 5).astype(np.float64)
    d = np.zeros((h, w), dtype=np.float32)
    for x, y, sx, sy, z in u:
        x0, y0 = int(x * w), int(y * h)
        d[y0: y0 + 1 + int(sy * h / 3), x0: x0 + 1 + int(sx * w / 3)] = np.float32(0.001 / (20.0 + 300.0 * z))
    return d


@pytest.mark.parametrize("w,h", [(1, 1), (2, 1), (5, 3), (64, 64), (100, 37), (640, 360)])
def test_pyramid_is_the_block_minimum(oracle, w, h):
    L = oracle.hiz_layout(w, h)
    assert (L.level_width[0], L.level_height[0]) == (w, h)
    assert (L.level_width[L.n_levels - 1], L.level_height[L.n_levels - 1]) == (1, 1)
    depth = synth.uniform01(synth.SEED_BASE + 40, 0, w * h).reshape(h, w).astype(np.float32)
    pyr = oracle.hiz_build(depth)
    assert len(pyr) == L.total_texels == sum(L.level_width[k] * L.level_height[k] for k in range(L.n_levels))
    for k in range(L.n_levels):
        lw, lh = L.level_width[k], L.level_height[k]
        assert (lw, lh) == (((w - 1) >> k) + 1, ((h - 1) >> k) + 1)
        lvl = pyr[L.level_offset[k]: L.level_offset[k] + lw * lh].reshape(lh, lw)
        for y in range(0, lh, max(1, lh // 7)):
            for x in range(0, lw, max(1, lw // 7)):
                assert lvl[y, x] == depth[y << k: (y + 1) << k, x << k: (x + 1) << k].min()
    with pytest.raises(oracle.OracleError):
        oracle.hiz_layout(0, 4)
    with pytest.raises(oracle.OracleError):
        oracle.hiz_layout(65537, 1)


def test_far_depth_culls_nothing_near_depth_culls_everything_in_front(oracle):
    cam = synth.camera_uniform(eye=(0, 0, 50), pitch_deg=0)
    meshes, inst = _cloud(3000, synth.SEED_BASE + 41)
    w, h = 320, 256
    full = _all_bits(len(inst))
    far = oracle.hiz_build(np.zeros((h, w), np.float32))
    assert np.array_equal(oracle.occlusion_mask(cam, meshes, inst, far, w, h, full), full)
    near = oracle.hiz_build(np.ones((h, w), np.float32))
    out = _bits(oracle.occlusion_mask(cam, meshes, inst, near, w, h, full), len(inst))
    assert 0 < out.sum() < len(inst) // 2          # what survives touches the near plane or is off screen
    # bits that are clear on input stay clear
    half = full.copy()
    half[::2] = 0
    got = oracle.occlusion_mask(cam, meshes, inst, far, w, h, half)
    assert np.array_equal(got, half)
    # a projection that is not the reference's kind is refused
    bad = cam.copy()
    bad["projection"][11] = 1.0
    with pytest.raises(oracle.OracleError):
        oracle.occlusion_mask(bad, meshes, inst, far, w, h, full)


def test_twin_matches_golden(oracle):
    """The C twin against the fixture written from the independent numpy statement (make_golden.py: occlusion_case)."""
    g = golden("occlusion_1500.npz")
    h, w = g["depth"].shape
    pyr = oracle.hiz_build(g["depth"])
    assert pyr.tobytes() == g["pyramid"].tobytes()
    got = oracle.occlusion_mask(g["camera"], g["meshes"], g["instances"], pyr, w, h, g["mask_in"])
    assert np.array_equal(got, g["mask_out"])


@pytest.mark.gpu
def test_gpu_matches_golden(ctx):
    import torch
    g = golden("occlusion_1500.npz")
    h, w = g["depth"].shape
    n = len(g["instances"])
    d_pyr = torch.zeros(len(g["pyramid"]), dtype=torch.float32, device="cuda")
    ctx.hiz_build_dev(ctx.upload(g["depth"]), w, h, d_pyr)
    assert d_pyr.cpu().numpy().tobytes() == g["pyramid"].tobytes()
    d_in = ctx.upload(g["mask_in"].view(np.int64))
    d_out = torch.zeros_like(d_in)
    ctx.occlusion_mask_dev(g["camera"], ctx.upload(g["meshes"]), len(g["meshes"]), ctx.upload(g["instances"]), n, d_pyr, w, h, d_in, d_out)
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), g["mask_out"])


def _sphere_rect_f64(cam, mesh, T, w, h):
    """Independent float64 statement of the footprint: the angular interval of the bounding sphere per axis
    (atan/asin, not the tangent-slope formula of the kernels).  Returns (x0, x1, y0, y1, nearest depth) in pixels
    without any margin, or None when the sphere is not wholly beyond the near plane."""
    V = cam["view"].reshape(4, 4).astype(np.float64).T
    P = cam["projection"].reshape(4, 4).astype(np.float64).T
    M = T.reshape(4, 4).astype(np.float64).T
    c0 = (mesh["max"].astype(np.float64) + mesh["min"].astype(np.float64)) / 2
    c = (V @ M @ np.array([*c0, 1.0]))[:3]
    ms = max(np.linalg.norm(M[:3, 0]), np.linalg.norm(M[:3, 1]), np.linalg.norm(M[:3, 2]))
    r = np.linalg.norm(mesh["max"].astype(np.float64) - mesh["min"].astype(np.float64)) / 2 * ms
    d = -c[2]
    if d - r <= float(cam["znear"]):
        return None
    out = []
    for a, p, off, n in ((c[0], P[0, 0], P[0, 2], w), (c[1], P[1, 1], P[1, 2], h)):
        theta, alpha = math.atan2(a, d), math.asin(min(1.0, r / math.hypot(a, d)))
        if theta + alpha >= math.pi / 2 or theta - alpha <= -math.pi / 2:
            return None
        out.append((p * math.tan(theta - alpha) - off, p * math.tan(theta + alpha) - off, n))
    (nx0, nx1, _), (ny0, ny1, _) = out
    return ((nx0 * 0.5 + 0.5) * w, (nx1 * 0.5 + 0.5) * w, (0.5 - ny1 * 0.5) * h, (0.5 - ny0 * 0.5) * h, float(cam["znear"]) / (d - r))


def test_culled_instances_are_hidden_at_full_resolution(oracle):
    """Conservative: whenever the twin culls an instance, every depth texel its bounding sphere can touch is nearer
    than the sphere's nearest point (checked on level 0 with an independent float64 footprint)."""
    cam = synth.camera_uniform(eye=(0, 0, 50), pitch_deg=0, jitter=(0.003, -0.002))
    meshes, inst = _cloud(6000, synth.SEED_BASE + 42)
    w, h = 200, 150
    depth = _random_depth(w, h, synth.SEED_BASE + 43)
    pyr = oracle.hiz_build(depth)
    full = _all_bits(len(inst))
    keep = _bits(oracle.occlusion_mask(cam, meshes, inst, pyr, w, h, full), len(inst))
    n_culled = 0
    for i in np.flatnonzero(~keep):
        rect = _sphere_rect_f64(cam, meshes[min(int(inst["mesh"][i]), len(meshes) - 1)], inst["transform"][i], w, h)
        assert rect is not None
        x0, x1, y0, y1, ds = rect
        xs = slice(max(0, int(math.floor(x0))), min(w, int(math.floor(x1)) + 1))
        ys = slice(max(0, int(math.floor(y0))), min(h, int(math.floor(y1)) + 1))
        assert xs.start < xs.stop and ys.start < ys.stop
        assert depth[ys, xs].min() > ds * (1 - 1e-5), i
        n_culled += 1
    assert n_culled > 200 and keep.sum() > 200
    # and it does cull what is plainly hidden: a sphere wholly inside one near rectangle, far behind it
    V = cam["view"].reshape(4, 4).astype(np.float64).T
    hidden = 0
    for i in np.flatnonzero(keep)[:2000]:
        rect = _sphere_rect_f64(cam, meshes[min(int(inst["mesh"][i]), len(meshes) - 1)], inst["transform"][i], w, h)
        if rect is None:
            continue
        x0, x1, y0, y1, ds = rect
        if x0 < 2 or y0 < 2 or x1 > w - 2 or y1 > h - 2:
            continue
        # widen to the power-of-two block the test may look at
        span = max(x1 - x0, y1 - y0) + 2
        k = max(0, math.ceil(math.log2(span))) + 1
        bx0, by0 = (int(x0 - 1) >> k) << k, (int(y0 - 1) >> k) << k
        blk = depth[by0: by0 + (2 << k), bx0: bx0 + (2 << k)]
        if blk.size and blk.min() > ds * (1 + 1e-5):
            hidden += 1
    assert hidden == 0      # everything that hidden would have been culled


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1, 1), (5, 3), (64, 64), (100, 37), (1920, 1080), (2048, 2048), (4097, 3)])
def test_gpu_pyramid_bit_exact(ctx, oracle, w, h):
    import torch
    depth = synth.uniform01(synth.SEED_BASE + 44, 0, w * h).reshape(h, w).astype(np.float32)
    want = oracle.hiz_build(depth)
    L = ctx.hiz_layout(w, h)
    Lr = oracle.hiz_layout(w, h)
    assert bytes(L) == bytes(Lr)
    d_pyr = torch.full((L.total_texels,), -1.0, dtype=torch.float32, device="cuda")
    ctx.hiz_build_dev(ctx.upload(depth), w, h, d_pyr)
    assert d_pyr.cpu().numpy().tobytes() == want.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("n,w,h", [(1, 64, 64), (63, 5, 3), (5000, 200, 150), (200_000, 1920, 1080)])
def test_gpu_occlusion_mask_bit_exact(ctx, oracle, n, w, h):
    import torch
    cam = synth.camera_uniform(eye=(0, 0, 50), pitch_deg=0, jitter=(0.003, -0.002))
    meshes, inst = _cloud(n, synth.SEED_BASE + 45)
    depth = _random_depth(w, h, synth.SEED_BASE + 46)
    pyr = oracle.hiz_build(depth)
    d_m, d_i = ctx.upload(meshes), ctx.upload(inst)
    d_pyr = torch.zeros(len(pyr), dtype=torch.float32, device="cuda")
    ctx.hiz_build_dev(ctx.upload(depth), w, h, d_pyr)
    # frustum mask from the path itself, then the occlusion refinement
    d_mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    ctx.cull_mask_dev(cam, d_m, len(meshes), d_i, n, d_mask)
    frustum = d_mask.cpu().numpy().view(np.uint64)
    want = oracle.occlusion_mask(cam, meshes, inst, pyr, w, h, frustum)
    d_out = torch.full_like(d_mask, -1)
    ctx.occlusion_mask_dev(cam, d_m, len(meshes), d_i, n, d_pyr, w, h, d_mask, d_out)
    got = d_out.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)
    assert np.array_equal(got & frustum, got)
    if n >= 5000:
        assert 0 < _bits(got, n).sum() < _bits(frustum, n).sum()
    ctx.occlusion_mask_dev(cam, d_m, len(meshes), d_i, n, d_pyr, w, h, d_mask, d_mask)          # in place
    assert np.array_equal(d_mask.cpu().numpy().view(np.uint64), want)
    # the refined mask feeds the ordinary expansion: an ordered draw list of the unoccluded instances
    ids = ctx.upload(np.minimum(inst["mesh"], len(meshes) - 1).astype(np.uint8))
    d_draws, d_cnt = ctx.empty(n * 20), torch.zeros(4, dtype=torch.int32, device="cuda")
    ctx.expand_mask_dev(d_mask, n, n, ids, d_m, len(meshes), d_draws, d_cnt)
    k = int(d_cnt[0])
    assert k == _bits(want, n).sum()
    draws = d_draws.cpu().numpy()[: k * 20].view(abi.DRAW)
    assert np.array_equal(draws["base_instance"], np.flatnonzero(_bits(want, n)))


@pytest.mark.gpu
def test_gpu_occlusion_with_poisoned_and_near_plane_instances(ctx, oracle):
    """NaN / inf transforms, zero scale, instances around and behind the camera: every comparison with NaN fails and
    leaves the instance visible; the kernel and the twin agree bit for bit."""
    import torch
    cam = synth.camera_uniform(eye=(0, 0, 50), pitch_deg=0)
    meshes, inst = _cloud(4000, synth.SEED_BASE + 48)
    inst["transform"][10, 12] = np.nan
    inst["transform"][11, 0] = np.inf
    inst["transform"][12, :12] = 0.0                       # zero scale: radius 0
    inst["transform"][13, 14] = 50.0                       # at the eye
    inst["transform"][14, 14] = 500.0                      # far behind the camera
    inst["transform"][15:400, 14] = 50.0 - synth.uniform01(synth.SEED_BASE + 49, 0, 385).astype(np.float32) * 3.0   # straddling the near plane
    w, h = 160, 120
    depth = np.full((h, w), np.float32(0.001 / 5.0), dtype=np.float32)     # a wall 5 units away: hides whatever is wholly behind it
    pyr = oracle.hiz_build(depth)
    full = _all_bits(len(inst))
    want = oracle.occlusion_mask(cam, meshes, inst, pyr, w, h, full)
    d_pyr = torch.zeros(len(pyr), dtype=torch.float32, device="cuda")
    ctx.hiz_build_dev(ctx.upload(depth), w, h, d_pyr)
    d_in = ctx.upload(full.view(np.int64))
    d_out = torch.zeros_like(d_in)
    ctx.occlusion_mask_dev(cam, ctx.upload(meshes), len(meshes), ctx.upload(inst), len(inst), d_pyr, w, h, d_in, d_out)
    got = d_out.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)
    keep = _bits(got, len(inst))
    assert keep[10] and keep[11] and keep[13] and keep[14]         # NaN, inf, at / behind the eye: never culled
    assert 0 < keep.sum() < len(inst)


@pytest.mark.gpu
def test_gpu_occlusion_bad_arguments(ctx):
    import torch
    cam = synth.camera_uniform()
    z = torch.zeros(64, dtype=torch.int64, device="cuda")
    lib, h = ctx.lib, ctx.h
    c = np.ascontiguousarray(cam).reshape(1)
    assert lib.vd_hiz_build_dev(h, None, 4, 4, z.data_ptr()) == abi.VD_ERR_INVALID_ARG
    assert lib.vd_hiz_build_dev(h, z.data_ptr(), 0, 4, z.data_ptr()) == abi.VD_ERR_INVALID_ARG
    assert lib.vd_occlusion_mask_dev(h, c.ctypes.data, z.data_ptr(), 1, z.data_ptr(), 0, z.data_ptr(), 4, 4, None, None) == abi.VD_OK
    assert lib.vd_occlusion_mask_dev(h, c.ctypes.data, z.data_ptr(), 1, z.data_ptr(), 8, z.data_ptr(), 4, 4, None, None) == abi.VD_ERR_INVALID_ARG
    bad = c.copy()
    bad["projection"][0][15] = 1.0
    assert lib.vd_occlusion_mask_dev(h, bad.ctypes.data, z.data_ptr(), 1, z.data_ptr(), 8, z.data_ptr(), 4, 4, z.data_ptr(), z.data_ptr()) == abi.VD_ERR_INVALID_ARG
    assert b"perspective" in lib.vd_last_error(h)


@pytest.mark.gpu
def test_gpu_two_pass_scheme_never_hides_a_visible_instance(ctx):
    """End to end with the path's own ray tracer as the rasteriser: depth = nearest hit per pixel of the occluders
    drawn in pass one; pass two tests everything else against its pyramid.  Every instance that owns a pixel of the
    ray-traced image of the WHOLE scene must survive; instances behind the big occluder must not."""
    import torch
    cam = synth.camera_uniform(eye=(0, 0, 50), pitch_deg=0, aspect=1.0)
    sv, si = synth.uv_sphere(1.0, 24)
    nodes, idx = ctx.bvh_build(sv, si)
    mn, mx = synth.mesh_bounds(sv)
    infos = np.zeros(1, dtype=abi.MESH_INFO)
    infos["min"], infos["max"], infos["index_count"] = mn, mx, len(idx)
    n_small = 1500
    u = synth.uniform01(synth.SEED_BASE + 47, 0, 4 * n_small).reshape(n_small, 4).astype(np.float64)
    mats = [np.diag([30.0, 30.0, 30.0, 1.0])]
    mats[0][:3, 3] = (0, 0, -20)
    for x, y, z, s in u:
        M = np.diag([1 + 3 * s, 1 + 3 * s, 1 + 3 * s, 1.0])
        M[:3, 3] = ((x - 0.5) * 300, (y - 0.5) * 300, -60 - 240 * z)
        mats.append(M)
    inst = np.stack([synth.instance_from_matrix(M.T.reshape(16), 0) for M in mats])
    n = len(inst)
    tl = ctx.tlas_build(inst, infos)
    # pixel-centre rays
    W = H = 256
    V = cam["view"].reshape(4, 4).astype(np.float64).T
    C2W = cam["clip_to_world"].reshape(4, 4).astype(np.float64).T
    py, px = np.mgrid[0:H, 0:W]
    nx, ny = (px.reshape(-1) + 0.5) / W * 2 - 1, 1 - (py.reshape(-1) + 0.5) / H * 2
    p = (C2W @ np.stack([nx, ny, np.ones_like(nx), np.ones_like(nx)])).T
    p = p[:, :3] / p[:, 3:4]
    eye = cam["view_position"][:3].astype(np.float64)
    dirs = p - eye
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    rays = np.zeros(W * H, dtype=abi.RAY)
    rays["eye"], rays["dir"] = eye, dirs

    def render(which):
        sub = inst[which]
        hits = ctx.trace((ctx.tlas_build(sub, infos), sub, infos, nodes, sv, idx), rays)
        P = rays["eye"].astype(np.float64) + hits["dist"].astype(np.float64)[:, None] * rays["dir"].astype(np.float64)
        zv = (np.c_[P, np.ones(len(P))] @ V.T)[:, 2]
        depth = np.where(hits["hit"] == 1, float(cam["znear"]) / np.maximum(-zv, 1e-9), 0.0).astype(np.float32)
        return hits, depth.reshape(H, W)

    hits_all, _ = render(np.arange(n))
    seen = np.unique(hits_all["instance"][hits_all["hit"] == 1])
    # pass one drew the big occluder only
    _, depth = render(np.array([0]))
    L = ctx.hiz_layout(W, H)
    d_pyr = torch.zeros(L.total_texels, dtype=torch.float32, device="cuda")
    ctx.hiz_build_dev(ctx.upload(depth), W, H, d_pyr)
    d_m, d_i = ctx.upload(infos), ctx.upload(inst)
    d_mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    ctx.cull_mask_dev(cam, d_m, 1, d_i, n, d_mask)
    frustum = _bits(d_mask.cpu().numpy().view(np.uint64), n)
    ctx.occlusion_mask_dev(cam, d_m, 1, d_i, n, d_pyr, W, H, d_mask, d_mask)
    keep = _bits(d_mask.cpu().numpy().view(np.uint64), n)
    assert frustum[seen].all() and keep[seen].all()                      # nothing visible was lost
    culled = frustum & ~keep
    assert culled.sum() > 100                                            # and the occluder hides a good part
    assert not np.isin(np.flatnonzero(culled), seen).any()
    del tl
