"""Parity of the HIP SAH BLAS builder against the CPU oracle, through the C ABI: node arrays
and the permuted index buffer are compared bit for bit."""
import numpy as np
import pytest

from conftest import fields_equal, golden
from voidin_amd import abi, synth
from voidin_amd.runtime import BvhBuilder, VoidinError

pytestmark = pytest.mark.gpu

BLAS = ["blas_plane.npz", "blas_sphere_1_1.npz", "blas_soup64.npz", "blas_knot_2k.npz", "blas_sphere_1_10.npz",
        "blas_plane_rot.npz", "blas_cube_obj.npz",   # reference-held inputs: mesh/mod.rs:269-272, assets/cube/cube.obj
        "blas_soup_nan.npz"]                         # NaN vertices build (f32::min/max ignore a NaN: blas.rs:190-198)


def diff_report(got, want):
    if len(got) != len(want):
        return f"node count {len(got)} != {len(want)}"
    for f in want.dtype.names:
        bad = np.nonzero((got[f] != want[f]).reshape(len(want), -1).any(axis=1))[0]
        if len(bad):
            return f"field {f}: {len(bad)} nodes differ, first {bad[:5]}: got {got[f][bad[0]]} want {want[f][bad[0]]}"
    return "equal"


@pytest.mark.parametrize("name", BLAS)
def test_golden_fixtures(ctx, name):
    g = golden(name)
    nodes, idx = ctx.bvh_build(g["vertices"], g["indices"])
    assert fields_equal(nodes, g["nodes"]), diff_report(nodes, g["nodes"])
    assert np.array_equal(idx, g["indices_out"])


@pytest.mark.parametrize("n_tri", [1, 3, 4, 5, 17, 64, 65, 300, 511, 512, 513, 700, 1024, 1025, 2048, 2049, 3000])
def test_soup_sizes_vs_oracle(ctx, oracle, n_tri):
    # sizes around the wave (64), the LDS-subtree limit (512) and the phase-A item size (1024)
    v, i = synth.triangle_soup(n_tri, seed=synth.SEED_BASE + 30 + n_tri)
    want_nodes, want_idx = oracle.bvh_build(v, i)
    nodes, idx = ctx.bvh_build(v, i)
    assert fields_equal(nodes, want_nodes), diff_report(nodes, want_nodes)
    assert np.array_equal(idx, want_idx)


def test_wide_payload_path_vs_oracle(ctx, oracle, ctx_options):
    """Meshes above 2^25 triangles move an 8-byte {pos0, 21 predicate bits} payload through the rounds of phase A instead of
    the 4-byte {pos0, 7 bits of the current axis} one; VD_OPT_BLAS_WIDE_PAYLOAD forces that path at testable sizes."""
    ctx_options("blas.wide_payload", 1)
    for v, i in (synth.knot_mesh(256, 64), synth.triangle_soup(3000, seed=5), synth.triangle_soup(9000, seed=6)):
        want_nodes, want_idx = oracle.bvh_build(v, i)
        nodes, idx = ctx.bvh_build(v, i)
        assert fields_equal(nodes, want_nodes), diff_report(nodes, want_nodes)
        assert np.array_equal(idx, want_idx)


@pytest.mark.parametrize("shape", [(128, 32), (512, 64), (1024, 256)])
def test_knot_meshes_vs_oracle(ctx, oracle, shape):
    # up to 524k triangles: several phase-A levels over many segments
    v, i = synth.knot_mesh(*shape)
    want_nodes, want_idx = oracle.bvh_build(v, i)
    nodes, idx = ctx.bvh_build(v, i)
    assert fields_equal(nodes, want_nodes), diff_report(nodes, want_nodes)
    assert np.array_equal(idx, want_idx)


@pytest.mark.parametrize("n_tri,ratio", [(60, 3.0), (300, 1.25), (512, 1.14), (1500, 1.045), (5000, 1.012)])   # x stays below 1e30
def test_chain_like_trees_vs_oracle(ctx, oracle, n_tri, ratio):
    """Triangles at geometrically growing distances: every split peels a few prims off the far end, so the tree is one
    long chain - the worst case for the builder's work lists (nodes of one size class nest instead of sitting side by
    side) and for the depth of everything that walks the tree."""
    x = (ratio ** np.arange(n_tri, dtype=np.float64)).astype(np.float32)
    v = np.zeros((3 * n_tri, 3), dtype=np.float32)
    v[0::3, 0] = x; v[1::3, 0] = x * np.float32(1.01); v[2::3, 0] = x
    v[1::3, 1] = 0.5; v[2::3, 2] = 0.5
    i = np.arange(3 * n_tri, dtype=np.uint32)
    want_nodes, want_idx = oracle.bvh_build(v, i)
    depth = np.zeros(len(want_nodes), dtype=np.int64)
    for k in range(len(want_nodes)):                           # children come after their parent in pre-order
        if want_nodes["count"][k] == 0 and k != 1:
            l = int(want_nodes["left_first"][k]); depth[l] = depth[l + 1] = depth[k] + 1
    assert depth.max() > 25                                    # really a chain, not a bushy tree
    nodes, idx = ctx.bvh_build(v, i)
    assert fields_equal(nodes, want_nodes), diff_report(nodes, want_nodes)
    assert np.array_equal(idx, want_idx)


@pytest.mark.parametrize("n_tri", [40, 700, 5000])
def test_extreme_vertex_values_agree_with_oracle(ctx, oracle, n_tri):
    """A NaN vertex does NOT stop the reference: Rust's f32::min/max (glam scalar Vec3::min/max, blas.rs:190-198) ignore
    a NaN operand, the NaN centroid fails every `<` (blas.rs:173) and goes right - the tree builds, and must be the
    oracle's tree, bit for bit, with no NaN in any box.  An infinite vertex makes every candidate cost of its node
    inf or NaN (`cost < optimal_cost` never holds, blas.rs:156): the reference crashes there (blas.rs:137-140,114-116),
    the oracle and the library both answer VD_ERR_DEGENERATE.  A coordinate beyond the 1e30 bound seeds
    (blas.rs:185-186) and heavily duplicated vertices still build - the same tree."""
    v, i = synth.triangle_soup(n_tri, seed=77)
    nan1 = v.copy(); nan1[5, 1] = np.nan
    nan3 = v.copy(); nan3[5, 1] = np.nan; nan3[30:33, 2] = np.nan; nan3[61] = np.nan      # one coordinate, a whole triangle's z, a whole vertex
    for vv in (nan1, nan3):
        want_nodes, want_idx = oracle.bvh_build(vv, i)
        nodes, idx = ctx.bvh_build(vv, i)
        assert fields_equal(nodes, want_nodes), diff_report(nodes, want_nodes)
        assert np.array_equal(idx, want_idx)
        assert not np.isnan(nodes["min"]).any() and not np.isnan(nodes["max"]).any()
    for val in (np.inf, -np.inf):
        v2 = v.copy(); v2[5, 1] = val
        with pytest.raises(oracle.OracleError):
            oracle.bvh_build(v2, i)
        with pytest.raises(VoidinError) as e:
            ctx.bvh_build(v2, i)
        assert e.value.code == abi.VD_ERR_DEGENERATE
    big = v.copy(); big[5, 1] = 1e31
    dup = v.copy(); dup[::7] = dup[0]
    for vv in (big, dup):
        want_nodes, want_idx = oracle.bvh_build(vv, i)
        nodes, idx = ctx.bvh_build(vv, i)
        assert fields_equal(nodes, want_nodes), diff_report(nodes, want_nodes)
        assert np.array_equal(idx, want_idx)


def test_generated_nan_centroids_agree_with_oracle(ctx, oracle):
    """NaNs the arithmetic GENERATES differ between the checker and the device (inf - inf is 0xFFC00000 on x86 and
    0x7FC00000 on gfx950): vertices at +inf and -inf in one triangle make its centroid inf + -inf = NaN on both sides,
    with different sign bits; a centroid sum that overflows (3e38 + 3e38) makes an inf centroid and an inf split plane.
    Nothing may depend on a NaN's sign: it drops out of `cb`, the predicate is false either way.  In the BLAS such
    triangles always come with a box whose area overflows, so every candidate cost of their node is inf or NaN and the
    reference crashes there (blas.rs:137-140) - both sides must say VD_ERR_DEGENERATE, neither may hang or differ.  (A
    generated NaN that BUILDS exists on the TLAS side: tests/golden/tlas_nan_60.npz, inf - inf in transform_point3.)"""
    v, i = synth.triangle_soup(300, seed=78)
    g = v.copy(); g[3, 0] = np.inf; g[4, 0] = -np.inf
    h = v.copy(); h[9:12, 2] = 3e38
    k = v.copy(); k[3, 0] = np.inf; k[4, 0] = -np.inf; k[100, 1] = np.nan          # a given NaN beside the generated one
    for vv in (g, h, k):
        with pytest.raises(oracle.OracleError) as eo:
            oracle.bvh_build(vv, i)
        assert eo.value.code == abi.VD_ERR_DEGENERATE
        with pytest.raises(VoidinError) as e:
            ctx.bvh_build(vv, i)
        assert e.value.code == abi.VD_ERR_DEGENERATE


def test_signalling_nan_vertices_agree_with_oracle(ctx, oracle):
    """A SIGNALLING NaN (0x7fa00000) is a NaN to `a != a` on the CPU and must be one to the device's v_min_f32 / v_max_f32
    too: in IEEE mode the bare instruction quiets and RETURNS it (a NaN box; found by this test at vertex 55), so the
    precompute pass quiets vertex coordinates as it loads them: ignored by every box, false in every `<`, both sign bits."""
    v, i = synth.triangle_soup(900, seed=79)
    v = v.copy()
    bits = v.view(np.uint32)
    bits[7, 0] = 0x7FA00000; bits[55, 2] = 0xFFA00001; bits[300, 1] = 0x7F800001; bits[601, 2] = 0x7FA00001; bits[1202, 0] = 0xFF800001
    assert np.isnan(v).sum() == 5
    want_nodes, want_idx = oracle.bvh_build(v, i)
    nodes, idx = ctx.bvh_build(v, i)
    assert fields_equal(nodes, want_nodes), diff_report(nodes, want_nodes)
    assert np.array_equal(idx, want_idx)
    assert not np.isnan(nodes["min"]).any() and not np.isnan(nodes["max"]).any()
    meshes = synth.mesh_infos()
    inst = synth.instances(500, seed=synth.SEED_BASE + 61, extent=80.0)
    t = inst["transform"].view(np.uint32)
    t[17, 12] = 0x7FA00000; t[200, 0] = 0xFFA00000; t[333, 9] = 0x7F800001
    got, want = ctx.tlas_build(inst, meshes), oracle.tlas_build(inst, meshes)
    assert fields_equal(got, want)
    assert ctx.tlas_refit(inst, meshes, got).tobytes() == oracle.tlas_refit(inst, meshes, want).tobytes() == want.tobytes()


def test_builder_api_permutes_callers_indices(ctx):
    # BvhBuilder::new(&[Vec3], &mut [UVec3]).build() -> Bvh{nodes}; caller's slice permuted (blas.rs:95-100)
    g = golden("blas_soup64.npz")
    idx = g["indices"].copy()
    bvh = BvhBuilder(ctx, g["vertices"], idx).set_bin_number(32).build()   # num_bins is ignored (blas.rs:136)
    assert fields_equal(bvh.nodes, g["nodes"]) and np.array_equal(idx, g["indices_out"])


def test_degenerate_and_bad_input(ctx):
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    idx = np.tile(np.array([0, 1, 2], np.uint32), 5)
    with pytest.raises(VoidinError) as e:
        ctx.bvh_build(v, idx)
    assert e.value.code == abi.VD_ERR_DEGENERATE
    with pytest.raises(VoidinError) as e:
        ctx.bvh_build(v, np.array([0, 1, 7], np.uint32))
    assert e.value.code == abi.VD_ERR_INVALID_ARG
    # big degenerate: 2000 identical triangles -> phase A rejects every candidate
    with pytest.raises(VoidinError) as e:
        ctx.bvh_build(v, np.tile(np.array([0, 1, 2], np.uint32), 2000))
    assert e.value.code == abi.VD_ERR_DEGENERATE


def test_device_entry_point_and_determinism(ctx, oracle):
    import torch
    v, i = synth.knot_mesh(256, 64)
    n_tri = len(i) // 3
    want_nodes, want_idx = oracle.bvh_build(v, i)
    outs = []
    for _ in range(2):
        d_v = ctx.upload(v)
        d_i = ctx.upload(i.copy())
        d_n = ctx.empty(2 * n_tri * 32)
        n_nodes = ctx.bvh_build_dev(d_v, len(v), d_i, n_tri, d_n, 2 * n_tri)
        torch.cuda.synchronize()
        outs.append((d_n.cpu().numpy()[: n_nodes * 32].view(abi.BVH_NODE), d_i.cpu().numpy().view(np.uint32)[: 3 * n_tri]))
    assert fields_equal(outs[0][0], want_nodes) and np.array_equal(outs[0][1], want_idx)
    assert outs[0][0].tobytes() == outs[1][0].tobytes() and np.array_equal(outs[0][1], outs[1][1])


def test_denormal_vertex_coordinates_survive_the_quieting_pass(ctx, oracle):
    """blas_precompute_kernel passes every vertex coordinate through a canonicalising v_max x, x (signalling NaNs); that
    instruction must leave f32 denormals alone (the kernels run with denormals preserved, like the x86 oracle): a mesh
    squeezed into the denormal range around zero, and one with a few denormal coordinates among ordinary ones, build the
    oracle's tree bit for bit."""
    v, i = synth.triangle_soup(700, seed=81)
    tiny = (v * np.float32(1e-39)).astype(np.float32)                      # every coordinate denormal (|x| < 1.2e-38)
    assert (np.abs(tiny[tiny != 0]) < np.float32(1.1754944e-38)).all() and (tiny != 0).sum() > 1000
    mixed = v.copy(); mixed[::13] = tiny[::13]; mixed[5, 1] = np.float32(-1e-45)
    for vv in (tiny, mixed):
        try:
            want_nodes, want_idx = oracle.bvh_build(vv, i)
        except oracle.OracleError as e:                                     # all-denormal extents: areas underflow to 0 -> no split passes
            assert e.code == abi.VD_ERR_DEGENERATE
            with pytest.raises(VoidinError) as g:
                ctx.bvh_build(vv, i)
            assert g.value.code == abi.VD_ERR_DEGENERATE
            continue
        nodes, idx = ctx.bvh_build(vv, i)
        assert fields_equal(nodes, want_nodes), diff_report(nodes, want_nodes)
        assert np.array_equal(idx, want_idx)
