"""Oracle pinning for the cull/emit path: C oracle == numpy restatement == golden fixtures.
(The reference has no tests or vectors of its own — SURVEY.md §4 — so these freeze OUR
restatement of shaders/emit_draws.wgsl:13-64.)"""
import numpy as np
import pytest

from conftest import golden
from oracle import np_restate as npr
from voidin_amd import abi, synth

CASES = ["cull_model_wide.npz", "cull_model_small.npz", "cull_jitter_wide.npz", "cull_jitter_small.npz",
         "cull_model_scene.npz", "cull_model_scene_nave.npz", "cull_model_scene_x3.npz"]   # the last three: the reference's own demo scene (make_golden.model_scene_cases)


@pytest.mark.parametrize("name", CASES)
def test_c_oracle_matches_golden(oracle, name):
    g = golden(name)
    d = oracle.cull_emit(g["camera"], g["meshes"], g["instances"])
    assert d.tobytes() == g["draws"].tobytes()
    comp, cnt = oracle.compact(d)
    assert cnt == int(g["count"]) and comp[:cnt].tobytes() == g["compact"].tobytes()


@pytest.mark.parametrize("name", CASES)
def test_numpy_restatement_matches_golden(name):
    g = golden(name)
    d = npr.cull_emit(g["camera"][()], g["meshes"], g["instances"])
    assert d.tobytes() == g["draws"].tobytes()


def test_every_slot_written_and_fields(oracle):
    # emit_draws.wgsl:55-63: culled slots keep the mesh fields, base_instance = index
    g = golden("cull_model_small.npz")
    d, inst, m = g["draws"], g["instances"], g["meshes"]
    assert np.array_equal(d["base_instance"], np.arange(len(d)))
    assert np.array_equal(d["vertex_count"], m["index_count"][inst["mesh"]])
    assert np.array_equal(d["base_index"], m["base_index"][inst["mesh"]])
    assert set(np.unique(d["instance_count"])) <= {0, 1}
    assert 0 < d["instance_count"].sum() < len(d)


def test_multithreaded_oracle_equals_scalar(oracle):
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    inst = synth.instances(50_000, scale_range=(0.02, 0.5), extent=500.0)
    a = oracle.cull_emit(cam, meshes, inst, threads=1)
    b = oracle.cull_emit(cam, meshes, inst, threads=4)
    c = npr.cull_emit(cam, meshes, inst)
    assert a.tobytes() == b.tobytes() == c.tobytes()


def test_zfar_test_never_fires_with_infinite_far(oracle):
    # emit_draws.wgsl:28-30 with zfar = +inf (camera.rs:38) never culls
    cam = synth.camera_uniform()
    assert np.isinf(cam["zfar"]) and cam["znear"] == np.float32(0.001)
    cam2 = cam.copy()
    cam2["zfar"] = np.float32(50.0)
    meshes = synth.mesh_infos()
    inst = synth.instances(20_000, scale_range=(0.01, 0.2), extent=800.0)
    a = oracle.cull_emit(cam, meshes, inst)["instance_count"].sum()
    b = oracle.cull_emit(cam2, meshes, inst)["instance_count"].sum()
    assert b <= a


def test_compaction_definition(oracle):
    g = golden("cull_model_small.npz")
    comp, cnt = oracle.compact(g["draws"], pad_tail=True)
    keep = g["draws"]["instance_count"] == 1
    assert cnt == keep.sum()
    assert np.array_equal(comp["base_instance"][:cnt], np.nonzero(keep)[0])
    assert not comp.view(np.uint8).reshape(len(comp), 20)[cnt:].any()


def test_empty_and_degenerate_inputs(oracle):
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    assert len(oracle.cull_emit(cam, meshes, np.zeros(0, abi.INSTANCE))) == 0
    inst = synth.instances(8)
    inst["mesh"][3] = 1000  # out of range: clamped (reference: undefined)
    d = oracle.cull_emit(cam, meshes, inst)
    assert d["vertex_count"][3] == meshes["index_count"][-1]
    inst["transform"][5] = np.nan
    d = oracle.cull_emit(cam, meshes, inst)
    assert d["instance_count"][5] == 1  # every comparison with NaN is false => visible
