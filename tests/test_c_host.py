"""The drop-in boundary from a plain C11 host compiled by gcc (`extern "C"`, plain pointers and sizes - what a cgo /
Rust FFI / C caller binds): tests/c/host_c_test.c drives cull + emit, the compaction in both forms, a BLAS build and a TLAS build
through include/voidin_abi.h and compares with the oracle's bytes, which this test writes next to the scene."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from voidin_amd import abi, synth

SRC = os.path.join(ROOT, "tests", "c", "host_c_test.c")
EXE = os.path.join(ROOT, "tests", "c", "host_c_test")
CSRC = os.path.join(ROOT, "voidin_amd", "csrc")


def _build():
    newest = max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(ROOT, "include", "voidin_abi.h")))
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < newest:
        subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), SRC,
                        "-L", CSRC, "-lvoidin_hip", f"-Wl,-rpath,{CSRC}", "-o", EXE], check=True, capture_output=True, timeout=300)
    return EXE


def test_c_host_compiles_with_gcc_as_c11():
    _build()


@pytest.mark.gpu
def test_c_host_drives_the_path_bit_exact(oracle, tmp_path):
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    inst = synth.instances(3000, seed=synth.SEED_BASE + 90, scale_range=(0.02, 0.6), extent=600.0)
    draws = oracle.cull_emit(cam, meshes, inst)
    compact, count = oracle.compact(draws)
    v, i = synth.knot_mesh(96, 24)
    v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 3)
    i = np.ascontiguousarray(i, dtype=np.uint32).reshape(-1)
    nodes, idx_out = oracle.bvh_build(v, i)
    tlas = oracle.tlas_build(inst, meshes)
    p = tmp_path / "scene.bin"
    with open(p, "wb") as f:
        f.write(struct.pack("<6I", len(meshes), len(inst), len(v), len(i) // 3, len(nodes), count))
        for a in (np.ascontiguousarray(cam, dtype=abi.CAMERA).reshape(1), meshes, inst, draws, compact[:count], v, i, nodes, idx_out, tlas):
            f.write(np.ascontiguousarray(a).tobytes())
    out = subprocess.run(["timeout", "300", _build(), str(p)], capture_output=True, text=True, timeout=400)
    assert out.returncode == 0 and "host_c_test OK" in out.stdout, out.stdout + out.stderr
