"""The C++ host-side mirror of the reference API (include/voidin.hpp) compiles against the C ABI
(CPU check) and, on a GPU box, drives the whole path bit-exact against the oracle."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "host_mirror_test")


def build():
    from oracle import ref
    ref.load()
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), SRC,
           "-L", os.path.join(ROOT, "voidin_amd", "csrc"), "-lvoidin_hip", "-L", os.path.join(ROOT, "oracle"), "-lvd_oracle",
           f"-Wl,-rpath,{os.path.join(ROOT, 'voidin_amd', 'csrc')}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", "-o", EXE]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)


def test_mirror_header_compiles_and_links():
    build()
    assert os.path.exists(EXE)


def _build_cpp(name):
    src = os.path.join(ROOT, "tests", "cpp", name + ".cpp")
    exe = os.path.join(ROOT, "tests", "cpp", name)
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), src,
               "-L", os.path.join(ROOT, "voidin_amd", "csrc"), "-lvoidin_hip", f"-Wl,-rpath,{os.path.join(ROOT, 'voidin_amd', 'csrc')}",
               "-o", exe]
        subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    return exe


def test_obj_reader_restates_tobj_gpu_load_options():
    """voidin::ObjModel::load (OBJ ingest of models/mod.rs:19-57) on a hand-written fixture; host code only."""
    src = os.path.join(ROOT, "tests", "cpp", "obj_reader_test.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "obj_reader_test")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), src,
           "-L", os.path.join(ROOT, "voidin_amd", "csrc"), "-lvoidin_hip", f"-Wl,-rpath,{os.path.join(ROOT, 'voidin_amd', 'csrc')}",
           "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "two_objects.obj")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "obj_reader_test OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_mirror_drives_the_path_bit_exact():
    if not os.path.exists(EXE):
        build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host_mirror_test OK" in out.stdout


def test_external_buffer_test_compiles():
    _build_cpp("external_buffer_test")


@pytest.mark.gpu
def test_external_buffer_import_round_trip():
    """an fd-exported allocation mapped through vd_import_external_buffer receives the command list."""
    exe = _build_cpp("external_buffer_test")
    out = subprocess.run(["timeout", "120", exe], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stdout + out.stderr
    if "SKIP" in out.stdout:
        pytest.skip(out.stdout.strip())
    assert "external_buffer_test OK" in out.stdout
