"""The C-ABI library loads and exports every symbol include/voidin_abi.h declares
(no compute calls: this runs without a GPU)."""
import ctypes as C
import os
import re

import numpy as np

from conftest import ROOT
from voidin_amd import abi


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "voidin_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vd_[a-z_0-9]+)\s*\(", src)))


def test_header_and_bindings_agree():
    assert declared_symbols() == sorted(abi.PROTOTYPES)


def test_integration_md_binds_every_export():
    """INTEGRATION.md section 2 (the Rust `extern "C"` block a maintainer adds) binds every function the header declares."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index("<!-- ffi:begin -->"): doc.index("<!-- ffi:end -->")]
    bound = sorted(set(re.findall(r"pub fn (vd_[a-z_0-9]+)\(", block)))
    assert bound == declared_symbols()


def test_library_exports_every_declared_symbol():
    lib = abi.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.vd_version()


def test_struct_sizes_match_the_wire_contract():
    # SURVEY.md §8a D1-D6
    assert abi.INSTANCE.itemsize == 144 and abi.INSTANCE.fields["mesh"][1] == 128
    assert abi.MESH_INFO.itemsize == 48 and abi.MESH_INFO.fields["max"][1] == 16
    assert abi.MESH_INFO.fields["vertex_offset"][1] == 32 and abi.MESH_INFO.fields["bvh_index"][1] == 36
    assert abi.DRAW.itemsize == 20 and abi.DRAW.fields["base_instance"][1] == 16
    assert abi.CAMERA.itemsize == 320 and abi.CAMERA.fields["view"][1] == 80
    assert abi.CAMERA.fields["frustum"][1] == 272 and abi.CAMERA.fields["zfar"][1] == 288
    assert abi.CAMERA.fields["znear"][1] == 292
    assert abi.BVH_NODE.itemsize == 32 and abi.TLAS_NODE.itemsize == 32
    assert abi.TLAS_NODE.fields["left_right"][1] == 12 and abi.TLAS_NODE.fields["instance_idx"][1] == 28


def test_null_ctx_is_an_error_not_a_crash():
    lib = abi.load()
    assert lib.vd_ctx_destroy(None) == abi.VD_ERR_INVALID_ARG
    assert lib.vd_cull_emit_dev(None, None, None, 0, None, 0, None) == abi.VD_ERR_INVALID_ARG
    assert lib.vd_bvh_build(None, None, 0, None, 0, None, 0, None) == abi.VD_ERR_INVALID_ARG
    assert lib.vd_last_gpu_ms(None) < 0


def test_no_cpu_fallback_in_product_package():
    """The product package must not import or call the oracle."""
    pkg = os.path.join(ROOT, "voidin_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "vd_ref_" not in txt and "np_restate" not in txt and "libvd_oracle" not in txt, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


def test_option_ids_of_the_python_mirror_are_the_headers():
    """voidin_amd/abi.py names the VdOption ids by hand; every enumerator of include/voidin_abi.h must be there with its value
    (an option the mirror does not know is one no test or A/B script can reach), and no id may be used twice."""
    import re
    from voidin_amd import abi
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "voidin_abi.h")).read()
    body = text[text.index("typedef enum VdOption"):text.index("} VdOption;")]
    header = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(VD_OPT_[A-Z0-9_]+)\s*=\s*(\d+)", body)}
    count = header.pop("VD_OPT_COUNT_")
    assert header and max(header.values()) < count
    assert len(set(header.values())) == len(header)
    mirror = {"VD_OPT_" + k.replace(".", "_").upper(): v for k, v in abi.OPTIONS.items()}
    assert mirror == header, (sorted(set(header) ^ set(mirror)), {k: (header.get(k), mirror.get(k)) for k in header if header.get(k) != mirror.get(k)})
    assert set(abi.OPTION_ENV.values()) <= set(abi.OPTIONS)


def test_the_boundary_header_is_plain_c():
    """The drop-in boundary is a C ABI (the Rust side binds it through `extern "C"`, a C host includes it): the header must
    parse as C11 on its own - no C++ constructs, no missing includes - with the struct sizes the wire contract fixes."""
    import subprocess
    import tempfile
    src = ('#include "voidin_abi.h"\n'
           "_Static_assert(sizeof(VdInstance) == 144 && sizeof(VdMeshInfo) == 48 && sizeof(VdDrawIndexedIndirect) == 20, \"wire structs\");\n"
           "_Static_assert(sizeof(VdCameraUniform) == 320 && sizeof(VdBvhNode) == 32 && sizeof(VdTlasNode) == 32, \"wire structs\");\n"
           "int main(void) { return vd_version() == 0; }\n")
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "abi_is_c.c")
        open(p, "w").write(src)
        r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), p],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
