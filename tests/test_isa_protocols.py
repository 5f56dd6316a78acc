"""The fence-free hand-offs between workgroups: the TLAS refit climb and the several-workgroup TLAS
chain publish data with relaxed agent-scope atomics and order them with `s_waitcnt vmcnt(0)` instead of release /
acquire fences (an agent-scope release writes back the whole L2 of the XCD: DESIGN.md 3.4).  Under the HIP memory model
that rests on what the compiler EMITS, so the emitted gfx950 code is checked here, on every build:

  * published data leaves as write-through stores (`sc1`) and is re-read L2-coherently (`sc1` loads);
  * the arrival / "readable" atomic that follows the data is preceded by `s_waitcnt vmcnt(0)`;
  * the chunk counts of `mask_scan_kernel` travel as ONE 8-byte {epoch, count} word (self-validating, race-free).

tests/test_gpu_stress.py is the run-time half (repeat and compare byte for byte)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "voidin_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-S", "--cuda-device-only"]


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    out = {}
    d = tmp_path_factory.mktemp("isa")
    for name in ("tlas", "cull"):
        path = str(d / f"{name}.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, os.path.join(CSRC, f"{name}.hip"), "-o", path], check=True, capture_output=True, timeout=600)
        out[name] = open(path).read()
    return out


def kernel_body(text, mangled_fragment):
    """Instructions of the first kernel whose symbol contains `mangled_fragment`."""
    m = re.search(r"^(_Z\w*%s\w*):\s*;.*$" % re.escape(mangled_fragment), text, re.M)
    assert m, f"kernel {mangled_fragment} not found"
    end = text.index(".Lfunc_end", m.end())         # (a kernel may hold several s_endpgm: early exits)
    return [l.strip() for l in text[m.end():end].splitlines() if l.strip() and not l.strip().startswith((";", "."))]


def test_refit_climb_publishes_write_through_and_waits_before_the_handshake(isa):
    body = kernel_body(isa["tlas"], "tlas_refit_up_kernelI10VdTlasNode")
    atomics = [i for i, l in enumerate(body) if l.startswith("global_atomic_add")]
    assert len(atomics) == 2, "announce (+1) and readable (+2)"
    stores = [i for i, l in enumerate(body) if l.startswith("global_store_dword ") and i < atomics[1]]
    assert len(stores) >= 6 and all("sc1" in body[i] for i in stores), "the box must leave as agent-scope write-through stores"
    between = body[max(stores) + 1: atomics[1]]
    assert any(l.startswith("s_waitcnt vmcnt(0)") for l in between), "the box must have reached memory before 'readable' is counted"
    loads = [l for l in body[atomics[1] + 1:] if l.startswith("global_load_dword ")][:6]
    assert len(loads) == 6 and all("sc1" in l for l in loads), "the sibling's box must be read L2-coherently"


def test_several_workgroup_chain_exchanges_tagged_words(isa):
    body = kernel_body(isa["tlas"], "tlas_build_mw_kernelI10VdTlasNode")
    assert any(l.startswith("global_store_dwordx2") and "sc1" in l for l in body), "the {area, slot, tag} key is ONE 8-byte agent-scope store"
    assert any(l.startswith("global_load_dwordx2") and "sc1" in l for l in body), "... polled by agent-scope loads"
    # the merge: slot stores (sc1), then the explicit wait, then the merge counter
    waits = [i for i, l in enumerate(body) if l.startswith("s_waitcnt vmcnt(0)")]
    sc1_stores = [i for i, l in enumerate(body) if l.startswith("global_store_dword ") and "sc1" in l]
    assert sc1_stores and any(any(s < w for s in sc1_stores) and any(s > w for s in sc1_stores) for w in waits)


def test_mask_scan_counts_are_self_validating_words(isa):
    body = kernel_body(isa["cull"], "mask_scan_kernel")
    assert any(l.startswith("global_store_dwordx2") and "sc1" in l for l in body), "{epoch, count} leaves as one 8-byte agent-scope store"
    assert any(l.startswith("global_load_dwordx2") and "sc1" in l for l in body), "... and is accepted only when its tag is this launch's epoch"
    assert not any(l.startswith("buffer_wbl2") for l in body), "no agent-scope release fence (it writes back the XCD's whole L2)"
