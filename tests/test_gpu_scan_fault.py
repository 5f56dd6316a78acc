"""The ordered compaction scan (decoupled look-back, voidin_amd/csrc/vd_common.hpp) never waits forever: a workgroup that
never publishes its total ends the launch with an error - VD_ERR_HIP from the host-pointer entry point; count 0, a zeroed
padded buffer and a sticky fault for the device-pointer form - and the next call on the same context is clean.  The lost workgroup is simulated by vd_debug_scan_fault, a hook that exists only in the tuning build
of the library (make -C voidin_amd/csrc tuning; -DVD_TUNING), so this test runs in a child process that loads that build."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CSRC = os.path.join(ROOT, "voidin_amd", "csrc")
TUNING = os.path.join(CSRC, "libvoidin_hip_tuning.so")

CHILD = textwrap.dedent("""
    import ctypes as C, sys, time
    import numpy as np
    sys.path.insert(0, %r)
    from oracle import ref
    from voidin_amd import abi, synth
    from voidin_amd.runtime import Context
    ctx = Context(0)
    lib = ctx.lib
    lib.vd_debug_scan_fault.restype = C.c_int
    lib.vd_debug_scan_fault.argtypes = [C.c_void_p, C.c_int]
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    inst = synth.instances(300_000, scale_range=(0.02, 0.6), extent=600.0)       # fused form: 1024-instance tiles, 293 of them
    want, wn = ref.compact(ref.cull_emit(cam, meshes, inst))
    got, n = ctx.cull_compact(cam, meshes, inst)
    assert n == wn and got[:n].tobytes() == want[:wn].tobytes()
    for tile in (0, 17, 292):
        assert lib.vd_debug_scan_fault(ctx.h, tile) == 0
        t0 = time.time()
        try:
            ctx.cull_compact(cam, meshes, inst)
        except RuntimeError as e:
            assert "VD_ERR_HIP" in str(e) and "gave up" in str(e), str(e)
        else:
            raise AssertionError("a lost workgroup went unnoticed (tile %%d)" %% tile)
        assert time.time() - t0 < 60.0
        assert lib.vd_debug_scan_fault(ctx.h, -1) == 0
        got, n = ctx.cull_compact(cam, meshes, inst)                           # same context, next call: clean
        assert n == wn and got[:n].tobytes() == want[:wn].tobytes(), tile
    # the device-pointer form (what a frame loop calls): count 0 and - with pad_tail - an all-zero buffer, so neither consumer of
    # visibility.rs:188-192 draws anything; the NEXT call on the context says VD_ERR_HIP once, the one after is clean (ADVICE r5)
    import torch
    n = len(inst)
    d_m, d_i = ctx.upload(meshes), ctx.upload(inst)
    d_out, d_cnt = ctx.empty(n * 20), torch.full((4,), 7, dtype=torch.int32, device="cuda")
    d_out.fill_(0xAB)
    assert lib.vd_debug_scan_fault(ctx.h, 17) == 0
    t0 = time.time()
    ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, n, d_out, d_cnt, True)
    torch.cuda.synchronize()
    assert time.time() - t0 < 30.0
    assert int(d_cnt[0].item()) == 0 and not bool(d_out[: n * 20].any())
    assert lib.vd_debug_scan_fault(ctx.h, -1) == 0
    try:
        ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, n, d_out, d_cnt, True)
    except RuntimeError as e:
        assert "VD_ERR_HIP" in str(e) and "gave up" in str(e), str(e)
    else:
        raise AssertionError("the fault of the previous launch was not reported")
    ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, n, d_out, d_cnt, True)
    torch.cuda.synchronize()
    wp, wpn = ref.compact(ref.cull_emit(cam, meshes, inst), pad_tail=True)
    assert int(d_cnt[0].item()) == wpn and d_out.cpu().numpy()[: n * 20].tobytes() == wp.tobytes()
    print("scan fault test OK")
""")


def test_lost_workgroup_is_an_error_not_a_hang():
    src_newest = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")))
    if not os.path.exists(TUNING) or os.path.getmtime(TUNING) < src_newest:
        subprocess.run(["make", "-C", CSRC, "tuning"], check=True, capture_output=True, timeout=1200)
    env = dict(os.environ, VOIDIN_HIP_LIB=TUNING)
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "scan fault test OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
