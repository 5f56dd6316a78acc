"""Where phase B's time goes, by node size class and step: the -DVD_TUNING build counts wave-cycles (s_memtime
deltas, one wave = one counter) inside blas_small_kernel; this prints them for the 8 388 608-triangle knot as shares of all waves'
lifetimes and as milliseconds of the kernel (share x phase-B time of the same build).
    make -C voidin_amd/csrc tuning && VOIDIN_HIP_LIB=$PWD/voidin_amd/csrc/libvoidin_hip_tuning.so python tools/blas_small_classes.py"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import synth  # noqa: E402
from voidin_amd.runtime import Context  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--u", type=int, default=2048)
ap.add_argument("--v", type=int, default=2048)
args = ap.parse_args()
ctx = Context(0)
lib = ctx.lib
if not hasattr(lib, "vd_debug_blas_small_classes"):
    sys.exit("needs the tuning build: VOIDIN_HIP_LIB=.../libvoidin_hip_tuning.so")
lib.vd_debug_blas_small_classes.restype = C.c_int
lib.vd_debug_blas_small_classes.argtypes = [C.c_void_p, C.c_void_p]
v, i = synth.knot_mesh(args.u, args.v)
n_tri = len(i) // 3
d_v, d_n = ctx.upload(v), ctx.empty(2 * n_tri * 32)
for _ in range(2):
    d_i = ctx.upload(i)
    torch.cuda.synchronize()
    ctx.bvh_build_dev(d_v, len(v), d_i, n_tri, d_n, 2 * n_tri)
st = ctx.bvh_last_build_stats()
out = np.zeros(64, dtype=np.uint64)
assert lib.vd_debug_blas_small_classes(ctx.h, out.ctypes.data) == 0
life = float(out[40])
ms_b = st["ms_phase_b"]
print(f"{n_tri} triangles, tuning build: phase B {ms_b:.2f} ms (the counters' own atomics included), {int(out[44])} waves; "
      f"{st['n_small_roots']} subtree roots")
names = ["<= 32 (lane groups, per batch)", "33..64 (registers)", "65..128", "129..256", "257..512"]
steps = ["setup", "21 trials", "evaluation", "final shuffle", "children+handover"]
print(f"{'class':32s} {'nodes':>9s} " + " ".join(f"{s_:>18s}" for s_ in steps) + f" {'class total':>14s}")
tot_all = 0.0
for c, nm in enumerate(names):
    row = out[c * 8: c * 8 + 8].astype(np.float64)
    cells = " ".join(f"{row[k] / life * 100:6.2f} % {row[k] / life * ms_b:6.2f} ms" for k in range(5))
    tot = row[:5].sum()
    tot_all += tot
    extra = f"  ({int(row[6])} batches)" if c == 0 else (f"  (block-wide roots: {row[6] / life * ms_b:.2f} ms)" if row[6] else "")
    print(f"{nm:32s} {int(row[5]):9d} {cells} {tot / life * 100:6.2f} % {tot / life * ms_b:5.2f} ms{extra}")
other = [("load of the subtree image", out[42]), ("list drain wait + DFS renumber + copy-out", out[43])]
for nm, v_ in other:
    print(f"{nm:42s} {float(v_) / life * 100:6.2f} % {float(v_) / life * ms_b:6.2f} ms")
print(f"{'node bodies in all':42s} {tot_all / life * 100:6.2f} %; the rest of a wave's life is polling for work, barriers of the block-wide root and the tail")
