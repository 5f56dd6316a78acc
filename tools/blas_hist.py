# needs the tuning build: make -C voidin_amd/csrc tuning && VOIDIN_HIP_LIB=voidin_amd/csrc/libvoidin_hip_tuning.so python tools/blas_hist.py
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import synth
from voidin_amd.runtime import Context
ctx = Context(0)
v, i = synth.knot_mesh(2048, 2048)
n_tri = len(i)//3
d_v = ctx.upload(v); d_n = ctx.empty(2*n_tri*32)
for r in range(2):
    d_i = ctx.upload(i); ctx.bvh_build_dev(d_v, len(v), d_i, n_tri, d_n, 2*n_tri); torch.cuda.synchronize()
buf = np.zeros(2*200000, np.uint32)
ctx.lib.vd_debug_blas_cycles.restype = C.c_int
ctx.lib.vd_debug_blas_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
n = ctx.lib.vd_debug_blas_cycles(ctx.h, buf.ctypes.data, len(buf))
d = buf[:n].reshape(-1,2); cyc = d[:,0].astype(np.float64); N = d[:,1]
print('roots', len(d), 'prims mean', N.mean(), 'min', N.min(), 'max', N.max())
us = cyc/2100.0  # s_memtime ticks: ~2.1 GHz on this part (a subtree takes ~0.6 ms: 1.2 M ticks)
print('block time us: mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f sum(ms) %.1f' % (us.mean(), np.median(us), np.percentile(us,90), np.percentile(us,99), us.max(), us.sum()/1e3))
for lo,hi in [(0,64),(64,128),(128,256),(256,384),(384,513)]:
    m=(N>=lo)&(N<hi)
    if m.any(): print(f'  N in [{lo},{hi}): {m.sum()} roots, mean {us[m].mean():.1f} us, max {us[m].max():.1f}')
