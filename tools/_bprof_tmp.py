import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
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
d = buf[:n].reshape(-1,2)
tot = (d[:,0] & 0xffff).astype(np.float64) * 256
t_root = ((d[:,0] >> 16) & 0xff).astype(np.float64) * 4096
t_waves = ((d[:,0] >> 24) & 0xff).astype(np.float64) * 4096
N = d[:,1] & 1023; t_lane = ((d[:,1] >> 10) & 1023).astype(np.float64) * 4096; t_renum = (d[:,1] >> 20).astype(np.float64) * 4096; nw = N
print("roots", len(d), "N mean", N.mean(), "wide nodes/root mean", nw.mean())
print("cycles: total mean %.0f, root node done at %.0f, wave phase done at %.0f, lane phase done at %.0f, renumber scan done at %.0f" % (tot.mean(), t_root.mean(), t_waves.mean(), t_lane.mean(), t_renum.mean()))
idle0 = d[:,1] & 1023; tsb = ((d[:,1] >> 10) & 2047).astype(np.float64) * 4096; tle = (d[:,1] >> 21).astype(np.float64) * 4096
print("small phase: wave0 idle polls %.1f, begins at %.0f, last batch ends at %.0f" % (idle0.mean(), tsb.mean(), tle.mean()))
m = (N > 300) & (N < 400)
print(" N 300-400: total %.0f root %.0f waves %.0f wide nodes %.1f" % (tot[m].mean(), t_root[m].mean(), t_waves[m].mean(), nw[m].mean()))
