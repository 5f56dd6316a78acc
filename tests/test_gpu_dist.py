"""End-to-end N>1 path on ONE GPU: two ranks (gloo rendezvous, both on cuda:0) run
ShardedVisibility — local cull to a bitmask, mask all-gather, local expansion — and every rank must
end with the single-GPU compacted draw list, bit for bit.  (RCCL itself needs one GPU per rank;
the 8-GPU run is the driver's.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from voidin_amd import synth

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q, n_mesh=16):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from voidin_amd import dist as vdist
        from voidin_amd.runtime import Context
        torch.cuda.set_device(0)
        ctx = Context(0)
        cam, meshes = synth.camera_uniform(), synth.mesh_infos(n_mesh)
        lo, hi = vdist.shard_range(n, rank, world)
        shard = synth.instances(hi - lo, n_mesh=n_mesh, seed=79, offset=lo, scale_range=(0.02, 0.6), extent=600.0, with_inverse=False)
        d_m, d_i = ctx.upload(meshes), ctx.upload(shard)
        sv = vdist.ShardedVisibility(ctx, n, d_m, len(meshes), d_i)
        d_out = ctx.empty(n * 20)
        d_cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
        for _ in range(2):
            sv.step(cam, d_out, d_cnt)
        torch.cuda.synchronize()
        cnt = int(d_cnt[0].item())
        q.put((rank, cnt, d_out[: cnt * 20].cpu().numpy().tobytes()))
        ctx.close()
    except Exception as e:  # surface the failure in the parent
        q.put((rank, -1, repr(e).encode()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,n_mesh", [(2, 200_003, 16), (3, 64_000, 16), (2, 50_001, 300)])
def test_sharded_visibility_two_ranks_one_gpu(oracle, world, n, n_mesh):
    """n_mesh = 300: the replicated instance->mesh table holds 2-byte ids (gathered as bytes)."""
    cam, meshes = synth.camera_uniform(), synth.mesh_infos(n_mesh)
    inst = synth.instances(n, n_mesh=n_mesh, seed=79, scale_range=(0.02, 0.6), extent=600.0, with_inverse=False)
    want, wn = oracle.compact(oracle.cull_emit(cam, meshes, inst))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q, n_mesh)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=300)
    for rank, cnt, blob in res:
        assert cnt >= 0, blob
        assert cnt == wn and blob == want[:wn].tobytes(), f"rank {rank}"


def _blas_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from voidin_amd import dist as vdist
        from voidin_amd.runtime import Context
        torch.cuda.set_device(0)
        ctx = Context(0)
        meshes = [synth.uv_sphere(1.0, 10), synth.triangle_soup(64), synth.knot_mesh(96, 24), synth.knot_mesh(40, 16, seed=5)]
        res = vdist.build_blas_batch(ctx.bvh_build, meshes, device="cuda")
        q.put((rank, [(n.tobytes(), i.tobytes()) for n, i in res]))
        ctx.close()
    except Exception as e:
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_blas_batch_two_ranks_one_gpu(oracle):
    """Scene load with many meshes: rank r builds meshes r, r+2, ... on the GPU, owners broadcast; every
    rank holds every BLAS, bit-exact against the oracle's builder."""
    meshes = [synth.uv_sphere(1.0, 10), synth.triangle_soup(64), synth.knot_mesh(96, 24), synth.knot_mesh(40, 16, seed=5)]
    want = [oracle.bvh_build(v, i) for v, i in meshes]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_blas_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(2)]
    for p in procs:
        p.join(timeout=300)
    for rank, blobs in res:
        assert isinstance(blobs, list), blobs
        for (nb, ib), (wn, wi) in zip(blobs, want):
            assert ib == wi.tobytes(), f"rank {rank}: index permutation"
            from conftest import fields_equal
            got = np.frombuffer(nb, dtype=wn.dtype)
            assert len(got) == len(wn) and fields_equal(got, wn), f"rank {rank}: nodes"
