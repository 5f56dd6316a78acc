"""N>1 path on CPU: world_size-2 and -3 gloo runs of the shard partition + draw-list exchange.
The per-rank compaction is produced by the CPU oracle here (tests may use it as the checker /
stand-in producer); the exchange code is the product's voidin_amd.dist."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from voidin_amd import abi, synth
from voidin_amd import dist as vdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref
        cam, meshes = synth.camera_uniform(), synth.mesh_infos()
        lo, hi = vdist.shard_range(n, rank, world)
        shard = synth.instances(hi - lo, seed=77, offset=lo, scale_range=(0.02, 0.6), extent=600.0, with_inverse=False)
        draws = ref.cull_emit(cam, meshes, shard)
        draws["base_instance"] += np.uint32(lo)          # what vd_cull_compact_shard_dev writes
        comp, cnt = ref.compact(draws)
        local = torch.from_numpy(comp.view(np.uint8).reshape(-1).copy())
        counts = vdist.allgather_counts(torch.tensor([cnt], dtype=torch.int32))
        out = torch.zeros(int(counts.sum()) * 20 + 64, dtype=torch.uint8)
        total = vdist.allgather_draws(local, counts, out)
        q.put((rank, total, out[: total * 20].numpy().tobytes(), [int(c) for c in counts]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 30_001), (3, 10_000)])
def test_sharded_exchange_equals_single_rank(oracle, world, n):
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    inst = synth.instances(n, seed=77, scale_range=(0.02, 0.6), extent=600.0, with_inverse=False)
    want, wn = oracle.compact(oracle.cull_emit(cam, meshes, inst))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, total, blob, counts in res:
        assert total == wn and sum(counts) == wn
        assert blob == want[:wn].tobytes(), f"rank {rank}: gathered list differs from single-rank compaction"


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 10_000_000):
        for world in (1, 2, 3, 8):
            r = [vdist.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
