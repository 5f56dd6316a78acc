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
    for n in (0, 1, 7, 130, 10_000_000):
        for world in (1, 2, 3, 8):
            r = [vdist.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            s = vdist.shard_size(n, world)
            assert all(hi - lo <= s for lo, hi in r)


def _mask_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref
        cam, meshes = synth.camera_uniform(), synth.mesh_infos()
        lo, hi = vdist.shard_range(n, rank, world)
        shard = synth.instances(hi - lo, seed=78, offset=lo, scale_range=(0.02, 0.6), extent=600.0, with_inverse=False)
        vis = ref.cull_emit(cam, meshes, shard)["instance_count"].astype(np.uint8)
        S, wps = vdist.shard_size(n, world), vdist.mask_words(vdist.shard_size(n, world))
        bits = np.zeros(wps * 64, np.uint8)
        bits[: hi - lo] = vis
        words = np.packbits(bits.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.int64)   # what vd_cull_mask_dev writes
        mask = torch.from_numpy(words.copy())
        all_masks = torch.zeros(wps * world, dtype=torch.int64)
        dist.all_gather_into_tensor(all_masks, mask)
        ids = torch.zeros(S, dtype=torch.int32)
        ids[: hi - lo] = torch.from_numpy(shard["mesh"].astype(np.int32))
        all_ids = torch.zeros(S * world, dtype=torch.int32)
        dist.all_gather_into_tensor(all_ids, ids)
        q.put((rank, all_masks.numpy().tobytes(), all_ids.numpy().tobytes()))
    finally:
        dist.destroy_process_group()


def test_bitmask_exchange_reconstructs_the_scene(oracle):
    """world-size-2 gloo run of the mask wire format: the gathered masks + replicated mesh ids
    decode (here in numpy; on the GPU by vd_expand_mask_dev) to the single-rank compaction."""
    world, n = 2, 20_003
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    inst = synth.instances(n, seed=78, scale_range=(0.02, 0.6), extent=600.0, with_inverse=False)
    want, wn = oracle.compact(oracle.cull_emit(cam, meshes, inst))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mask_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    S, wps = vdist.shard_size(n, world), vdist.mask_words(vdist.shard_size(n, world))
    for rank, mblob, iblob in res:
        words = np.frombuffer(mblob, np.uint8)
        bits = np.unpackbits(words, bitorder="little").reshape(world, wps * 64)
        ids = np.frombuffer(iblob, np.int32).reshape(world, S)
        surv = np.concatenate([r * S + np.nonzero(bits[r][:S])[0] for r in range(world)])
        surv = surv[surv < n]
        assert np.array_equal(surv, want["base_instance"][:wn])
        mesh_of = np.concatenate([ids[r] for r in range(world)])
        assert np.array_equal(meshes["index_count"][mesh_of[surv]], want["vertex_count"][:wn])


def _blas_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref
        meshes = [synth.uv_sphere(1.0, 3), synth.triangle_soup(64), synth.knot_mesh(24, 8), synth.plane_mesh(6)]
        built = []
        def build(v, i):                       # stand-in producer on CPU: the oracle's builder
            built.append(len(i))
            return ref.bvh_build(v, i)
        res = vdist.build_blas_batch(build, meshes)
        q.put((rank, len(built), [(n.tobytes(), i.tobytes()) for n, i in res]))
    finally:
        dist.destroy_process_group()


def test_blas_batch_is_partitioned_by_mesh_and_replicated(oracle):
    """Different meshes build on different ranks; every rank ends with every BLAS (gloo, world 2)."""
    meshes = [synth.uv_sphere(1.0, 3), synth.triangle_soup(64), synth.knot_mesh(24, 8), synth.plane_mesh(6)]
    want = [oracle.bvh_build(v, i) for v, i in meshes]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_blas_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n_built, blobs in res:
        assert n_built == 2, "each rank builds its own half of the meshes"
        for (nb, ib), (wn, wi) in zip(blobs, want):
            assert nb == wn.tobytes() and ib == wi.tobytes(), f"rank {rank}"


def _records_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref
        cam, meshes = synth.camera_uniform(), synth.mesh_infos()
        lo, hi = vdist.shard_range(n, rank, world)
        shard = synth.instances(hi - lo, seed=81, offset=lo, scale_range=(0.02, 0.6), extent=600.0, with_inverse=False)
        vis = ref.cull_emit(cam, meshes, shard)["instance_count"] == 1
        idx = (np.nonzero(vis)[0] + lo).astype(np.uint32)                 # what vd_mask_to_indices_dev writes for this shard
        counts = vdist.allgather_counts(torch.tensor([len(idx)], dtype=torch.int32))
        out = torch.zeros(int(counts.sum()) * 4 + 16, dtype=torch.uint8)
        total = vdist.allgather_records(torch.from_numpy(idx.view(np.uint8).copy()), counts, out, 4)
        # the replicated instance -> mesh table, as ShardedVisibility builds it (unsigned clamp, narrowest width)
        table = vdist.mesh_id_table(torch.from_numpy(shard.view(np.uint8).reshape(-1).copy()), hi - lo, len(meshes), vdist.shard_size(n, world))
        all_ids = torch.zeros(table.numel() * world, dtype=torch.uint8)
        dist.all_gather_into_tensor(all_ids, table)
        q.put((rank, total, out[: total * 4].numpy().tobytes(), all_ids.numpy().tobytes()))
    finally:
        dist.destroy_process_group()


def test_indices_wire_format_reconstructs_the_scene(oracle):
    """--gather indices on CPU (gloo, world 3): exact-size exchange of u32 survivor indices + the replicated mesh-id
    table give every rank what vd_indices_to_draws_dev needs; rebuilt here in numpy == single-rank compaction."""
    world, n = 3, 12_345
    cam, meshes = synth.camera_uniform(), synth.mesh_infos()
    inst = synth.instances(n, seed=81, scale_range=(0.02, 0.6), extent=600.0, with_inverse=False)
    want, wn = oracle.compact(oracle.cull_emit(cam, meshes, inst))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_records_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, total, iblob, tblob in res:
        idx = np.frombuffer(iblob, np.uint32)
        assert total == wn and np.array_equal(idx, want["base_instance"][:wn]), f"rank {rank}"
        ids = np.frombuffer(tblob, np.uint8)                                # 16 meshes: one byte per instance, row = global index
        assert np.array_equal(meshes["index_count"][ids[idx]], want["vertex_count"][:wn])


def test_mesh_id_table_clamps_unsigned():
    """ids >= 2^31 must clamp to n_mesh - 1 like `min(u32 id, n_mesh - 1)` in the kernels, not to 0."""
    inst = synth.instances(7, seed=3, with_inverse=False)
    inst["mesh"] = np.array([0, 5, 15, 16, 0x7fffffff, 0x80000000, 0xffffffff], dtype=np.uint32)
    raw = torch.from_numpy(inst.view(np.uint8).reshape(-1).copy())
    for n_mesh, width in ((16, 1), (300, 2), (70000, 4)):
        t = vdist.mesh_id_table(raw, 7, n_mesh, 8).numpy()
        assert vdist.id_width(n_mesh) == width and t.size == 8 * width
        got = t.view({1: np.uint8, 2: np.uint16, 4: np.uint32}[width])
        want = np.minimum(inst["mesh"].astype(np.uint64), n_mesh - 1)
        assert np.array_equal(got[:7].astype(np.uint64), want) and got[7] == 0
