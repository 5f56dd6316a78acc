"""Multi-GPU sharding of the cull path (NEW — the reference is single-GPU).

Instances are independent (shaders/emit_draws.wgsl:38-63 touches only slot i), so rank r owns the
contiguous shard [r*S, min(N, (r+1)*S)), S = ceil(N / world).  Every rank ends each frame with the
ordered, compacted draw list of the WHOLE scene, bit-identical to the single-GPU list.

xGMI (≈153 GB/s per link) is ~40x slower than HBM, so the exchange is sized for it:

  * ShardedVisibility (default): the wire format is ONE BIT per instance.  Each rank culls its
    shard into a bitmask (vd_cull_mask_dev), the masks are all-gathered (1.25 MB per 10 M
    instances), and every rank expands the concatenated masks into the ordered draw list locally
    (vd_expand_mask_dev) from a replicated instance->mesh table that is all-gathered once per
    scene (mesh assignment is static; only transforms animate).
  * allgather_draws: the literal exchange of the 20-byte commands (exact-size one-shot direct
    all-gather: every rank sends its list straight into every peer's final buffer, one xGMI link
    per peer) — kept for consumers that do not hold the instance->mesh table.

Two hosts of the same exchange:

  * RcclVisibility — the product path: the exchange lives in the C ABI (vd_dist_*, voidin_amd/csrc/dist.hip):
    cull_mask -> ncclAllGather -> expand_mask enqueued on the context's stream by ONE call, RCCL bound by the
    library itself.  torch.distributed (any backend, gloo is enough) only carries the 128-byte communicator id.
  * ShardedVisibility — the same steps with torch.distributed collectives between the two kernels; runs on gloo,
    so it is what the CPU / shared-GPU tests exercise (RCCL refuses two ranks on one device).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

DRAW_BYTES = 20


def shard_size(n: int, world: int) -> int:
    return (n + world - 1) // world


def shard_range(n: int, rank: int, world: int):
    """Uniform contiguous shards [r*S, min(n, (r+1)*S)), S = ceil(n / world)."""
    s = shard_size(n, world)
    return min(n, rank * s), min(n, (rank + 1) * s)


def mask_words(shard: int) -> int:
    return (shard + 63) // 64


def id_width(n_mesh: int) -> int:
    """Bytes per entry of the instance -> mesh table: the rule of launch_mask_pass (voidin_amd/csrc/cull.hip)."""
    return 1 if n_mesh <= 256 else (2 if n_mesh <= 65536 else 4)


def mesh_id_table(d_inst_u8: torch.Tensor, n_local: int, n_mesh: int, rows: int) -> torch.Tensor:
    """`rows` table entries of id_width(n_mesh) bytes each (flat uint8): min(u32 instance.mesh, n_mesh - 1) for the
    first n_local instances of the shard, 0 beyond.  The clamp is UNSIGNED, as in the kernels and the oracle
    (emit path: `min(li.mesh, n_mesh - 1u)`): ids >= 2^31 clamp to n_mesh - 1, not to 0."""
    w = id_width(n_mesh)
    ids = torch.zeros(rows, dtype=torch.int64, device=d_inst_u8.device)
    if n_local:
        col = d_inst_u8[: n_local * 144].view(torch.int32).view(-1, 36)[:, 32]
        ids[:n_local] = (col.to(torch.int64) & 0xFFFFFFFF).clamp(max=n_mesh - 1)
    return ids.to(torch.int32).view(torch.uint8).view(-1, 4)[:, :w].contiguous().view(-1)


class ShardedVisibility:
    """Per-frame visibility of a scene whose instances are sharded by rank.  Four consumers (bench.py --gather):

      step          full list on every rank; wire = 1 bit per instance (bitmask all-gather + local expansion)
      step_indices  full list on every rank; wire = 4 B per survivor (SURVEY.md 8e option) + local rebuild
      step_draws    full list on every rank; wire = the 20-byte commands themselves (the literal north-star exchange)
      step_shard    every rank keeps the compacted list of ITS shard only (global base_instance): no exchange
    """

    def __init__(self, ctx, n_total: int, d_meshes, n_mesh: int, d_inst_shard, group=None):
        self.ctx, self.group = ctx, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_total, self.n_mesh, self.d_meshes = n_total, n_mesh, d_meshes
        self.S = shard_size(n_total, self.world)
        self.lo, self.hi = shard_range(n_total, self.rank, self.world)
        self.n_local = self.hi - self.lo
        self.wps = mask_words(self.S)
        dev = d_inst_shard.device
        self.d_inst = d_inst_shard
        # local mask (zero padded to wps words) and the gathered masks of all shards
        self.d_mask = torch.zeros(self.wps, dtype=torch.int64, device=dev)
        self.d_mask_all = torch.zeros(self.wps * self.world, dtype=torch.int64, device=dev)
        # replicated instance -> mesh table: column 32 of the 36-dword instance records, gathered once (as bytes: the
        # RCCL backend has no 16-bit integer type), stored at the narrowest width the mesh table allows (the expansion
        # pass reads it every frame); global instance i sits at row i (shards are S rows each)
        self.id_bytes = id_width(n_mesh)
        ids = mesh_id_table(d_inst_shard, self.n_local, n_mesh, self.S)
        self.d_mesh_ids = torch.empty(self.S * self.world * self.id_bytes, dtype=torch.uint8, device=dev)
        if self.world > 1:
            dist.all_gather_into_tensor(self.d_mesh_ids, ids, group=group)
        else:
            self.d_mesh_ids.copy_(ids)
        self.d_idx = self.d_idx_all = self.d_local = None
        self.d_cnt = torch.zeros(4, dtype=torch.int32, device=dev)

    def step(self, camera, d_out, d_count):
        """d_out: n_total * 20 bytes; d_count: int32[>=1].  Enqueues on the ctx stream."""
        self.ctx.cull_mask_dev(camera, self.d_meshes, self.n_mesh, self.d_inst, self.n_local, self.d_mask)
        if self.world > 1:
            dist.all_gather_into_tensor(self.d_mask_all, self.d_mask, group=self.group)
            masks = self.d_mask_all
        else:
            masks = self.d_mask
        self.ctx.expand_mask_dev(masks, self.n_total, self.S, self.d_mesh_ids, self.d_meshes, self.n_mesh, d_out, d_count,
                                 id_bytes=self.id_bytes)

    def step_shard(self, camera, d_out_local, d_count_local):
        """Own shard only: d_out_local holds n_local commands with GLOBAL base_instance.  No exchange."""
        self.ctx.cull_compact_dev(camera, self.d_meshes, self.n_mesh, self.d_inst, self.n_local, d_out_local, d_count_local,
                                  False, self.lo)

    def step_indices(self, camera, d_out, d_count):
        """Survivor indices (4 B each) are exchanged, every rank rebuilds all commands.  The sizes are data dependent,
        so the step reads the counts back (one host round trip)."""
        dev = self.d_inst.device
        if self.d_idx is None:
            self.d_idx = torch.empty(max(self.S, 4), dtype=torch.int32, device=dev)
            self.d_idx_all = torch.empty(max(self.S * self.world, 4), dtype=torch.int32, device=dev)
        self.ctx.cull_mask_dev(camera, self.d_meshes, self.n_mesh, self.d_inst, self.n_local, self.d_mask)
        self.ctx.mask_to_indices_dev(self.d_mask, self.n_local, self.lo, self.d_idx, self.d_cnt)
        if self.world > 1:
            counts = allgather_counts(self.d_cnt[:1], self.group)
            total = allgather_records(self.d_idx.view(torch.uint8), counts, self.d_idx_all.view(torch.uint8), 4, self.group)
            src = self.d_idx_all
        else:
            total, src = int(self.d_cnt[0].item()), self.d_idx
        self.ctx.indices_to_draws_dev(src, total, self.d_mesh_ids, self.S * self.world, self.d_meshes, self.n_mesh, d_out,
                                      id_bytes=self.id_bytes)
        d_count[:1].fill_(total)

    def step_draws(self, camera, d_out, d_count):
        """The literal exchange: every rank compacts its shard to 20-byte commands and they are all-gathered."""
        dev = self.d_inst.device
        if self.d_local is None:
            self.d_local = torch.empty(max(self.S, 1) * DRAW_BYTES, dtype=torch.uint8, device=dev)
        self.step_shard(camera, self.d_local, self.d_cnt)
        if self.world > 1:
            counts = allgather_counts(self.d_cnt[:1], self.group)
            total = allgather_records(self.d_local, counts, d_out, DRAW_BYTES, self.group)
        else:
            total = int(self.d_cnt[0].item())
            d_out[: total * DRAW_BYTES].copy_(self.d_local[: total * DRAW_BYTES])
        d_count[:1].fill_(total)


class RcclVisibility:
    """The sharded visibility step behind the C ABI: `vd_dist_step_full_dev` / `vd_dist_step_draws_dev`
    (include/voidin_abi.h, "Multi-GPU exchange over RCCL").  Same interface as ShardedVisibility.

    The communicator id is made by rank 0 (vd_dist_unique_id) and broadcast through `group` (a torch.distributed
    group of any backend) - or passed in as `unique_id` by a host that has its own channel.  world = 1 works without
    any process group: a one-rank RCCL communicator, the functional check of the path on one GPU."""

    def __init__(self, ctx, n_total: int, d_meshes, n_mesh: int, d_inst_shard, group=None, unique_id: bytes | None = None,
                 rank: int | None = None, world: int | None = None):
        import ctypes as C

        from . import abi
        self.ctx, self.lib = ctx, ctx.lib
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world, self.rank = world, rank
        self.n_total, self.n_mesh, self.d_meshes = n_total, n_mesh, d_meshes
        self.S = shard_size(n_total, world)
        self.lo, self.hi = shard_range(n_total, rank, world)
        self.n_local = self.hi - self.lo
        self.d_inst = d_inst_shard
        if unique_id is None:
            buf = (C.c_ubyte * abi.VD_DIST_ID_BYTES)()
            if rank == 0:
                ctx._chk(self.lib.vd_dist_unique_id(C.addressof(buf)))
            if world > 1:
                t = torch.tensor(list(bytes(buf)), dtype=torch.uint8)
                backend = dist.get_backend(group)
                t = t.cuda() if backend == "nccl" else t
                dist.broadcast(t, src=dist.get_global_rank(group, 0) if group else 0, group=group)
                unique_id = bytes(t.cpu().tolist())
            else:
                unique_id = bytes(buf)
        assert len(unique_id) == abi.VD_DIST_ID_BYTES
        h = C.c_void_p()
        idb = C.create_string_buffer(unique_id, abi.VD_DIST_ID_BYTES)
        ctx._chk(self.lib.vd_dist_create(ctx.h, C.addressof(idb), rank, world, C.byref(h)))
        self.h = h
        ctx._chk(self.lib.vd_dist_set_scene_dev(self.h, abi.ptr(d_inst_shard), self.n_local, n_total, n_mesh))
        self.info = abi.DistInfo()
        ctx._chk(self.lib.vd_dist_info(self.h, C.byref(self.info)))
        self.wps, self.id_bytes = self.info.mask_words_per_shard, self.info.id_bytes
        self._abi, self._C = abi, C

    def _cam(self, camera):
        import numpy as np
        return np.ascontiguousarray(camera, dtype=self._abi.CAMERA).reshape(1)

    def step(self, camera, d_out, d_count):
        """Full list on every rank; wire = 1 bit per instance.  One call, three enqueues on the ctx stream."""
        cam = self._cam(camera)
        self.ctx._chk(self.lib.vd_dist_step_full_dev(self.h, cam.ctypes.data, self._abi.ptr(self.d_meshes), self.n_mesh,
                                                     self._abi.ptr(self.d_inst), self._abi.ptr(d_out), self._abi.ptr(d_count)))

    def step_draws(self, camera, d_out, d_count):
        """Full list on every rank; wire = the 20-byte commands (counts read back on the host)."""
        cam = self._cam(camera)
        self.ctx._chk(self.lib.vd_dist_step_draws_dev(self.h, cam.ctypes.data, self._abi.ptr(self.d_meshes), self.n_mesh,
                                                      self._abi.ptr(self.d_inst), self._abi.ptr(d_out), self._abi.ptr(d_count)))

    def step_indices(self, camera, d_out, d_count):
        """Full list on every rank; wire = 4 B per survivor (counts read back on the host)."""
        cam = self._cam(camera)
        self.ctx._chk(self.lib.vd_dist_step_indices_dev(self.h, cam.ctypes.data, self._abi.ptr(self.d_meshes), self.n_mesh,
                                                        self._abi.ptr(self.d_inst), self._abi.ptr(d_out), self._abi.ptr(d_count)))

    def step_shard(self, camera, d_out_local, d_count_local):
        self.ctx.cull_compact_dev(camera, self.d_meshes, self.n_mesh, self.d_inst, self.n_local, d_out_local, d_count_local,
                                  False, self.lo)

    def allgather(self, d_send, d_recv, bytes_per_rank: int):
        self.ctx._chk(self.lib.vd_dist_allgather_dev(self.h, self._abi.ptr(d_send), self._abi.ptr(d_recv), bytes_per_rank))

    def cull_to_mask(self, camera):
        """The first leg alone (bench breakdown): own shard -> the VdDist's mask buffer."""
        self.ctx.cull_mask_dev(camera, self.d_meshes, self.n_mesh, self.d_inst, self.n_local, self.info.d_mask)

    def allgather_masks(self):
        self.allgather(self.info.d_mask, self.info.d_mask_all, self.wps * 8)

    def expand_all(self, d_out, d_count):
        self.ctx.expand_mask_dev(self.info.d_mask_all, self.n_total, self.S, self.info.d_mesh_ids, self.d_meshes, self.n_mesh,
                                 d_out, d_count, id_bytes=self.id_bytes)

    def close(self):
        if getattr(self, "h", None):
            self.lib.vd_dist_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def allgather_counts(local_count: torch.Tensor, group=None) -> torch.Tensor:
    """local_count: 1-element int32/int64 tensor on the compute device -> [world] int64 (host)."""
    world = dist.get_world_size(group)
    mine = local_count.reshape(1).to(torch.int64)
    out = torch.empty(world, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.cpu()


def allgather_records(local_u8: torch.Tensor, counts: torch.Tensor, out_u8: torch.Tensor, record_bytes: int, group=None):
    """Place rank q's first counts[q] records at byte offset record_bytes*sum(counts[:q]) of out_u8 on every rank:
    exact-size one-shot direct all-gather (every rank sends its list straight into every peer's final buffer, one
    xGMI link per peer).  local_u8 / out_u8 are flat uint8 tensors on the compute device."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [int(c) for c in counts]
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    total = offs[-1]
    assert out_u8.numel() >= total * record_bytes
    mine = local_u8[: counts[rank] * record_bytes]
    out_u8[offs[rank] * record_bytes: offs[rank + 1] * record_bytes].copy_(mine)
    if world == 1:
        return total
    # gloo has no device-tensor send/recv: ranks that share a GPU in the functional tests stage through the host
    staged = mine.is_cuda and dist.get_backend(group) == "gloo"
    send_buf = mine.cpu() if staged else mine
    ops, landed = [], []
    for step in range(1, world):
        dst = (rank + step) % world
        src = (rank - step) % world
        if counts[rank]:
            ops.append(dist.P2POp(dist.isend, send_buf, dist.get_global_rank(group, dst) if group else dst, group))
        if counts[src]:
            view = out_u8[offs[src] * record_bytes: offs[src + 1] * record_bytes]
            buf = torch.empty(view.numel(), dtype=torch.uint8) if staged else view
            if staged:
                landed.append((view, buf))
            ops.append(dist.P2POp(dist.irecv, buf, dist.get_global_rank(group, src) if group else src, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for view, buf in landed:
        view.copy_(buf)
    return total


def allgather_draws(local_draws_u8: torch.Tensor, counts: torch.Tensor, out_u8: torch.Tensor, group=None):
    """allgather_records for 20-byte DrawIndexedIndirect commands."""
    return allgather_records(local_draws_u8, counts, out_u8, DRAW_BYTES, group)


def build_blas_batch(build_fn, meshes, group=None, device=None):
    """Scene-load step for many meshes (SURVEY.md §8e "replicas only", §8f N3): one BLAS build does not
    shard — every split of BvhBuilder is a global, order-dependent pass over its segment
    (crates/bvh/src/blas.rs:135-182) — but different meshes are independent, so rank r builds meshes
    r, r + world, ... and the results are broadcast from their owners.

    build_fn(vertices (V,3) f32, indices (3T,) u32) -> (nodes: VdBvhNode array, permuted indices); the
    product passes Context.bvh_build.  `meshes` is the same list on every rank (the asset files are
    replicated).  Returns [(nodes, indices)] for ALL meshes on every rank, in input order.
    device: where the exchange buffers live ("cuda" for RCCL, "cpu" for gloo); default by backend."""
    import numpy as np
    from . import abi
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if device is None:
        device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = {k: build_fn(v, i) for k, (v, i) in enumerate(meshes) if k % world == rank}
    out = []
    for k, (v, i) in enumerate(meshes):
        owner = k % world
        src = dist.get_global_rank(group, owner) if group else owner
        n_idx = int(np.asarray(i).size)
        n_nodes = torch.tensor([len(mine[k][0]) if owner == rank else 0], dtype=torch.int64, device=device)
        dist.broadcast(n_nodes, src=src, group=group)
        nn = int(n_nodes.item())
        buf = torch.empty(nn * abi.BVH_NODE.itemsize + n_idx * 4, dtype=torch.uint8, device=device)
        if owner == rank:
            nodes, idx = mine[k]
            blob = np.concatenate([np.ascontiguousarray(nodes).view(np.uint8).reshape(-1),
                                   np.ascontiguousarray(idx, dtype=np.uint32).view(np.uint8).reshape(-1)])
            buf.copy_(torch.from_numpy(blob))
        dist.broadcast(buf, src=src, group=group)
        host = buf.cpu().numpy()
        out.append((host[: nn * abi.BVH_NODE.itemsize].view(abi.BVH_NODE).copy(), host[nn * abi.BVH_NODE.itemsize:].view(np.uint32).copy()))
    return out
