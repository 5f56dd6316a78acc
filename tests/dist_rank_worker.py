"""One rank of the multi-rank C-ABI exchange tests (tests/test_gpu_dist_ranks.py starts N of these as fresh processes
before anything touches the GPU).  The rank binds the RCCL named by $VD_RCCL_LIB (the tests' double,
tests/cpp/fake_rccl.cpp: N ranks share one GPU), builds ITS shard of the seeded scene, runs vd_dist_step_full_dev /
vd_dist_step_draws_dev / vd_dist_step_indices_dev through voidin_amd.dist.RcclVisibility and compares every rank's list
with the whole-scene vd_cull_compact_dev result the parent wrote (count + BLAKE2 digest, raw bytes when small).
Prints one JSON line.  argv[1] = JSON job description."""
import ctypes
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def scene_shard(job, lo, hi):
    """Instances [lo, hi) of the job's scene - the parent builds [0, n) with the same function."""
    import numpy as np
    from voidin_amd import synth
    inst = synth.instances(hi - lo, n_mesh=job["n_mesh"], seed=job["seed"], offset=lo, with_inverse=False, **job["kw"])
    if hi > lo:
        gi = np.arange(lo, hi)
        inst["mesh"][gi % 97 == 0] = 0xFFFFFFF0                 # ids beyond the table: the unsigned clamp is part of the contract
        z0, z1 = job.get("hidden", (0, 0))                      # a whole shard behind the camera: a rank with zero survivors
        sel = (gi >= z0) & (gi < z1)
        if sel.any():
            t = np.zeros(16, dtype=np.float32)
            t[0] = t[5] = t[10] = 0.01
            t[12:16] = (2.0, 5.0, 1000.0, 1.0)
            inst["transform"][sel] = t
    return inst


def digest(b):
    return hashlib.blake2b(b, digest_size=16).hexdigest()


def main():
    job = json.loads(sys.argv[1])
    rank, world, n = job["rank"], job["world"], job["n"]
    res = {"rank": rank, "ok": False}
    try:
        import numpy as np
        import torch
        from voidin_amd import abi, synth
        from voidin_amd import dist as vdist
        from voidin_amd.runtime import Context, VoidinError
        torch.cuda.set_device(0)
        ctx = Context(0)
        # the communicator id: rank 0 makes it (vd_dist_unique_id), the others read it from the job directory
        id_path = os.path.join(job["dir"], "id.bin")
        if rank == 0:
            buf = (ctypes.c_ubyte * abi.VD_DIST_ID_BYTES)()
            ctx._chk(ctx.lib.vd_dist_unique_id(ctypes.addressof(buf)))
            with open(id_path + ".tmp", "wb") as f:
                f.write(bytes(buf))
            os.rename(id_path + ".tmp", id_path)
        t0 = time.time()
        while not os.path.exists(id_path):
            if time.time() - t0 > 120:
                raise RuntimeError("no communicator id from rank 0")
            time.sleep(0.01)
        uid = open(id_path, "rb").read()
        lo, hi = vdist.shard_range(n, rank, world)
        inst = scene_shard(job, lo, hi)
        cam, meshes = synth.camera_uniform(), synth.mesh_infos(job["n_mesh"])
        d_m = ctx.upload(meshes)
        d_i = ctx.upload(inst) if hi > lo else ctx.empty(144)
        rv = vdist.RcclVisibility(ctx, n, d_m, len(meshes), d_i, unique_id=uid, rank=rank, world=world)
        res["rccl"] = [int(rv.info.rccl_version), rv.info.rccl_library.decode()]
        res["shard"] = [lo, hi, int(rv.info.shard_size), int(rv.info.n_local)]
        d_out = ctx.empty(max(n, 1) * 20)
        d_cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
        want_n, want_digest = job["want_n"], job["want_digest"]
        want_raw = open(os.path.join(job["dir"], "want.bin"), "rb").read() if job.get("want_raw") else None
        bad = []
        if job.get("fail_send"):
            # the error path: this rank's k-th ncclSend fails inside the group of step_draws.  The call must report
            # VD_ERR_COMM AND leave no group open: the next collective on this thread has to go through.
            try:
                rv.step_draws(cam, d_out, d_cnt)
                failed = False
            except VoidinError as e:
                failed = e.code == abi.VD_ERR_COMM
                res["error_text"] = str(e)[:200]
            res["step_failed_with_comm_error"] = failed
            ctx.synchronize()
            probe = torch.full((8,), rank + 1, dtype=torch.uint8, device="cuda")
            got = torch.zeros(8 * world, dtype=torch.uint8, device="cuda")
            rv.allgather(probe, got, 8)
            ctx.synchronize()
            res["allgather_after_failure_ok"] = got.cpu().tolist() == [q + 1 for q in range(world) for _ in range(8)]
            modes = ()
        else:
            modes = (("full", rv.step), ("draws", rv.step_draws), ("indices", rv.step_indices))
        for name, fn in modes:
            for rep in range(2):
                d_out.fill_(0xEE)
                d_cnt.zero_()
                fn(cam, d_out, d_cnt)
                ctx.synchronize()
                cnt = int(d_cnt[0].item())
                blob = d_out[: cnt * 20].cpu().numpy().tobytes()
                if cnt != want_n or digest(blob) != want_digest or (want_raw is not None and blob != want_raw):
                    first = None
                    if want_raw is not None:
                        a = np.frombuffer(blob[: min(len(blob), len(want_raw)) // 20 * 20], dtype=np.uint8).reshape(-1, 20)
                        b = np.frombuffer(want_raw[: len(a) * 20], dtype=np.uint8).reshape(-1, 20)
                        d = np.nonzero((a != b).any(axis=1))[0]
                        first = int(d[0]) if d.size else None
                    bad.append([name, rep, cnt, first])
        # what the double saw: the world > 1 branches ran through ncclAllGather and the grouped ncclSend / ncclRecv
        try:
            fake = ctypes.CDLL(os.environ["VD_RCCL_LIB"])
            st = (ctypes.c_uint64 * 7)()
            fake.vd_fake_rccl_stats(st)
            res["fake_stats"] = dict(zip(("allgathers", "groups", "sends", "recvs", "bytes_sent", "bytes_received", "failed_sends"), [int(x) for x in st]))
        except (KeyError, OSError, AttributeError):
            res["fake_stats"] = None
        res["bad"] = bad
        res["ok"] = not bad
        rv.close()
        ctx.close()
    except Exception as e:  # surfaced by the parent
        import traceback
        res["error"] = repr(e) + "\n" + traceback.format_exc()[-1500:]
    print("RANK_RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
