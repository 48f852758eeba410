"""N > 1 path on CPU: world_size-2 gloo run of the camera sharding + record gather (the GPU run uses RCCL)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import oracle_py as O
    from orbslam3_cpp_fork_amd import shard, synth
    from orbslam3_cpp_fork_amd._lib import KP_DTYPE

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, h, w, cap = 2, 160, 200, 640
    cam = shard.camera_of_rank(rank, world)
    frames = synth.make_stream(h, w, B, camera_idx=cam)
    # the per-rank results come from the oracle here (no GPU in this test); the packing/gather code is the product's
    ex = O.OracleExtractor(500, 1.2, 4, 20, 7)
    kps = np.zeros((B, cap), KP_DTYPE)
    desc = np.zeros((B, cap, 32), np.uint8)
    cnt = np.zeros((B, 2), np.int32)
    for f in range(B):
        rc, k, d = ex(frames[f], (0, 0))
        kps[f, : len(k)], desc[f, : len(k)], cnt[f] = k, d, (len(k), rc)
    g = shard.RecordGather(B, cap, torch.device("cpu"), world, rank)
    t_k = torch.from_numpy(kps.view(np.float32).reshape(B, cap, 7))
    t_d, t_c = torch.from_numpy(desc), torch.from_numpy(cnt)
    recv = g(t_k, t_d, t_c)
    ok = True
    if rank == 0:
        for r in range(world):
            c2, k2, d2 = g.unpack(recv[r])
            fr = synth.make_stream(h, w, B, camera_idx=shard.camera_of_rank(r, world))
            for f in range(B):
                rc, k, d = ex(fr[f], (0, 0))
                ok &= int(c2[f, 0]) == len(k) and int(c2[f, 1]) == rc
                ok &= bool((k2[f, : len(k)] == k).all()) and bool((d2[f, : len(k)] == d).all())
        ok &= len(recv) == world
    else:
        ok &= recv is None
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_gather_world2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_camera_assignment():
    from orbslam3_cpp_fork_amd import shard

    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            cams = shard.cameras_of_rank(r, world, 8)
            assert len(cams) == 8 // world
            seen += cams
        assert sorted(seen) == list(range(8))
        assert [shard.camera_of_rank(r, world) for r in range(world)] == list(range(world))
    assert shard.record_bytes(2, 10) == 2 * 8 + 2 * 10 * 60
