"""One rank of tests/test_gpu_multirank.py: its own process, its own GPU.  Extracts the frames of its cameras of the rig workload, takes part
in orbx_comm_create / orbx_gather (RCCL behind the C ABI, csrc/orbx_comm.hip), and leaves what it computed -- and, on the root, what it
received from every rank -- in <outdir>/rank<r>.npz for the parent to compare with the oracle.

    python tests/multirank_gather_worker.py <rank> <world> <outdir> <rows> <cols> <frames per camera>

The rendezvous token travels through <outdir>/token.bin (written by rank 0, polled by the others): no torch.distributed in this path."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, outdir, rows, cols, per_cam = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    import numpy as np
    import torch

    from orbslam3_cpp_fork_amd import shard, synth
    from orbslam3_cpp_fork_amd.extractor import ORBextractor

    ndev = torch.cuda.device_count()
    dev_index = rank % ndev
    if world > ndev:
        raise SystemExit("worker: %d ranks but %d devices (RCCL refuses two ranks on one device)" % (world, ndev))
    torch.cuda.set_device(dev_index)
    cams = shard.cameras_of_rank(rank, world, 8)
    frames = np.concatenate([synth.make_stream(rows, cols, per_cam, camera_idx=cam) for cam in cams])
    B = len(frames)
    ex = ORBextractor(1000, 1.2, 8, 20, 7, device=dev_index)
    cap = ex.max_keypoints(rows, cols)
    dev = "cuda:%d" % dev_index
    d = torch.from_numpy(frames).to(dev)
    kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    cnt = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def exchange(tok):
        path = os.path.join(outdir, "token.bin")
        if tok is not None:
            with open(path + ".tmp", "wb") as f:
                f.write(tok)
            os.replace(path + ".tmp", path)
            return tok
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 120:
                raise SystemExit("worker %d: no rendezvous token after 120 s" % rank)
            time.sleep(0.05)
        return open(path, "rb").read()

    g = shard.CommGather(B, cap, dev_index, world, rank, root=0, exchange=exchange)
    times = []
    for step in range(3):  # the same exchange three times: communicator reuse, and a latency figure without the first-call setup
        ex.extract_batch_device(d, B, rows, cols, cols, rows * cols, kps, desc, cnt, cap, (0, 1000), stream=st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ak, ad, ac = g(kps, desc, cnt, stream=st)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    out = {"kps": kps.cpu().numpy(), "desc": desc.cpu().numpy(), "cnt": cnt.cpu().numpy(), "cams": np.array(cams), "gather_ms": np.array(times) * 1e3}
    if rank == 0:
        out.update(all_kps=ak.cpu().numpy(), all_desc=ad.cpu().numpy(), all_cnt=ac.cpu().numpy())
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    g.close()
    ex.close()
    print("rank %d of %d done: %d frames, gather %.3f ms" % (rank, world, B, times[-1] * 1e3), flush=True)


if __name__ == "__main__":
    main()
