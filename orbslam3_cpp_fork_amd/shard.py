"""Multi-GPU sharding of the extraction path.

ORBextractor::operator() depends only on its own image (reference src/ORBextractor.cc:1052-1138), so the
path shards by camera/frame with no data-path collective; the one exchange step is a gather of the fixed-
capacity per-frame results to rank 0 (the consumer: Tracking runs in one process in the reference).  One
process per GPU, torch.distributed ("nccl" == RCCL on ROCm; "gloo" in the CPU tests).

Record layout gathered per rank, all uint8 views of the extractor's own output buffers:
    counts  [B][2] int32   (n, monoIndex)
    kps     [B][cap] 28-byte cv::KeyPoint records
    desc    [B][cap][32] uint8
~60 KB per frame at 1000 features: the gather is latency-bound, not xGMI-bandwidth-bound, so it is issued
as ONE flat byte tensor per rank per step (a single collective instead of three).
"""
import numpy as np


def camera_of_rank(rank, world, cameras_per_rank=1, slot=0):
    """Camera index owned by `rank`: cameras are dealt round-robin, camera c -> rank c % world."""
    return slot * world + rank


def cameras_of_rank(rank, world, n_cameras):
    return [c for c in range(n_cameras) if c % world == rank]


def record_bytes(B, cap):
    return B * 2 * 4 + B * cap * 28 + B * cap * 32


class RecordGather:
    """Packs (counts, keypoints, descriptors) of one step into one byte tensor and gathers it on rank 0."""

    def __init__(self, B, cap, device, world, rank, dst=0, host_staged=False):
        import torch

        self.B, self.cap, self.world, self.rank, self.dst = B, cap, world, rank, dst
        self.nbytes = record_bytes(B, cap)
        self.host_staged = host_staged  # backends without device collectives (gloo rehearsal): gather through host memory
        self.send = torch.empty(self.nbytes, dtype=torch.uint8, device=device)
        rdev = "cpu" if host_staged else device
        self.recv = [torch.empty(self.nbytes, dtype=torch.uint8, device=rdev) for _ in range(world)] if rank == dst else None
        self._o1 = B * 2 * 4
        self._o2 = self._o1 + B * cap * 28

    def pack(self, d_kps, d_desc, d_cnt):
        import torch

        self.send[: self._o1].copy_(d_cnt.reshape(-1).view(torch.uint8))
        self.send[self._o1: self._o2].copy_(d_kps.reshape(-1).view(torch.uint8))
        self.send[self._o2:].copy_(d_desc.reshape(-1))
        return self.send

    def __call__(self, d_kps, d_desc, d_cnt, async_op=False):
        """Pack on the current stream, then gather.  async_op=True returns the collective's work handle (call .wait()
        on it before the send buffer is packed again); the result is in self.recv on rank `dst`."""
        import torch.distributed as dist

        self.pack(d_kps, d_desc, d_cnt)
        send = self.send.cpu() if self.host_staged else self.send
        work = dist.gather(send, self.recv, dst=self.dst, async_op=async_op)
        return work if async_op else self.recv

    def unpack(self, buf):
        """buf: one rank's byte tensor (on any device) -> (counts[B,2] int32, kps[B,cap] KP_DTYPE, desc[B,cap,32])."""
        from ._lib import KP_DTYPE

        a = buf.cpu().numpy()
        cnt = a[: self._o1].view(np.int32).reshape(self.B, 2)
        kps = a[self._o1: self._o2].view(KP_DTYPE).reshape(self.B, self.cap)
        desc = a[self._o2:].reshape(self.B, self.cap, 32)
        return cnt, kps, desc


class CommGather:
    """The same exchange step through the C ABI (orbx_comm_create / orbx_gather, csrc/orbx_comm.hip): RCCL called from the
    library itself, as a C++ host would use it -- no pack, the three record arrays go out as they lie in HBM.

    The 128-byte rendezvous token is created by `root` and handed to the other ranks through `exchange(token_bytes_or_None) ->
    token_bytes` (a callable the host supplies: a torch.distributed / MPI broadcast, a file ...); with world == 1 nothing is
    exchanged."""

    def __init__(self, B, cap, device_index, world, rank, root=0, exchange=None):
        import ctypes as C

        import torch

        from . import _lib

        self.B, self.cap, self.world, self.rank, self.root = B, cap, world, rank, root
        L = _lib.lib()
        tok = (C.c_uint8 * 128)()
        if rank == root:
            _lib.check(L.orbx_comm_unique_id(tok), "orbx_comm_unique_id")
        if world > 1:
            if exchange is None:
                raise ValueError("CommGather: world > 1 needs an exchange callable for the rendezvous token")
            data = exchange(bytes(tok) if rank == root else None)
            tok = (C.c_uint8 * 128).from_buffer_copy(data)
        self._h = C.c_void_p()
        _lib.check(L.orbx_comm_create(C.byref(self._h), device_index, rank, world, tok), "orbx_comm_create")
        dev = "cuda:%d" % device_index
        if rank == root:
            self.all_kps = torch.empty((world, B, cap, 7), dtype=torch.float32, device=dev)
            self.all_desc = torch.empty((world, B, cap, 32), dtype=torch.uint8, device=dev)
            self.all_cnt = torch.empty((world, B, 2), dtype=torch.int32, device=dev)
        else:
            self.all_kps = self.all_desc = self.all_cnt = None

    def __call__(self, d_kps, d_desc, d_cnt, stream=0):
        """Enqueue the gather on `stream` (a raw hipStream_t value); on the root the result lands in all_kps / all_desc / all_cnt."""
        import ctypes as C

        from . import _lib

        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(_lib.lib().orbx_gather(self._h, p(d_kps), p(d_desc), p(d_cnt), self.B, self.cap, p(self.all_kps), p(self.all_desc), p(self.all_cnt),
                                          self.root, C.c_void_p(stream)), "orbx_gather")
        return self.all_kps, self.all_desc, self.all_cnt

    def close(self):
        from . import _lib

        if self._h:
            _lib.lib().orbx_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        # Also during interpreter shutdown: the destroy entry points are safe at any point of process teardown (the library releases the
        # device side of live handles from its own exit handler and only frees host state afterwards), and a handle that is destroyed is
        # one the runtime does not find alive in its own teardown.
        try:
            self.close()
        except Exception:
            pass
