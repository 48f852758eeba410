#!/usr/bin/env python3
"""Which engine does the runtime use for a device -> pinned-host copy?  Run under rocprofv3 --kernel-trace --memory-copy-trace:
SDMA copies show up in the memory-copy trace, shader blits as __amd_rocclr_copyBuffer kernels.  Variants:
  A  copy on an idle stream, nothing pending
  B  copy on the stream that just ran a kernel (stream order)
  C  copy on a second stream behind hipStreamWaitEvent(event of the kernel stream)
  D  host waits for the kernel's event, then copy on a second (idle) stream
Each variant prints its wall time; markers (tiny H2D copies of distinct sizes) separate them in the trace."""
import ctypes as C
import time

import torch

hip = C.CDLL("libamdhip64.so")
n = 16 << 20
x = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
src = torch.zeros(n, dtype=torch.uint8, device="cuda")
h = C.c_void_p()
assert hip.hipHostMalloc(C.byref(h), n, 0) == 0
s1 = torch.cuda.Stream()
s2 = torch.cuda.Stream()
marker = torch.zeros(4096, dtype=torch.uint8).pin_memory()
dm = torch.zeros(4096, dtype=torch.uint8, device="cuda")


def mark(k):
    torch.cuda.synchronize()
    dm[: 64 * (k + 1)].copy_(marker[: 64 * (k + 1)])
    torch.cuda.synchronize()


def kernel(stream):
    with torch.cuda.stream(stream):
        for _ in range(8):
            x.add_(1)  # ~64 MB read+write each


def d2h(stream):
    assert hip.hipMemcpyAsync(h, C.c_void_p(src.data_ptr()), n, 2, C.c_void_p(stream.cuda_stream)) == 0


for rep in range(2):
    mark(0)
    t = time.perf_counter(); d2h(s2); torch.cuda.synchronize(); print("A idle stream            %.3f ms" % ((time.perf_counter() - t) * 1e3))
    mark(1)
    t = time.perf_counter(); kernel(s1); d2h(s1); torch.cuda.synchronize(); print("B same stream as kernel  %.3f ms" % ((time.perf_counter() - t) * 1e3))
    mark(2)
    t = time.perf_counter(); kernel(s1); ev = torch.cuda.Event(); ev.record(s1); s2.wait_event(ev); d2h(s2); kernel(s1); torch.cuda.synchronize()
    print("C second stream + event  %.3f ms" % ((time.perf_counter() - t) * 1e3))
    mark(3)
    t = time.perf_counter(); kernel(s1); ev = torch.cuda.Event(); ev.record(s1); kernel(s1); ev.synchronize(); d2h(s2); torch.cuda.synchronize()
    print("D host wait, idle stream %.3f ms" % ((time.perf_counter() - t) * 1e3))
    mark(4)
