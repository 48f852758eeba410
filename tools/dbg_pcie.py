#!/usr/bin/env python3
"""PCIe diagnostics on the GPU box: which NUMA node the GPU hangs off, and the H2D / D2H rate of one large pinned copy as a function of
where the pinned pages live (the process is bound to each NUMA node in turn before it allocates)."""
import ctypes as C
import glob
import os
import sys
import time

hip = C.CDLL("libamdhip64.so")


def chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s -> %d" % (what, rc))


def numa_nodes():
    out = {}
    for p in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        n = int(p.rsplit("node", 1)[1])
        cpus = open(p + "/cpulist").read().strip()
        out[n] = cpus
    return out


def parse_cpulist(s):
    r = set()
    for part in s.split(","):
        if "-" in part:
            a, b = part.split("-")
            r.update(range(int(a), int(b) + 1))
        elif part:
            r.add(int(part))
    return r


def main():
    chk(hip.hipSetDevice(0), "hipSetDevice")
    bdf = C.create_string_buffer(64)
    chk(hip.hipDeviceGetPCIBusId(bdf, 64, 0), "hipDeviceGetPCIBusId")
    bdf = bdf.value.decode().lower()
    try:
        gnode = open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip()
    except OSError:
        gnode = "?"
    nodes = numa_nodes()
    print("GPU 0 at %s, numa_node %s; nodes: %s; affinity of this process: %d cpus" % (bdf, gnode, nodes, len(os.sched_getaffinity(0))))
    n = 80 << 20
    d = C.c_void_p()
    chk(hip.hipMalloc(C.byref(d), n), "hipMalloc")
    s = C.c_void_p()
    chk(hip.hipStreamCreate(C.byref(s)), "hipStreamCreate")
    allowed = os.sched_getaffinity(0)
    for node, cpus in list(nodes.items()) + [(-1, None)]:
        if cpus is not None:
            cs = parse_cpulist(cpus) & allowed
            if not cs:
                continue
            os.sched_setaffinity(0, cs)
        else:
            os.sched_setaffinity(0, allowed)
        h = C.c_void_p()
        chk(hip.hipHostMalloc(C.byref(h), n, 0), "hipHostMalloc")
        C.memset(h, 1, n)  # first touch from this node
        for kind, name in ((1, "H2D"), (2, "D2H")):
            best = 0
            for _ in range(5):
                t0 = time.perf_counter()
                if kind == 1:
                    chk(hip.hipMemcpyAsync(d, h, n, 1, s), "copy")
                else:
                    chk(hip.hipMemcpyAsync(h, d, n, 2, s), "copy")
                chk(hip.hipStreamSynchronize(s), "sync")
                best = max(best, n / (time.perf_counter() - t0) / 1e9)
            print("pinned pages allocated while bound to node %s: %s %.1f GB/s" % (node if cpus is not None else "all", name, best))
        hip.hipHostFree(h)
    os.sched_setaffinity(0, allowed)


if __name__ == "__main__":
    main()
