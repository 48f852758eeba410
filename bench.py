#!/usr/bin/env python3
"""bench.py -- frames/s of the ORB extract + match hot path on MI355X (BASELINE.json metric).

Workload at N = 1 (BASELINE.json configs[1] + the matching half of the metric): a synthetic 640x480 8-bit camera stream,
ORBextractor(1000 features, 1.2, 8 levels, iniThFAST 20, minThFAST 7), every frame matched against its predecessor with the
brute-force SearchByBoW loop (TH_LOW 50, ratio 0.7, rotation check).  A "step" = one chunk of --frames frames.

Two figures come out of one run:
  value           chunks already resident in HBM when the timed region starts (the contract's definition);
  pcie_inclusive  the same steps fed from pinned HOST memory through the streaming ABI (orbx_extract_async / orbx_wait): the timed
                  region contains the H2D copy of every frame and the D2H copy of every keypoint / descriptor / match record,
                  three chunks in flight (upload k + 1 || kernels k || download k - 1).  Never reported as `value`.

N > 1: `python bench.py --gpus N` starts N ranks itself (child processes, before anything touches the GPU) unless a launcher
(torch.distributed.run) already did; one rank per GPU over RCCL; fails non-zero when fewer than N devices or ranks exist.  Stream
workloads give every rank its own camera (weak scaling); the rig workload (BASELINE config 5) deals 8 cameras of 1920x1080 over the
ranks (strong scaling); the stereo workload gives every rank its own pairs.  The one exchange step is the gather of the fixed-capacity
records to rank 0.  Prints ONE JSON line (rank 0); `value` = units of all ranks / max-over-ranks wall time of the timed steps.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
PCIE_PEAK_GBS = 63.0   # PCIe Gen5 x16 spec (same guide)
PCIE_MEASURED_H2D_GBS = 57.0  # best single pinned copy measured on the GPU box (tools/dbg_pcie.py)
# wave64 VALU instruction = 2 cycles on a SIMD-32 (same guide, "Wave scheduling"): 256 CUs x 4 SIMDs x 2.4 GHz / 2
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2

WORKLOADS = {
    # name: (rows, cols, nfeatures)
    "vga_stream_640x480_8lvl_1000feat_extract+match": (480, 640, 1000),
    "euroc_stream_752x480_8lvl_1000feat_extract+match": (480, 752, 1000),
    # BASELINE config 5: 8 cameras, dealt over the ranks (8 / N cameras per GPU)
    "rig_8cam_1920x1080_8lvl_1000feat_extract+match": (1080, 1920, 1000),
    # BASELINE config 4: stereo pairs, two extractor handles per step, lapping {0, 0}, then Frame::ComputeStereoMatches
    "kitti_stereo_1241x376_8lvl_2000feat_extract+stereo": (376, 1241, 2000),
}
RIG_CAMERAS = 8


def level_sizes(rows, cols, nlevels=8, scale=1.2):
    """Per-level (w, h) with the constructor's float32 sequence (reference src/ORBextractor.cc:409-425, 1142-1143)."""
    sf = np.float32(1.0)
    out = []
    for _ in range(nlevels):
        inv = np.float32(1.0) / sf
        out.append((int(np.rint(np.float32(cols) * inv)), int(np.rint(np.float32(rows) * inv))))
        sf = np.float32(sf * np.float32(scale))
    return out


def algorithmic_bytes(rows, cols, nfeatures, nlevels=8):
    """HBM bytes one frame must move, split per kernel class."""
    P = [w * h for w, h in level_sizes(rows, cols, nlevels)]
    S = sum(P)
    return {
        "pyramid": sum(P[:-1]) + sum(P[1:]),  # read level l-1, write level l
        "fast": S,  # every level read once
        "blur": 2 * S,  # read + write
        "output": 48 * nfeatures,  # 16 B keypoint + 32 B descriptor
        "total": sum(P[:-1]) + sum(P[1:]) + 3 * S + 48 * nfeatures,
    }


# ------------------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle = this repository's single-thread restatement of the reference path
# ------------------------------------------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _pct(a, q):
    return float(np.percentile(np.asarray(a), q))


def cpu_baseline(frames, nfeatures, lap=(0, 1000), warm=20, timed=400, stereo=None):
    """One pinned thread, `warm` untimed frames then `timed` frames of the same synthetic stream; per-frame wall time around the
    operator() call alone (what REGISTER_TIMES brackets, src/Frame.cc:448-459) and around extract + match; then an all-cores run
    (one frame per thread at a time) for context.  stereo = (L, R, mb, mbf): a unit is a stereo pair (two extractions + stereo matches)."""
    from oracle import oracle_py as O

    native = O.select_native()  # -march=native build made on this very host (falls back to the portable x86-64-v3 build)
    cores = sorted(os.sched_getaffinity(0))
    old = set(cores)
    pin = cores[len(cores) // 2]
    os.sched_setaffinity(0, {pin})
    try:
        if stereo is None:
            ex = O.OracleExtractor(nfeatures, 1.2, 8, 20, 7)

            def unit(i, prev):
                f = i % len(frames)
                t0 = time.perf_counter()
                rc, k, d = ex(frames[f], lap)
                t1 = time.perf_counter()
                if prev is not None:
                    O.match_groups(prev[1], d, prev[0]["angle"], k["angle"], th=50, ratio=0.7, strict_lt=False, check_ori=True)
                t2 = time.perf_counter()
                return (k, d), t1 - t0, t2 - t0
        else:
            L, R, mb, mbf = stereo
            ol, orr = O.OracleExtractor(nfeatures), O.OracleExtractor(nfeatures)

            def unit(i, prev):
                f = i % len(L)
                t0 = time.perf_counter()
                _, a, da = ol(L[f], (0, 0))
                t1 = time.perf_counter()
                _, b, db = orr(R[f], (0, 0))
                O.stereo_matches(ol, orr, a, da, b, db, mb, mbf)
                t2 = time.perf_counter()
                return None, t1 - t0, t2 - t0

        prev = None
        for i in range(warm):
            prev, _, _ = unit(i, prev)
        te, tu = [], []
        for i in range(warm, warm + timed):
            prev, a, b = unit(i, prev)
            te.append(a * 1e3)
            tu.append(b * 1e3)
    finally:
        os.sched_setaffinity(0, old)
    # all-cores context run: one extractor per thread, frames in parallel (the oracle calls release the GIL)
    import threading

    nthr = len(cores)
    per = max(8, min(60, timed // 4))
    done = [0] * nthr

    def worker(t):
        if stereo is None:
            e = O.OracleExtractor(nfeatures, 1.2, 8, 20, 7)
            pv = None
            for i in range(per + 2):
                rc, k, d = e(frames[(t * 7 + i) % len(frames)], lap)
                if pv is not None:
                    O.match_groups(pv[1], d, pv[0]["angle"], k["angle"], th=50, ratio=0.7, strict_lt=False, check_ori=True)
                pv = (k, d)
                done[t] += 1
        else:
            L, R, mb, mbf = stereo
            a_, b_ = O.OracleExtractor(nfeatures), O.OracleExtractor(nfeatures)
            for i in range(per + 2):
                f = (t * 7 + i) % len(L)
                _, a, da = a_(L[f], (0, 0))
                _, b, db = b_(R[f], (0, 0))
                O.stereo_matches(a_, b_, a, da, b, db, mb, mbf)
                done[t] += 1

    thr = [threading.Thread(target=worker, args=(t,)) for t in range(nthr)]
    t0 = time.perf_counter()
    for t in thr:
        t.start()
    for t in thr:
        t.join()
    dt_all = time.perf_counter() - t0
    unit_name = "frames/s" if stereo is None else "pairs/s"
    what = "extract + match vs previous frame" if stereo is None else "two extractions + ComputeStereoMatches"
    return {
        "value": round(1e3 / _pct(tu, 50), 3),
        "unit": unit_name,
        "cores": 1,
        "kind": "port",
        "sample": ("%d units of the same synthetic input after %d warm-up, %s per unit, oracle/orb_oracle.cpp (this repository's scalar restatement of the "
                   "reference path; -O3 -march=%s -ffp-contract=off), ONE thread pinned to core %d; value = 1 / median. CAVEAT: the real reference links "
                   "OpenCV's SIMD-dispatched FAST / resize / GaussianBlur and may be faster than this scalar port; no OpenCV exists on this box to time it."
                   % (timed, warm, what, "native" if native else "x86-64-v3", pin)),
        "ms_per_unit": {"median": round(_pct(tu, 50), 4), "p10": round(_pct(tu, 10), 4), "p90": round(_pct(tu, 90), 4)},
        "ms_operator_call_only": {"median": round(_pct(te, 50), 4), "p10": round(_pct(te, 10), 4), "p90": round(_pct(te, 90), 4)},
        "all_cores": {"value": round(sum(done) / dt_all, 2), "unit": unit_name, "threads": nthr, "units": sum(done)},
        "cpu_model": _cpu_model(),
        "nproc": nthr,
    }


# ------------------------------------------------------------------------------------------------------------------------------
# process set-up
# ------------------------------------------------------------------------------------------------------------------------------
def self_launch(args, argv):
    """--gpus N without a launcher: start the N ranks as child processes.  Nothing in this process has touched the GPU
    (torch.cuda.device_count() does not initialise it), and the children are fresh interpreters."""
    n = args.gpus
    backend = os.environ.get("ORBX_DIST_BACKEND", "nccl")
    import torch

    ndev = torch.cuda.device_count()
    share = backend != "nccl" and os.environ.get("ORBX_SHARE_GPU") == "1"  # rehearsal of the control flow on a one-GPU box (gloo)
    if ndev < n and not share:
        sys.stderr.write("bench.py: --gpus %d but only %d device(s) visible\n" % (n, ndev))
        sys.exit(3)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in alive:  # a rank failed: the others would wait in a collective for ever
                    q.terminate()
        time.sleep(0.05)
    sys.exit(rc)


class Ctx:
    pass


def setup(args):
    import torch
    import torch.distributed as dist

    c = Ctx()
    c.rank = int(os.environ.get("RANK", "0"))
    c.world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != c.world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, c.world))
    # "nccl" IS RCCL on ROCm.  ORBX_DIST_BACKEND=gloo (+ ORBX_SHARE_GPU=1) lets the multi-rank control flow be rehearsed on a one-GPU
    # box (RCCL refuses two ranks on one device); the gather then stages through host memory.
    c.backend = os.environ.get("ORBX_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no GPU visible (there is no CPU fallback)")
    if c.world > ndev and not (c.backend != "nccl" and os.environ.get("ORBX_SHARE_GPU") == "1"):
        raise SystemExit("bench.py: %d ranks but only %d device(s) visible" % (c.world, ndev))
    c.local_rank = local_rank % ndev
    torch.cuda.set_device(c.local_rank)
    c.dev = torch.device("cuda", c.local_rank)
    if c.world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if c.backend == "nccl":
            dist.init_process_group("nccl", rank=c.rank, world_size=c.world, device_id=c.dev)
        else:
            dist.init_process_group(c.backend, rank=c.rank, world_size=c.world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: the process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    c.torch, c.dist = torch, dist
    return c


def max_over_ranks(c, dt):
    if c.world == 1:
        return dt
    t = c.torch.tensor([dt], dtype=c.torch.float64, device=c.dev if c.backend == "nccl" else "cpu")
    c.dist.all_reduce(t, op=c.dist.ReduceOp.MAX)
    return float(t.item())


def fence(c, lanes=()):
    for ln in lanes:
        ln.drain()
    if c.world > 1:
        c.dist.barrier()
    c.torch.cuda.synchronize()


def profile_summaries():
    """Committed profile summaries of THIS command (tools/collect_profiles.sh): bench.py cannot read hardware counters itself, so the
    HBM traffic and the VALU utilisation in the JSON line are those of the profiled run, tagged with their source file."""
    out = {}
    for key, names in (("pmc", ("r03_pmc_summary.json", "r02_pmc_summary.json")), ("valu", ("r03_valu_summary.json", "r02_valu_summary.json"))):
        out[key] = None
        for name in names:  # the newest round's summary that is committed
            try:
                out[key] = (name, json.load(open(os.path.join(ROOT, "profiles", name))))
                break
            except Exception:
                pass
    return out


# ------------------------------------------------------------------------------------------------------------------------------
def quiesce_host():
    """Before a timed region: collect now and keep the cycle collector out of the region (as timeit does).  A full collection with torch
    loaded takes ~40 ms -- the host pass, which has three chunks (~4.5 ms of device work) in flight, drains completely behind one."""
    gc.collect()
    gc.disable()


# stereo workload (config 4)
# ------------------------------------------------------------------------------------------------------------------------------
def run_stereo(args, c, rows, cols, nfeat):
    """step = B stereo pairs per rank: extract left, extract right (two handles), ComputeStereoMatches for every pair."""
    torch = c.torch
    from orbslam3_cpp_fork_amd import _lib, shard, synth
    from orbslam3_cpp_fork_amd.extractor import ORBextractor

    B = min(args.frames, 128)
    dev = c.dev
    pairs = [synth.make_stereo_pair(rows, cols, camera_idx=(i + c.rank) % 8, frame_idx=i // 8 + 100 * c.rank) for i in range(B)]
    L, R = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    dL, dR = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
    mb, mbf = 0.5372, 386.1448  # KITTI 00-02

    class SLane:
        """One chunk of stereo pairs in flight: two extractor handles (left / right eye), its buffers, a stream with its own hardware queue."""

        def __init__(self):
            self.gl, self.gr = ORBextractor(nfeat, device=c.local_rank), ORBextractor(nfeat, device=c.local_rank)
            self.cap = self.gl.max_keypoints(rows, cols)
            cap = self.cap
            z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
            self.kl, self.kr, self.dl, self.dr = z((B, cap, 7), torch.float32), z((B, cap, 7), torch.float32), z((B, cap, 32), torch.uint8), z((B, cap, 32), torch.uint8)
            self.cl, self.cr, self.ur, self.dp, self.kept = z((B, 2), torch.int32), z((B, 2), torch.int32), z((B, cap), torch.float32), z((B, cap), torch.float32), z(B, torch.int32)
            self.st = _lib.lib().orbx_stream_create(c.local_rank)
            assert self.st
            self.ts = torch.cuda.ExternalStream(self.st, device=dev)
            self.gather = shard.RecordGather(B, cap, dev, c.world, c.rank, host_staged=(c.backend != "nccl")) if c.world > 1 else None
            self.ev = []

        def step(self, timed=False):
            st, cap = self.st, self.cap
            with torch.cuda.stream(self.ts):
                self.gl.extract_batch_device(dL, B, rows, cols, cols, rows * cols, self.kl, self.dl, self.cl, cap, (0, 0), stream=st)
                self.gr.extract_batch_device(dR, B, rows, cols, cols, rows * cols, self.kr, self.dr, self.cr, cap, (0, 0), stream=st)
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                rc = _lib.lib().orbx_stereo_matches_batch_device(self.gl._h, self.gr._h, _lib.ptr(self.kl), _lib.ptr(self.dl), _lib.ptr(self.cl), _lib.ptr(self.kr),
                                                                 _lib.ptr(self.dr), _lib.ptr(self.cr), B, cap, mb, mbf, _lib.ptr(self.ur), _lib.ptr(self.dp),
                                                                 _lib.ptr(self.kept), st)
                assert rc == 0
                if timed:
                    e1.record()
                    self.ev.append((e0, e1))
                if self.gather is not None:
                    self.gather(self.kl, self.dl, self.cl)  # left-eye records to rank 0 (15 MB per rank per step)

        def close(self):
            torch.cuda.synchronize()
            self.gl.close(), self.gr.close()
            _lib.lib().orbx_stream_destroy(self.st)

    trace = (lambda m: print("bench.py[trace]: " + m, file=sys.stderr, flush=True)) if os.environ.get("ORBX_BENCH_TRACE") else (lambda m: None)
    lanes = [SLane() for _ in range(max(1, args.lanes))]
    nwarm = max(args.warmup, len(lanes))
    for i in range(nwarm):
        lanes[i % len(lanes)].step()
    fence(c)
    trace("warm-up done")
    quiesce_host()  # the timed region carries no instrumentation (no stage events, no timing events): see the VGA workload
    t0 = time.perf_counter()
    for i in range(args.steps):
        lanes[i % len(lanes)].step()
    trace("timed steps enqueued")
    fence(c)
    trace("timed steps done")
    dt = max_over_ranks(c, time.perf_counter() - t0)
    gc.enable()
    # the stages with one chunk in flight (nothing beside them)
    ln = lanes[0]
    for x in lanes:
        x.gl.set_profiling(False)
    ln.gl.set_profiling(True)
    ln.ev = []
    for k in range(3):
        ln.step(timed=True)
        fence(c)
        trace("solo step %d done" % k)
    gl, cl, kept, ev = ln.gl, ln.cl, ln.kept, ln.ev
    faults = [(i, nm, f) for i, x in enumerate(lanes) for nm, f in (("left", x.gl.debug_fault()), ("right", x.gr.debug_fault())) if f]
    if faults:  # a count a kernel consumes was out of range (orbx_debug_fault): the figures of this run are not to be trusted
        print("bench.py: inconsistent counts on record: %s" % faults, file=sys.stderr, flush=True)
    if c.rank != 0:
        for x in lanes:
            x.close()
        return
    stage = gl.stage_ms()
    stage["stereo_matches"] = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    value = B * args.steps * c.world / dt
    ab = algorithmic_bytes(rows, cols, nfeat)
    dom = max(("pyramid", "fast", "octree", "blur", "orient_describe"), key=lambda k: stage[k])
    per_frame = ab.get(dom, ab["output"])
    achieved = per_frame * B / (stage[dom] * 1e-3) / 1e9
    out = {
        "metric": "stereo pairs/s ORB extract x2 + ComputeStereoMatches (1241x376, 8 lvl, 2000 feat)", "value": round(value, 2), "unit": "pairs/s", "n_gpus": c.world,
        "steps": args.steps, "warmup": nwarm, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": args.workload, "pairs_per_step_per_gpu": B, "chunks_in_flight": len(lanes), "rows": rows, "cols": cols, "nfeatures": nfeat, "nlevels": 8,
                   "parallelism": "stereo pairs sharded over %d rank(s)%s" % (c.world, ", RCCL gather of the left-eye records to rank 0" if c.world > 1 else ""),
                   "keypoints_per_eye": float(cl[:, 0].float().mean()), "stereo_matches_per_pair": float(kept.float().mean())},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                     "algorithmic_bytes_per_launch": per_frame * B, "avg_launch_ms": round(stage[dom], 5)},
        "stage_ms_left_eye_one_chunk_in_flight": {k: round(v, 5) for k, v in stage.items()},
        "count_faults": faults,
    }
    if c.world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(None, nfeat, warm=4, timed=60, stereo=(L, R, mb, mbf))
        out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
    print(json.dumps(out), flush=True)
    for x in lanes:
        x.close()


# ------------------------------------------------------------------------------------------------------------------------------
# stream / rig workloads
# ------------------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (200 x 256 frames = 51 200 frames, ~0.2 s: 60 steps still read 1-3 %% below 200 / 600 on the same box, 20 gave +-3 %% run to run)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=256, help="frames per step and per GPU, resident in HBM (rig workload: per step over ALL GPUs, 8 cameras x frames/8)")
    ap.add_argument("--workload", default="vga_stream_640x480_8lvl_1000feat_extract+match", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive pass (host frames in, host records out)")
    ap.add_argument("--no-match", action="store_true", help="extract only")
    ap.add_argument("--no-single-call", action="store_true", help="skip the single synchronous orbx_extract latency figures (single_call_ms)")
    ap.add_argument("--c-gather", action="store_true", help="multi-rank runs: gather the records with orbx_gather (RCCL behind the C ABI) instead of torch.distributed.gather")
    ap.add_argument("--lanes", type=int, default=3, help="chunks in flight, each with its own handles, buffers and a HIP stream that owns a hardware queue: "
                    "the latency-bound kernels of one chunk (quadtree, claim resolution) run beside the VALU-bound ones of the others; 1 = strictly serial")
    ap.add_argument("--rows", action="store_true", help="measure the SURVEY 8(f) rows N1-N3 (device path vs the CPU restatement on the same inputs) "
                    "instead of the headline workload: one JSON line per row (tests/measure_next_rows.py)")
    ap.add_argument("--bow", action="store_true", help="SearchByBoW proper: add ORBVocabulary::transform (synthetic k=10, L=6 tree) per frame and match "
                    "inside the FeatureVector groups instead of brute force (not the BASELINE configuration; reported as its own workload)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    if args.rows:
        from tests import measure_next_rows

        return measure_next_rows.main()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, sys.argv[1:])

    c = setup(args)
    torch, dist = c.torch, c.dist
    rows, cols, nfeat = WORKLOADS[args.workload]
    if args.workload.startswith("kitti_stereo"):
        run_stereo(args, c, rows, cols, nfeat)
        if c.world > 1:
            dist.destroy_process_group()
        return

    from orbslam3_cpp_fork_amd import shard, synth
    from orbslam3_cpp_fork_amd.extractor import ORBextractor, pinned_array
    from orbslam3_cpp_fork_amd.matcher import ORBmatcher

    rank, world, dev, local_rank, backend = c.rank, c.world, c.dev, c.local_rank, c.backend
    rig = args.workload.startswith("rig_")
    if rig:
        # BASELINE config 5: 8 cameras dealt over the ranks; per step every camera delivers frames / 8 consecutive frames
        if RIG_CAMERAS % world:
            raise SystemExit("bench.py: the rig workload needs --gpus in {1, 2, 4, 8}")
        cams = shard.cameras_of_rank(rank, world, RIG_CAMERAS)
        per_cam = max(2, args.frames // RIG_CAMERAS)
        frames_np = np.concatenate([synth.make_stream(rows, cols, per_cam, camera_idx=cam) for cam in cams])
        B = len(frames_np)
        idx = np.arange(B)
        prev = np.where(idx % per_cam == 0, idx + per_cam - 1, idx - 1)  # predecessor inside the same camera (wraps around)
        pairs = np.stack([prev, idx], 1).astype(np.int32)
        total_frames_per_step = per_cam * RIG_CAMERAS
    else:
        # one camera per rank (frames shard by camera; consecutive frames of a camera stay on one GPU for matching)
        B = args.frames
        frames_np = synth.make_stream(rows, cols, B, camera_idx=shard.camera_of_rank(rank, world))
        pairs = np.stack([(np.arange(B) - 1) % B, np.arange(B)], 1).astype(np.int32)  # frame f-1 (query) vs frame f (train)
        total_frames_per_step = B * world
    d_imgs = torch.from_numpy(frames_np).to(dev)
    d_pairs = torch.from_numpy(pairs).to(dev)
    # Four DISTINCT resident chunks rotate through the lanes (the stream, and the stream mirrored vertically / horizontally / both: other
    # bytes, same statistics, consecutive frames still match): 4 x 78.6 MB of level-0 input at VGA is more than the 256 MiB Infinity
    # Cache, so no step finds the frames it reads left behind by the step before.
    d_chunks = [d_imgs, d_imgs.flip(1).contiguous(), d_imgs.flip(2).contiguous(), d_imgs.flip(1, 2).contiguous()]
    NCH = max(1, min(len(d_chunks), int(os.environ.get("ORBX_BENCH_CHUNKS", "4"))))  # ORBX_BENCH_CHUNKS=2: round 2's protocol (two chunks alternating), for like-for-like comparisons

    voc = None
    if args.bow and not args.no_match:
        from orbslam3_cpp_fork_amd.vocabulary import ORBVocabulary

        voc = ORBVocabulary.from_arrays(10, 6, *synth.make_vocabulary(10, 6, seed=11), device=local_rank)

    class Lane:
        """One in-flight chunk: its own extractor / matcher handles, output buffers and HIP stream.  Consecutive steps
        alternate between the lanes, so chunk i+1's throughput-bound kernels run beside chunk i's latency-bound ones
        (quadtree, descriptor gathers, claim resolution), exactly as a double-buffered camera stream would."""

        def __init__(self):
            self.ex = ORBextractor(nfeat, 1.2, 8, 20, 7, device=local_rank)
            self.mt = ORBmatcher(0.7, True, device=local_rank)
            self.cap = self.ex.max_keypoints(rows, cols)
            self.d_kps = torch.zeros((B, self.cap, 7), dtype=torch.float32, device=dev)
            self.d_desc = torch.zeros((B, self.cap, 32), dtype=torch.uint8, device=dev)
            self.d_cnt = torch.zeros((B, 2), dtype=torch.int32, device=dev)
            self.d_match = torch.full((B, self.cap), -1, dtype=torch.int32, device=dev)
            self.d_nm = torch.zeros(B, dtype=torch.int32, device=dev)
            # a stream with a hardware queue of its own (orbx_stream_create): torch's pooled streams share a few queues, and two chunks
            # on one queue would run strictly one after the other
            from orbslam3_cpp_fork_amd import _lib as _L

            self.stream = _L.lib().orbx_stream_create(local_rank)
            assert self.stream
            self.tstream = torch.cuda.ExternalStream(self.stream, device=dev)
            self.gather = shard.RecordGather(B, self.cap, dev, world, rank, host_staged=(backend != "nccl")) if world > 1 else None
            self.cgather = None
            if args.c_gather and world > 1 and backend == "nccl":
                # the same exchange through the C ABI (orbx_comm_create / orbx_gather): RCCL called by liborbx.so, no pack, enqueued on
                # the chunk's own stream.  The rendezvous token travels over the process group torch already has.
                def exchange(tok):
                    box = [tok]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]

                self.cgather, self.gather = shard.CommGather(B, self.cap, local_rank, world, rank, exchange=exchange), None
            self.pending = None
            self.match_ev = []
            self.bow_ev = []
            if voc is not None:
                z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
                self.bow = {"bow_words": z((B, self.cap), torch.int32), "bow_vals": z((B, self.cap), torch.float64), "n_words": z(B, torch.int32),
                            "fv_nodes": z((B, self.cap), torch.int32), "fv_off": z((B, self.cap + 1), torch.int32), "fv_feat": z((B, self.cap), torch.int32),
                            "n_nodes": z(B, torch.int32)}

        def step(self, timed=False, chunk=0):
            with torch.cuda.stream(self.tstream):
                self.ex.extract_batch_device(d_chunks[chunk % NCH], B, rows, cols, cols, rows * cols, self.d_kps, self.d_desc, self.d_cnt, self.cap, (0, 1000), stream=self.stream)
                if not args.no_match:
                    if voc is not None:
                        if timed:
                            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            b0.record()
                        voc.transform_batch_device(self.d_desc, self.d_cnt, B, self.cap, 4, self.bow, stream=self.stream)
                        if timed:
                            b1.record()
                            self.bow_ev.append((b0, b1))
                    if timed:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    if voc is not None:
                        from orbslam3_cpp_fork_amd import _lib as _L

                        rc = _L.lib().orbm_match_pairs_bow_device(self.mt._h, _L.ptr(self.d_desc), _L.ptr(self.d_kps), _L.ptr(self.d_cnt), self.cap, _L.ptr(d_pairs), B,
                                                                  _L.ptr(self.bow["fv_nodes"]), _L.ptr(self.bow["fv_off"]), _L.ptr(self.bow["fv_feat"]),
                                                                  _L.ptr(self.bow["n_nodes"]), 50, 0, 0.7, 1, _L.ptr(self.d_match), _L.ptr(self.d_nm), self.stream)
                        assert rc == 0
                    else:
                        self.mt.match_pairs_device(self.d_desc, self.d_kps, self.d_cnt, self.cap, d_pairs, B, self.d_match, self.d_nm, stream=self.stream)
                    if timed:
                        e1.record()
                        self.match_ev.append((e0, e1))
                if self.cgather is not None:
                    self.cgather(self.d_kps, self.d_desc, self.d_cnt, stream=self.stream)
                if self.gather is not None:
                    # Asynchronous RCCL gather on the collective's own stream.  It reads the packed copy of the records, not the
                    # extractor's buffers, so the NEXT chunk's extraction and matching run beside it; only the next pack has to
                    # wait for this gather to have drained the send buffer.
                    if self.pending is not None:
                        self.pending.wait()
                    self.pending = self.gather(self.d_kps, self.d_desc, self.d_cnt, async_op=True)

        def close(self):
            from orbslam3_cpp_fork_amd import _lib as _L

            if not self.stream:
                return
            torch.cuda.synchronize()
            self.ex.close()
            self.mt.close()
            _L.lib().orbx_stream_destroy(self.stream)
            self.stream = None

        def drain(self):
            if self.pending is not None:
                with torch.cuda.stream(self.tstream):
                    self.pending.wait()
                self.pending = None

    lanes = [Lane() for _ in range(max(1, args.lanes))]
    cap = lanes[0].cap

    nwarm = max(args.warmup, len(lanes))  # every lane allocates its workspace and uploads its tables in its first step: never inside the timed region
    for i in range(nwarm):
        lanes[i % len(lanes)].step(chunk=i)
    fence(c, lanes)
    # The timed region carries no instrumentation: the HIP events between the stages (7 per extraction, 2 per match) cost ~1 % at 20
    # steps, mostly host time in the first steps while the chunk lanes fill.  ORBX_BENCH_STAGE_EVENTS=1 puts them back (A/B).
    stage_events = os.environ.get("ORBX_BENCH_STAGE_EVENTS") is not None
    for ln in lanes:
        ln.ex.set_profiling(stage_events)
    quiesce_host()
    t0 = time.perf_counter()
    for i in range(args.steps):
        lanes[i % len(lanes)].step(timed=stage_events, chunk=nwarm + i)
    fence(c, lanes)
    dt = max_over_ranks(c, time.perf_counter() - t0)
    gc.enable()
    if not stage_events:  # the stage figures come from four more steps per lane, all lanes in flight, after the clock has stopped
        for ln in lanes:
            ln.ex.set_profiling(True)
        for i in range(4 * len(lanes)):
            lanes[i % len(lanes)].step(timed=True, chunk=i)
        fence(c, lanes)
    stages = []
    for ln in lanes:
        try:
            stages.append(ln.ex.stage_ms())
        except Exception:  # a lane that ran no timed step has nothing recorded
            pass
    stage = {k: float(np.mean([st[k] for st in stages])) for k in stages[0]}
    mev = [a.elapsed_time(b) for ln in lanes for a, b in ln.match_ev]
    stage["match"] = float(np.mean(mev)) if mev else 0.0
    bev = [a.elapsed_time(b) for ln in lanes for a, b in ln.bow_ev]
    if bev:
        stage["bow_transform"] = float(np.mean(bev))
    for ln in lanes:
        ln.ex.set_profiling(False)
    # the same stages with ONE chunk in flight (nothing running beside them): what a kernel costs on its own
    stage_serial = None
    if len(lanes) > 1:
        ln = lanes[0]
        ln.ex.set_profiling(True)
        ln.match_ev = []
        for i in range(4):
            ln.step(timed=True, chunk=i)
            fence(c, lanes)
        stage_serial = ln.ex.stage_ms()
        stage_serial["match"] = float(np.mean([a.elapsed_time(b) for a, b in ln.match_ev])) if ln.match_ev else 0.0
        ln.ex.set_profiling(False)
    cnt = lanes[0].d_cnt.cpu().numpy()
    nm = lanes[0].d_nm.cpu().numpy()
    faults = [(i, f) for i, x in enumerate(lanes) for f in (x.ex.debug_fault(),) if f]
    if faults:
        print("bench.py: inconsistent counts on record: %s" % faults, file=sys.stderr, flush=True)
    value = total_frames_per_step * args.steps / dt
    # the exchange step on its own (BASELINE.md config 5 asks for the gather latency): five gathers of lane 0's records, one at a time,
    # barrier before and a device synchronisation after each; maximum over the ranks of the median
    gather_ms = None
    if world > 1:
        ln, ts = lanes[0], []
        for _ in range(5):
            fence(c, lanes)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            with torch.cuda.stream(ln.tstream):
                if ln.cgather is not None:
                    ln.cgather(ln.d_kps, ln.d_desc, ln.d_cnt, stream=ln.stream)
                else:
                    ln.gather(ln.d_kps, ln.d_desc, ln.d_cnt, async_op=False)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        gather_ms = max_over_ranks(c, float(np.median(ts))) * 1e3

    # ---- PCIe-inclusive pass: the same steps fed from pinned host memory through the streaming ABI ------------------------------
    pcie = None
    if not args.no_pcie and voc is None:
        for ln in lanes:  # the resident pass is over: give its handles and queues back
            ln.close()
        from orbslam3_cpp_fork_amd._lib import KP_DTYPE

        # four distinct chunks rotate (the stream and its three mirror images: other bytes, same statistics), so no step re-reads a
        # chunk an earlier step left in the caches
        flips = (frames_np, frames_np[:, ::-1, :], frames_np[:, :, ::-1], frames_np[:, ::-1, ::-1])
        h_imgs = [pinned_array((B, rows, cols), np.uint8) for _ in range(max(1, min(4, int(os.environ.get("ORBX_BENCH_HOST_CHUNKS", "4")))))]
        for k, hbuf in enumerate(h_imgs):
            hbuf[:] = flips[k]
        NH, depth = int(os.environ.get("ORBX_BENCH_HOST_HANDLES", "1")), int(os.environ.get("ORBX_BENCH_HOST_DEPTH", "3"))  # one handle, three chunks in flight: the path is bound by the H2D copies (two or three handles side by side, ORBX_BENCH_HOST_HANDLES, measured 157-161 k against 167 k frames/s)

        class HostLane:
            def __init__(self):
                self.ex = ORBextractor(nfeat, 1.2, 8, 20, 7, device=local_rank)
                self.ex.set_async_depth(depth)
                self.mt = ORBmatcher(0.7, True, device=local_rank)
                self.h_kps = [pinned_array((B, cap), KP_DTYPE) for _ in range(depth)]
                self.h_desc = [pinned_array((B, cap, 32), np.uint8) for _ in range(depth)]
                self.h_cnt = [pinned_array((B, 2), np.int32) for _ in range(depth)]
                self.d_match = [torch.full((B, cap), -1, dtype=torch.int32, device=dev) for _ in range(depth)]
                self.d_nm = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(depth)]
                self.h_match = [pinned_array((B, cap), np.int32) for _ in range(depth)]
                self.h_nm = [pinned_array((B,), np.int32) for _ in range(depth)]
                self.inflight = []
                self.n = 0

            def submit(self, imgs):
                slot = self.n % depth
                self.n += 1
                if len(self.inflight) == depth:
                    self.ex.wait(self.inflight.pop(0))  # keypoints / descriptors / counts AND the chained match lists of that chunk are on the host
                t = self.ex.extract_async(imgs, self.h_kps[slot], self.h_desc[slot], self.h_cnt[slot], cap, (0, 1000))
                dk, dd, dc, st = self.ex.async_device_buffers(t)
                if not args.no_match:
                    self.mt.match_pairs_device(dd, dk, dc, cap, d_pairs, B, self.d_match[slot], self.d_nm[slot], stream=st)
                    self.ex.async_download(t, self.h_match[slot], self.d_match[slot])
                    self.ex.async_download(t, self.h_nm[slot], self.d_nm[slot])
                self.inflight.append(t)
                self.last = slot

            def drain(self):
                while self.inflight:
                    self.ex.wait(self.inflight.pop(0))

        hl = [HostLane() for _ in range(NH)]

        def host_step(i):
            hl[i % NH].submit(h_imgs[i % len(h_imgs)])

        def host_drain():
            for x in hl:
                x.drain()

        for i in range(max(args.warmup, NH * depth)):
            host_step(i)
        host_drain()
        fence(c)
        quiesce_host()
        t0 = time.perf_counter()
        stamps = []
        for i in range(args.steps):
            host_step(i)
            stamps.append(time.perf_counter() - t0)
        host_drain()
        fence(c)
        dth = max_over_ranks(c, time.perf_counter() - t0)
        gc.enable()
        if os.environ.get("ORBX_BENCH_TRACE"):
            print("bench.py[trace]: host pass, ms between submits: %s; total %.2f ms" % (" ".join("%.2f" % (1e3 * d) for d in np.diff([0.0] + stamps)), 1e3 * dth), file=sys.stderr, flush=True)
        in_b = B * rows * cols
        out_b = B * cap * 60 + B * 8 + (0 if args.no_match else B * cap * 4 + B * 4)
        pcie = {
            "value": round(total_frames_per_step * args.steps / dth, 2), "unit": "frames/s", "ms_per_step": round(dth / args.steps * 1e3, 4),
            "h2d_GBps_per_gpu": round(in_b * args.steps / dth / 1e9, 2), "d2h_GBps_per_gpu": round(out_b * args.steps / dth / 1e9, 2),
            "pcie_limit_frames_per_s_per_gpu": round(PCIE_PEAK_GBS * 1e9 / (rows * cols), 1),
            # what one large pinned host -> device copy reaches on this pool with nothing else running (tools/dbg_pcie.py: 57 GB/s)
            "h2d_measured_limit_frames_per_s_per_gpu": round(PCIE_MEASURED_H2D_GBS * 1e9 / (rows * cols), 1),
            "what": "same steps from pinned host memory via orbx_extract_async / orbx_wait (3 chunks in flight): timed region = H2D of every frame + "
                    "kernels + D2H of every fixed-capacity keypoint / descriptor / count record and of the match lists; four distinct chunks rotate",
            "keypoints_per_frame": float(hl[0].h_cnt[hl[0].last][:, 0].mean()),
            "matches_per_frame": None if args.no_match else float(hl[0].h_nm[hl[0].last].mean()),
        }
        for x in hl:
            x.ex.close()
            x.mt.close()
        del h_imgs, hl

    # ---- the call Frame::ExtractORB really makes (reference src/Frame.cc:555-562): ONE synchronous orbx_extract per frame, host image in,
    # host records out (single-frame latency).  Median wall time of 40 calls after 8, per size.
    single_call = None
    if rank == 0 and world == 1 and not args.no_single_call:
        single_call = {}
        for (h1, w1, nf1) in ((480, 640, 1000), (480, 752, 1000), (376, 1241, 2000)):
            img = synth.make_frame(h1, w1, 2, 3)
            ex1 = ORBextractor(nf1, 1.2, 8, 20, 7, device=local_rank)
            ts = []
            for i in range(48):
                t1 = time.perf_counter()
                ex1(img, None, (0, 1000))
                ts.append(time.perf_counter() - t1)
            single_call["%dx%d/%d" % (w1, h1, nf1)] = round(float(np.median(ts[8:])) * 1e3, 4)
            if (h1, w1) == (376, 1241):
                # what the C++ adaptor's default costs on the stereo path: levels >= 1 of the pyramid ride home inside the call
                # (orbx_set_pyramid_download; an unchanged Frame::ComputeStereoMatches reads mvImagePyramid, src/Frame.cc:1039-1058)
                ex1.set_pyramid_download(True)
                ts = []
                for i in range(48):
                    t1 = time.perf_counter()
                    ex1(img, None, (0, 0))
                    ts.append(time.perf_counter() - t1)
                single_call["%dx%d/%d+pyramid_to_host" % (w1, h1, nf1)] = round(float(np.median(ts[8:])) * 1e3, 4)
            ex1.close()

    if rank == 0:
        ab = algorithmic_bytes(rows, cols, nfeat)
        # dominant kernel class = the stage with the largest mean HIP-event duration in the timed region
        kernels = {"pyramid": "k_resize_level", "fast": "k_fast_tiles", "octree": "k_octree", "blur": "k_blur_all", "orient_describe": "k_orient_describe",
                   "match": "k_bf_topk+k_group_resolve"}
        # matcher: descriptors of both frames read once + one match index per query ((nq + nt) * 32 + nq * 4)
        ab["match"] = int(2 * cnt[:, 0].mean() * 32 + cnt[:, 0].mean() * 4)
        # dominant kernel = the stage that takes longest with the machine to itself (inside the timed region every stage's wall time also
        # holds the share of the other chunks' kernels that ran beside it, which says nothing about the stage)
        solo = stage_serial if stage_serial is not None else stage
        dom = max(kernels, key=lambda k: solo[k])
        # algorithmic HBM bytes of that kernel for one launch (= B frames); kernels without a streaming term move
        # only the keypoint records (quadtree) or gather through L2 (orientation/BRIEF): price those by the output term.
        per_frame = ab.get(dom, ab["output"])
        achieved = per_frame * B / (stage[dom] * 1e-3) / 1e9
        prof = profile_summaries()
        traffic, traffic_src, valu = None, None, None
        first = kernels[dom].split("+")[0]
        if prof["pmc"]:
            name, summ = prof["pmc"]
            try:
                if summ["bench"]["config"]["frames_per_step_per_gpu"] == B and cols == summ["bench"]["config"]["cols"]:
                    traffic = int(sum(summ["kernels"][n]["hbm_traffic_MB_per_launch"] for n in kernels[dom].split("+")) * 1e6)
                    traffic_src = "profiles/" + name
            except Exception:
                traffic = None
        if prof["valu"]:
            name, summ = prof["valu"]
            try:
                k = summ["kernels"][first]
                valu = {"bound": "valu", "kernel": first, "achieved": k["valu_ginst_per_s"], "peak": k["kernel_ceiling_ginst_per_s"], "unit": "G wave-instr/s",
                        "frac": k["valu_frac"], "valu_insts_per_launch": k["valu_insts_per_launch"],
                        "peak_is": "this kernel's own issue ceiling 1 / (f_full / R_full + f_half / R_half) from its static rate-class mix (tools/valu_mix.py) and the "
                                   "two measured issue rates (tools/valu_calibrate.py: %.0f / %.0f G wave-instr/s); frac = achieved / peak" % (
                                       summ["calibration"].get("full_rate_ginst_per_s", summ["calibration"].get("vop2_ginst_per_s", 0)),
                                       summ["calibration"].get("half_rate_ginst_per_s", summ["calibration"].get("vop3_ginst_per_s", 0))),
                        "source": "profiles/" + name}
            except Exception:
                valu = None
        # config 3's matcher figure: descriptor pairs compared per second by the brute-force pass (nq x nt per frame pair), priced with the
        # match stage's own duration (one chunk in flight when that was measured)
        n_kp = cnt[:, 0].astype(np.float64)
        pair_count = float((n_kp[pairs[:, 0]] * n_kp[pairs[:, 1]]).sum())
        match_ms = (stage_serial or stage).get("match", 0.0)
        out = {
            "metric": "frames/s ORB extract+match (640x480, 8 lvl, 1000 feat)" if cols == 640 else "frames/s ORB extract+match",
            "value": round(value, 2),
            "value_pcie_inclusive": None if pcie is None else pcie["value"],
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": nwarm,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if rig else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": args.workload + ("" if not args.no_match else " [extract only]") + (" [+ BoW transform, SearchByBoW over FeatureVector groups]" if voc is not None else ""),
                "value_definition": "value: the chunks are resident in HBM when the timed region starts (the bench contract).  SURVEY.md 8(d)'s GPU-timing row "
                                    "('include H2D of the image and D2H of results') is value_pcie_inclusive: the same steps from pinned host memory, every "
                                    "frame uploaded and every record downloaded inside the timed region",
                "frames_per_step_per_gpu": B,
                "distinct_resident_chunks": NCH,
                "resident_input_MB": round(NCH * B * rows * cols / 1e6, 1),
                "chunks_in_flight": len(lanes),
                "stage_ms_note": "stage_ms / roofline.timed_region: HIP events on each chunk's own stream inside the timed region, i.e. WITH the other chunks' kernels running beside them; "
                                 "stage_ms_one_chunk_in_flight / roofline: the same stages measured right after it with nothing beside them",
                "rows": rows,
                "cols": cols,
                "nfeatures": nfeat,
                "nlevels": 8,
                "parallelism": ("8 cameras dealt over %d rank(s) (%d per GPU)" % (world, RIG_CAMERAS // world) if rig else "frames sharded by camera, %d rank(s)" % world)
                               + ((", RCCL gather of keypoints+descriptors to rank 0" + (" (orbx_gather, C ABI)" if args.c_gather and backend == "nccl" else "")) if world > 1 else ""),
                "keypoints_per_frame": float(cnt[:, 0].mean()),
                "matches_per_frame": float(nm.mean()) if not args.no_match else None,
            },
            # Roofline of the dominant kernel.  With several chunks in flight a launch shares the machine with the other chunks' kernels for
            # its whole duration, so its wall time inside the timed region is not what the kernel costs; the headline figures are
            # therefore taken with ONE chunk in flight (HIP events on the kernel's own stream, measured by this run right after the
            # timed region: the conditions of round 1 and of profiles/r02_one_chunk_kernel_stats.csv), and the timed-region
            # figures sit beside them under "timed_region".
            "roofline": {
                "bound": "hbm",
                "binding": "valu",  # what actually limits the kernel: vector-instruction issue (see "valu"); the HBM figures are the contract's
                "kernel": kernels[dom],
                "achieved": round(per_frame * B / (solo[dom] * 1e-3) / 1e9, 3),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(per_frame * B / (solo[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": per_frame * B,
                "avg_launch_ms": round(solo[dom], 5),
                "measured": ("HIP events on the kernel's stream with one chunk in flight, right after the timed region" if stage_serial is not None
                             else "HIP events on the kernel's stream inside the timed region (one chunk in flight)"),
                "valu": valu,
                "timed_region": {"chunks_in_flight": len(lanes), "avg_launch_ms": round(stage[dom], 5), "achieved": round(achieved, 3),
                                 "frac": round(achieved / HBM_PEAK_GBS, 6)},
            },
            "stage_ms": {k: round(v, 5) for k, v in stage.items()},
            "stage_ms_one_chunk_in_flight": None if stage_serial is None else {k: round(v, 5) for k, v in stage_serial.items()},
            "pipeline_hbm": {
                "algorithmic_bytes_per_frame": ab["total"],
                "achieved_GBps": round(ab["total"] * value / world / 1e9, 3),
                "frac_of_peak": round(ab["total"] * value / world / 1e9 / HBM_PEAK_GBS, 6),
            },
            "pcie_inclusive": pcie,
            "match_gpairs_per_s": None if args.no_match or match_ms <= 0 else round(pair_count / (match_ms * 1e-3) / 1e9, 3),
            "single_call_ms": single_call,
            "gather_ms": None if gather_ms is None else round(gather_ms, 4),
            "count_faults": faults,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames_np, nfeat)
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    # orderly teardown while the runtime is alive (handles, then the streams they were driven on); nothing is left to finalizers
    for ln in lanes:
        if ln.stream:
            ln.close()
    if voc is not None:
        voc.close() if hasattr(voc, "close") else None
    del d_imgs, d_pairs
    torch.cuda.synchronize()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
