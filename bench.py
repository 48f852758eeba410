#!/usr/bin/env python3
"""bench.py -- frames/s of the ORB extract + match hot path on MI355X (BASELINE.json metric).

Workload at N = 1 (BASELINE.json configs[1] + the matching half of the metric): a synthetic 640x480 8-bit
camera stream, ORBextractor(1000 features, 1.2, 8 levels, iniThFAST 20, minThFAST 7), every frame matched
against its predecessor with the brute-force SearchByBoW loop (TH_LOW 50, ratio 0.7, rotation check).
A "step" = one chunk of --frames frames already resident in HBM: 6 extractor kernels classes + 2 matcher
kernels over the whole chunk.  With N > 1 every rank runs the same chunk size on its own camera (weak
scaling, the path shards by frame) and the keypoints/descriptors are gathered to rank 0 over RCCL.

Prints ONE JSON line (rank 0).  `value` = frames of all ranks / max-over-ranks wall time of the timed steps.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (rows, cols, nfeatures)
    "vga_stream_640x480_8lvl_1000feat_extract+match": (480, 640, 1000),
    "euroc_stream_752x480_8lvl_1000feat_extract+match": (480, 752, 1000),
    "rig_1920x1080_8lvl_1000feat_extract+match": (1080, 1920, 1000),
    # BASELINE config 4: stereo pairs, two extractor handles per step, lapping {0, 0}, then Frame::ComputeStereoMatches
    "kitti_stereo_1241x376_8lvl_2000feat_extract+stereo": (376, 1241, 2000),
}


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


def cpu_baseline(frames, nfeatures, budget_s=12.0):
    """Single-thread CPU restatement (oracle/, kind "port") on a bounded sample of the same stream."""
    from oracle import oracle_py as O

    ex = O.OracleExtractor(nfeatures, 1.2, 8, 20, 7)
    prev = None
    rc, k, d = ex(frames[-1], (0, 1000))  # warm-up + predecessor of frame 0 (the GPU step wraps around the same way)
    prev = (k, d)
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < budget_s:
        f = done % len(frames)
        rc, k, d = ex(frames[f], (0, 1000))
        O.match_groups(prev[1], d, prev[0]["angle"], k["angle"], th=50, ratio=0.7, strict_lt=False, check_ori=True)
        prev = (k, d)
        done += 1
    dt = time.perf_counter() - t0
    return {
        "value": round(done / dt, 3),
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d frames of the same synthetic stream, extract + match vs previous frame, oracle/orb_oracle.cpp single thread (-O3 -march=x86-64-v3 -ffp-contract=off), %.1f s" % (done, dt),
    }


def run_stereo(args, rows, cols, nfeat):
    """Config 4 (single GPU): step = B stereo pairs: extract left, extract right (two handles), ComputeStereoMatches for every pair."""
    import torch

    from orbslam3_cpp_fork_amd import _lib, synth
    from orbslam3_cpp_fork_amd.extractor import ORBextractor

    B = min(args.frames, 128)
    dev = torch.device("cuda", 0)
    pairs = [synth.make_stereo_pair(rows, cols, camera_idx=i % 8, frame_idx=i // 8) for i in range(B)]
    L, R = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    st = ts.cuda_stream
    dL, dR = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
    gl, gr = ORBextractor(nfeat), ORBextractor(nfeat)
    cap = gl.max_keypoints(rows, cols)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    kl, kr, dl, dr = z((B, cap, 7), torch.float32), z((B, cap, 7), torch.float32), z((B, cap, 32), torch.uint8), z((B, cap, 32), torch.uint8)
    cl, cr, ur, dp, kept = z((B, 2), torch.int32), z((B, 2), torch.int32), z((B, cap), torch.float32), z((B, cap), torch.float32), z(B, torch.int32)
    mb, mbf = 0.5372, 386.1448  # KITTI 00-02
    ev = []

    def step(timed=False):
        gl.extract_batch_device(dL, B, rows, cols, cols, rows * cols, kl, dl, cl, cap, (0, 0), stream=st)
        gr.extract_batch_device(dR, B, rows, cols, cols, rows * cols, kr, dr, cr, cap, (0, 0), stream=st)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = _lib.lib().orbx_stereo_matches_batch_device(gl._h, gr._h, _lib.ptr(kl), _lib.ptr(dl), _lib.ptr(cl), _lib.ptr(kr), _lib.ptr(dr), _lib.ptr(cr), B, cap, mb,
                                                         mbf, _lib.ptr(ur), _lib.ptr(dp), _lib.ptr(kept), st)
        assert rc == 0
        if timed:
            e1.record()
            ev.append((e0, e1))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    gl.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stage = gl.stage_ms()
    stage["stereo_matches"] = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    value = B * args.steps / dt
    ab = algorithmic_bytes(rows, cols, nfeat)
    out = {
        "metric": "stereo pairs/s ORB extract x2 + ComputeStereoMatches (1241x376, 8 lvl, 2000 feat)", "value": round(value, 2), "unit": "pairs/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": args.workload, "pairs_per_step": B, "rows": rows, "cols": cols, "nfeatures": nfeat, "nlevels": 8,
                   "keypoints_per_eye": float(cl[:, 0].float().mean()), "stereo_matches_per_pair": float(kept.float().mean())},
        "roofline": {"bound": "hbm", "kernel": "k_fast_cells", "achieved": round(ab["fast"] * B / (stage["fast"] * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(ab["fast"] * B / (stage["fast"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": None,
                     "algorithmic_bytes_per_launch": ab["fast"] * B, "avg_launch_ms": round(stage["fast"], 5)},
        "stage_ms_left_eye": {k: round(v, 5) for k, v in stage.items()},
    }
    if not args.no_cpu_baseline:
        from oracle import oracle_py as O

        ol, orr = O.OracleExtractor(nfeat), O.OracleExtractor(nfeat)
        t0, done = time.perf_counter(), 0
        while time.perf_counter() - t0 < 12.0:
            f = done % B
            _, a, da = ol(L[f], (0, 0))
            _, b, db = orr(R[f], (0, 0))
            O.stereo_matches(ol, orr, a, da, b, db, mb, mbf)
            done += 1
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(done / dtc, 3), "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": "%d stereo pairs of the same set, two extractions + stereo matches, oracle single thread, %.1f s" % (done, dtc)}
        out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=256, help="frames per step and per GPU (resident in HBM)")
    ap.add_argument("--workload", default="vga_stream_640x480_8lvl_1000feat_extract+match", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-match", action="store_true", help="extract only")
    ap.add_argument("--lanes", type=int, default=1, help="chunks in flight (each on its own HIP stream and handles)")
    ap.add_argument("--rows", action="store_true", help="measure the SURVEY 8(f) rows N1-N3 (device path vs the CPU restatement on the same inputs) "
                    "instead of the headline workload: one JSON line per row (tests/measure_next_rows.py)")
    ap.add_argument("--bow", action="store_true", help="SearchByBoW proper: add ORBVocabulary::transform (synthetic k=10, L=6 tree) per frame and match "
                    "inside the FeatureVector groups instead of brute force (not the BASELINE configuration; reported as its own workload)")
    args = ap.parse_args()

    if args.rows:
        from tests import measure_next_rows

        return measure_next_rows.main()
    if args.workload.startswith("kitti_stereo"):
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise SystemExit("the stereo workload is single-GPU (pairs shard like frames; not wired into the multi-rank path)")
        return run_stereo(args, *WORKLOADS[args.workload])

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    local_rank %= max(torch.cuda.device_count(), 1)  # rehearsals with more ranks than GPUs share a device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # "nccl" IS RCCL on ROCm.  ORBX_DIST_BACKEND=gloo lets the multi-rank control flow be rehearsed on a one-GPU box
    # (RCCL refuses two ranks on one device); the gather then stages through host memory.
    backend = os.environ.get("ORBX_DIST_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from orbslam3_cpp_fork_amd import shard, synth
    from orbslam3_cpp_fork_amd.extractor import ORBextractor
    from orbslam3_cpp_fork_amd.matcher import ORBmatcher

    rows, cols, nfeat = WORKLOADS[args.workload]
    B = args.frames
    # one camera per rank (frames shard by camera; consecutive frames of a camera stay on one GPU for matching)
    frames_np = synth.make_stream(rows, cols, B, camera_idx=shard.camera_of_rank(rank, world))
    d_imgs = torch.from_numpy(frames_np).to(dev)

    cap = None
    pairs = np.stack([(np.arange(B) - 1) % B, np.arange(B)], 1).astype(np.int32)  # frame f-1 (query) vs frame f (train)
    d_pairs = torch.from_numpy(pairs).to(dev)

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
            self.tstream = torch.cuda.Stream(device=dev)  # explicit, non-default: NULL would select the handle's private stream
            self.stream = self.tstream.cuda_stream
            assert self.stream
            self.gather = shard.RecordGather(B, self.cap, dev, world, rank, host_staged=(backend != "nccl")) if world > 1 else None
            self.pending = None
            self.match_ev = []
            self.bow_ev = []
            if voc is not None:
                z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
                self.bow = {"bow_words": z((B, self.cap), torch.int32), "bow_vals": z((B, self.cap), torch.float64), "n_words": z(B, torch.int32),
                            "fv_nodes": z((B, self.cap), torch.int32), "fv_off": z((B, self.cap + 1), torch.int32), "fv_feat": z((B, self.cap), torch.int32),
                            "n_nodes": z(B, torch.int32)}

        def step(self, timed=False):
            with torch.cuda.stream(self.tstream):
                self.ex.extract_batch_device(d_imgs, B, rows, cols, cols, rows * cols, self.d_kps, self.d_desc, self.d_cnt, self.cap, (0, 1000), stream=self.stream)
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
                if self.gather is not None:
                    # Asynchronous RCCL gather on the collective's own stream.  It reads the packed copy of the records, not the
                    # extractor's buffers, so the NEXT chunk's extraction and matching run beside it; only the next pack has to
                    # wait for this gather to have drained the send buffer.
                    if self.pending is not None:
                        self.pending.wait()
                    self.pending = self.gather(self.d_kps, self.d_desc, self.d_cnt, async_op=True)

        def drain(self):
            if self.pending is not None:
                with torch.cuda.stream(self.tstream):
                    self.pending.wait()
                self.pending = None

    lanes = [Lane() for _ in range(max(1, args.lanes))]
    cap = lanes[0].cap

    def fence():
        for ln in lanes:
            ln.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        lanes[i % len(lanes)].step()
    fence()
    for ln in lanes:
        ln.ex.set_profiling(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        lanes[i % len(lanes)].step(timed=True)
    fence()
    dt = time.perf_counter() - t0
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
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    d_cnt, d_nm = lanes[0].d_cnt, lanes[0].d_nm

    total_frames = B * args.steps * world
    value = total_frames / dt
    cnt = d_cnt.cpu().numpy()
    nm = d_nm.cpu().numpy()

    if rank == 0:
        ab = algorithmic_bytes(rows, cols, nfeat)
        # dominant kernel class = the stage with the largest mean HIP-event duration in the timed region
        kernels = {"pyramid": "k_resize_level", "fast": "k_fast_cells", "octree": "k_octree", "blur": "k_blur_levels+k_blur_edges", "orient_describe": "k_orient_describe",
                   "match": "k_bf_topk+k_group_resolve"}
        # matcher: descriptors of both frames read once + one match index per query ((nq + nt) * 32 + nq * 4)
        ab["match"] = int(2 * cnt[:, 0].mean() * 32 + cnt[:, 0].mean() * 4)
        dom = max(kernels, key=lambda k: stage[k])
        # algorithmic HBM bytes of that kernel for one launch (= B frames); kernels without a streaming term move
        # only the keypoint records (quadtree) or gather through L2 (orientation/BRIEF): price those by the output term.
        per_frame = ab.get(dom, ab["output"])
        achieved = per_frame * B / (stage[dom] * 1e-3) / 1e9
        # HBM traffic of that kernel from the committed PMC passes of the same command (FETCH_SIZE x gfx950 correction +
        # WRITE_SIZE, collected by tools/collect_profiles.sh; bench.py itself cannot read hardware counters)
        traffic = None
        try:
            summ = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
            if summ.get("bench", {}).get("config", {}).get("frames_per_step_per_gpu") == B and cols == summ["bench"]["config"]["cols"]:
                names = kernels[dom].replace("||", "+").split("+")
                traffic = int(sum(summ["kernels"][n]["hbm_traffic_MB_per_launch"] for n in names) * 1e6)
        except Exception:
            traffic = None
        # share of SIMD issue slots the dominant kernel spends on VALU instructions, from the committed SQ-counter pass of the same
        # command (tools/sq_counters.sh + tools/sq_summary.py): the integer issue rate, not HBM, is this pipeline's ceiling
        valu_frac = None
        try:
            sq = json.load(open(os.path.join(ROOT, "profiles", "r01_sq_summary.json")))
            valu_frac = sq[kernels[dom].split("+")[0]]["valu_issue_frac"]
        except Exception:
            valu_frac = None
        out = {
            "metric": "frames/s ORB extract+match (640x480, 8 lvl, 1000 feat)" if cols == 640 else "frames/s ORB extract+match",
            "value": round(value, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": args.workload + ("" if not args.no_match else " [extract only]") + (" [+ BoW transform, SearchByBoW over FeatureVector groups]" if voc is not None else ""),
                "frames_per_step_per_gpu": B,
                "chunks_in_flight": len(lanes),
                "rows": rows,
                "cols": cols,
                "nfeatures": nfeat,
                "nlevels": 8,
                "parallelism": "frames sharded by camera, %d rank(s)%s" % (world, ", RCCL gather of keypoints+descriptors to rank 0" if world > 1 else ""),
                "keypoints_per_frame": float(cnt[:, 0].mean()),
                "matches_per_frame": float(nm.mean()) if not args.no_match else None,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernels[dom],
                "achieved": round(achieved, 3),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": traffic,
                "algorithmic_bytes_per_launch": per_frame * B,
                "avg_launch_ms": round(stage[dom], 5),
                "valu_issue_frac": valu_frac,
            },
            "stage_ms": {k: round(v, 5) for k, v in stage.items()},
            "pipeline_hbm": {
                "algorithmic_bytes_per_frame": ab["total"],
                "achieved_GBps": round(ab["total"] * value / world / 1e9, 3),
                "frac_of_peak": round(ab["total"] * value / world / 1e9 / HBM_PEAK_GBS, 6),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames_np, nfeat)
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
