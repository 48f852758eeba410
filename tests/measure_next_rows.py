"""Measurement of the SURVEY 8(f) rows built so far (N1 stereo matches, N2 BoW transform, N3 SearchByProjection): the device
path against the CPU restatement (oracle, one thread) on the same inputs, on this box.  One JSON line per row.
N1 / N3 are host-pointer, synchronous entry points (H2D + kernels + D2H per call); N2 is the device-batch entry point.
Lives under tests/ because it uses the oracle (as checker and as the timed CPU baseline); run it through
`python bench.py --rows` (GPU box), which prints one JSON line per row.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # checker / CPU baseline only
from orbslam3_cpp_fork_amd import synth
from orbslam3_cpp_fork_amd.extractor import ComputeStereoMatches, ORBextractor
from orbslam3_cpp_fork_amd.matcher import ORBmatcher
from orbslam3_cpp_fork_amd.vocabulary import ORBVocabulary


def timeit(fn, reps):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps


def main():
    for fn in (row_n1, row_n2, row_n3, row_n3_map, row_a13_windows, row_a13_triangulation, row_n3_batch):
        print(json.dumps(fn()), flush=True)


def row_n1():
    h, w, nf = 376, 1241, 2000
    left, right = synth.make_stereo_pair(h, w, camera_idx=5)
    gl, gr = ORBextractor(nf), ORBextractor(nf)
    _, kl, dl = gl(left, None, (0, 0))
    _, kr, dr = gr(right, None, (0, 0))
    ol, orr = O.OracleExtractor(nf), O.OracleExtractor(nf)
    ol(left, (0, 0))
    orr(right, (0, 0))
    mb, mbf = 0.5372, 386.1448
    ref = O.stereo_matches(ol, orr, kl, dl, kr, dr, mb, mbf)
    got = ComputeStereoMatches(gl, gr, kl, dl, kr, dr, mb, mbf)
    assert ref[0] == got[0] and (ref[1].view(np.uint32) == got[1].view(np.uint32)).all()
    tg = timeit(lambda: ComputeStereoMatches(gl, gr, kl, dl, kr, dr, mb, mbf), 30)
    tc = timeit(lambda: O.stereo_matches(ol, orr, kl, dl, kr, dr, mb, mbf), 10)
    return {"row": "N1 Frame::ComputeStereoMatches", "workload": "1241x376 stereo pair, 2000 features per eye (%d / %d keypoints, %d matches)" % (len(kl), len(kr), ref[0]),
            "gpu_ms_per_call": tg * 1e3, "cpu_oracle_ms_per_call": tc * 1e3, "speedup": tc / tg, "includes": "H2D of keypoints+descriptors, kernels, D2H (pyramids already resident)"}


def row_n2():
    B, rows, cols = 256, 480, 640
    fr = synth.make_stream(rows, cols, B)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    d = torch.from_numpy(fr).to(dev)
    ex = ORBextractor(1000)
    cap = ex.max_keypoints(rows, cols)
    kps = torch.zeros((B, cap, 7), device=dev)
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    cnt = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    ex.extract_batch_device(d, B, rows, cols, cols, rows * cols, kps, desc, cnt, cap, (0, 1000), stream=st.cuda_stream)
    voc = synth.make_vocabulary(10, 6, seed=11)
    gv = ORBVocabulary.from_arrays(10, 6, *voc)
    ov = O.OracleVocabulary(10, 6, *voc)
    out = {"bow_words": torch.zeros((B, cap), dtype=torch.int32, device=dev), "bow_vals": torch.zeros((B, cap), dtype=torch.float64, device=dev),
           "n_words": torch.zeros(B, dtype=torch.int32, device=dev), "fv_nodes": torch.zeros((B, cap), dtype=torch.int32, device=dev),
           "fv_off": torch.zeros((B, cap + 1), dtype=torch.int32, device=dev), "fv_feat": torch.zeros((B, cap), dtype=torch.int32, device=dev),
           "n_nodes": torch.zeros(B, dtype=torch.int32, device=dev)}
    for _ in range(3):
        gv.transform_batch_device(desc, cnt, B, cap, 4, out, stream=st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10):
        gv.transform_batch_device(desc, cnt, B, cap, 4, out, stream=st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    tg = e0.elapsed_time(e1) / 10 / B  # ms per frame
    c = cnt.cpu().numpy()
    D = desc.cpu().numpy()
    # parity on a few frames, CPU time on 32 frames
    for f in (0, 17, B - 1):
        r = ov.transform(D[f, : c[f, 0]], 4)
        nw = int(out["n_words"][f])
        assert nw == len(r[0]) and out["bow_vals"][f, :nw].cpu().numpy().tobytes() == r[1].tobytes()
    for f in range(32, 48):  # warm the 36 MB tree into the CPU caches as far as it goes
        ov.transform(D[f, : c[f, 0]], 4)
    t = time.perf_counter()
    for f in range(32):
        ov.transform(D[f, : c[f, 0]], 4)
    tc = (time.perf_counter() - t) / 32 * 1e3
    return {"row": "N2 ORBVocabulary::transform", "workload": "256 VGA frames x %.0f descriptors, k=10 L=6 tree (1 111 110 nodes), levelsup 4" % c[:, 0].mean(),
            "gpu_ms_per_frame": tg, "cpu_oracle_ms_per_frame": tc, "speedup": tc / tg, "includes": "device-resident batch (orbv_transform_batch_device), HIP events"}


def row_n3():
    frames = synth.make_stream(480, 640, 2, camera_idx=4)
    oex = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    _, kA, dA = oex(frames[0], (0, 1000))
    _, kB, dB = oex(frames[1], (0, 1000))
    sf = oex.scale_tables()[0]
    rng = np.random.default_rng(17)
    _, mq, _ = O.match_groups(dA, dB, kA["angle"], kB["angle"], 50, 0.7, False, True)
    ok = mq >= 0
    shift = np.median(np.stack([kB["x"][mq[ok]] - kA["x"][ok], kB["y"][mq[ok]] - kA["y"][ok]], 1), axis=0)
    uv = (np.stack([kA["x"], kA["y"]], 1) + shift + rng.normal(0, 3.0, (len(kA), 2))).astype(np.float32)
    valid = ((uv[:, 0] >= 0) & (uv[:, 0] <= 640) & (uv[:, 1] >= 0) & (uv[:, 1] <= 480)).astype(np.uint8)
    bounds = (0.0, 640.0, 0.0, 480.0)
    m = ORBmatcher(0.9, True)
    ref = O.search_by_projection(dA, uv, None, kA["octave"], kA["angle"], valid, dB, kB, None, None, sf, 15.0, 0, 0.0, bounds, 100, True)
    got = m.SearchByProjection(dA, uv, kA["octave"], kA["angle"], kB, dB, sf, 15.0, bounds, last_valid=valid)
    assert ref[0] == got[0] and ref[1].tolist() == got[1].tolist()
    tg = timeit(lambda: m.SearchByProjection(dA, uv, kA["octave"], kA["angle"], kB, dB, sf, 15.0, bounds, last_valid=valid), 50)
    tc = timeit(lambda: O.search_by_projection(dA, uv, None, kA["octave"], kA["angle"], valid, dB, kB, None, None, sf, 15.0, 0, 0.0, bounds, 100, True), 20)
    return {"row": "N3 SearchByProjection(Frame, Frame)", "workload": "%d projected points vs %d features, th = 15, %d matches" % (len(kA), len(kB), ref[0]),
            "gpu_ms_per_call": tg * 1e3, "cpu_oracle_ms_per_call": tc * 1e3, "speedup": tc / tg, "includes": "host grid build, H2D, 2 kernels, D2H"}


def row_n3_map():
    frames = synth.make_stream(480, 640, 2, camera_idx=7)
    oex = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    _, kA, dA = oex(frames[0], (0, 1000))
    _, kB, dB = oex(frames[1], (0, 1000))
    sf = oex.scale_tables()[0]
    rng = np.random.default_rng(31)
    _, mq, _ = O.match_groups(dA, dB, kA["angle"], kB["angle"], 50, 0.7, False, True)
    ok = mq >= 0
    shift = np.median(np.stack([kB["x"][mq[ok]] - kA["x"][ok], kB["y"][mq[ok]] - kA["y"][ok]], 1), axis=0)
    uv = (np.stack([kA["x"], kA["y"]], 1) + shift + rng.normal(0, 2.0, (len(kA), 2))).astype(np.float32)
    level = kA["octave"].astype(np.int32)
    radius = np.where(rng.random(len(kA)) < 0.5, 2.5, 4.0).astype(np.float32) * np.float32(3.0)
    bounds = (0.0, 640.0, 0.0, 480.0)
    m = ORBmatcher(0.8, True)
    ref = O.search_by_projection_map(dA, uv, None, level, radius, None, dB, kB, None, None, sf, 0.8, bounds, 100)
    got = m.SearchByProjectionMapPoints(dA, uv, level, radius, kB, dB, sf, bounds)
    assert ref[0] == got[0] and ref[1].tolist() == got[1].tolist()
    tg = timeit(lambda: m.SearchByProjectionMapPoints(dA, uv, level, radius, kB, dB, sf, bounds), 50)
    tc = timeit(lambda: O.search_by_projection_map(dA, uv, None, level, radius, None, dB, kB, None, None, sf, 0.8, bounds, 100), 20)
    return {"row": "N3 SearchByProjection(Frame, MapPoints)", "workload": "%d MapPoints vs %d features, th = 3, %d matches" % (len(kA), len(kB), ref[0]),
            "gpu_ms_per_call": tg * 1e3, "cpu_oracle_ms_per_call": tc * 1e3, "speedup": tc / tg, "includes": "host grid build, H2D, 2 kernels, D2H"}


def row_a13_windows():
    """SearchByProjection(KeyFrame, Sim3) (closing search) and Fuse (gate, every point on its own) through orbm_search_in_windows."""
    frames = synth.make_stream(480, 640, 2, camera_idx=5)
    oex = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    _, kA, dA = oex(frames[0], (0, 1000))
    _, kB, dB = oex(frames[1], (0, 1000))
    sf = oex.scale_tables()[0]
    s2 = (np.float32(1.0) / (sf * sf)).astype(np.float32)
    rng = np.random.default_rng(43)
    _, mq, _ = O.match_groups(dA, dB, kA["angle"], kB["angle"], 50, 0.7, False, True)
    ok = mq >= 0
    shift = np.median(np.stack([kB["x"][mq[ok]] - kA["x"][ok], kB["y"][mq[ok]] - kA["y"][ok]], 1), axis=0)
    uv = (np.stack([kA["x"], kA["y"]], 1) + shift + rng.normal(0, 1.0, (len(kA), 2))).astype(np.float32)
    level = np.clip(kA["octave"] + rng.integers(0, 2, len(kA)), 0, 7).astype(np.int32)
    bounds = (0.0, 640.0, 0.0, 480.0)
    m = ORBmatcher(0.8, True)
    ref = O.search_in_windows(dA, uv, None, level, None, dB, kB, None, None, sf, None, 4.0, 1, bounds, 50)
    got = m.SearchByProjectionSim3(dA, uv, level, kB, dB, sf, 4.0, 1.0, bounds)
    assert ref[0] == got[0] and ref[1].tolist() == got[1].tolist()
    reff = O.search_in_windows(dA, uv, None, level, None, dB, kB, None, None, sf, s2, 3.0, 0, bounds, 50)
    gotf = m.Fuse(dA, uv, level, kB, dB, sf, 3.0, bounds, inv_level_sigma2=s2)
    assert reff[0] == gotf[0] and reff[1].tolist() == gotf[1].tolist()
    tg = timeit(lambda: m.SearchByProjectionSim3(dA, uv, level, kB, dB, sf, 4.0, 1.0, bounds), 50)
    tc = timeit(lambda: O.search_in_windows(dA, uv, None, level, None, dB, kB, None, None, sf, None, 4.0, 1, bounds, 50), 20)
    tgf = timeit(lambda: m.Fuse(dA, uv, level, kB, dB, sf, 3.0, bounds, inv_level_sigma2=s2), 50)
    tcf = timeit(lambda: O.search_in_windows(dA, uv, None, level, None, dB, kB, None, None, sf, s2, 3.0, 0, bounds, 50), 20)
    return {"row": "A13 SearchByProjection(KeyFrame, Sim3) / Fuse", "workload": "%d points vs %d features, th = 4 / 3, %d / %d matches" % (len(kA), len(kB), ref[0], reff[0]),
            "gpu_ms_per_call": tg * 1e3, "cpu_oracle_ms_per_call": tc * 1e3, "speedup": tc / tg, "fuse_gpu_ms_per_call": tgf * 1e3,
            "fuse_cpu_oracle_ms_per_call": tcf * 1e3, "includes": "host grid build, H2D, 2 kernels, D2H"}


def row_a13_triangulation():
    frames = synth.make_stream(480, 640, 2, camera_idx=2)
    oex = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    _, kA, dA = oex(frames[0], (0, 1000))
    _, kB, dB = oex(frames[1], (0, 1000))
    sf = oex.scale_tables()[0]
    sg = (sf * sf).astype(np.float32)
    q_off, q_idx, t_off, t_idx = [0], [], [0], []
    for node in range(64):  # stand-in for the level-4 nodes of a vocabulary: split on six descriptor bits
        f1, f2 = np.nonzero(dA[:, 0] >> 2 == node)[0], np.nonzero(dB[:, 0] >> 2 == node)[0]
        if len(f1) and len(f2):
            q_idx += f1.tolist()
            t_idx += f2.tolist()
            q_off.append(len(q_idx))
            t_off.append(len(t_idx))
    groups = tuple(np.asarray(a, np.int32) for a in (q_off, q_idx, t_off, t_idx))
    _, mq, _ = O.match_groups(dA, dB, kA["angle"], kB["angle"], 50, 0.7, False, True)
    ok = mq >= 0
    dx, dy = float(np.median(kB["x"][mq[ok]] - kA["x"][ok])), float(np.median(kB["y"][mq[ok]] - kA["y"][ok]))
    t = np.array([dx, dy, 0.0]) / max(1e-6, np.hypot(dx, dy))
    Ki = np.linalg.inv(np.array([[460.0, 0, 320.0], [0, 455.0, 240.0], [0, 0, 1]]))
    F = (Ki.T @ np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ Ki).astype(np.float32)
    ep = np.array([1e6, 240.0], np.float32)
    m = ORBmatcher(0.6, True)
    ref = O.search_for_triangulation(dA, kA, None, None, dB, kB, None, None, groups, F, ep, sf, sg)
    got = m.SearchForTriangulation(kA, dA, None, groups, kB, dB, None, None, F, ep, sf, sg)
    assert ref[0] == got[0] and got[1][:, 1].tolist() == ref[1][ref[1] >= 0].tolist()
    tg = timeit(lambda: m.SearchForTriangulation(kA, dA, None, groups, kB, dB, None, None, F, ep, sf, sg), 50)
    tc = timeit(lambda: O.search_for_triangulation(dA, kA, None, None, dB, kB, None, None, groups, F, ep, sf, sg), 20)
    return {"row": "A13 SearchForTriangulation", "workload": "%d x %d features in %d common nodes, %d matches" % (len(kA), len(kB), len(q_off) - 1, ref[0]),
            "gpu_ms_per_call": tg * 1e3, "cpu_oracle_ms_per_call": tc * 1e3, "speedup": tc / tg, "includes": "H2D, 2 kernels, D2H"}


def row_n3_batch():
    """Device-resident batch form: 255 consecutive pairs of a 256-frame extraction batch, grid built on the device."""
    import ctypes as C

    from orbslam3_cpp_fork_amd import _lib

    B, rows, cols = 256, 480, 640
    fr = synth.make_stream(rows, cols, B)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    d = torch.from_numpy(fr).to(dev)
    ex = ORBextractor(1000)
    cap = ex.max_keypoints(rows, cols)
    kps = torch.zeros((B, cap, 7), device=dev)
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    cnt = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    ex.extract_batch_device(d, B, rows, cols, cols, rows * cols, kps, desc, cnt, cap, (0, 1000), stream=st.cuda_stream)
    torch.cuda.synchronize()
    P = B - 1
    pairs = torch.from_numpy(np.stack([np.arange(P), np.arange(1, B)], 1).astype(np.int32)).to(dev)
    # projection = last position + the stream's per-frame shift (estimated on the device from the keypoint means) + nothing else
    uv = kps[:P, :, :2].contiguous().clone()
    valid = torch.ones((P, cap), dtype=torch.uint8, device=dev)
    col = torch.zeros((P, cap), dtype=torch.int32, device=dev)
    loc = torch.zeros((P, cap), dtype=torch.int32, device=dev)
    nm = torch.zeros(P, dtype=torch.int32, device=dev)
    sf = np.ascontiguousarray(O.OracleExtractor(1000, 1.2, 8, 20, 7).scale_tables()[0], np.float32)
    m = ORBmatcher(0.9, True)
    L = _lib.lib()
    bounds = (0.0, float(cols), 0.0, float(rows))

    def launch():
        rc = L.orbm_search_by_projection_batch_device(m._h, P, _lib.ptr(pairs), _lib.ptr(desc), _lib.ptr(kps), _lib.ptr(cnt), B, cap, None, _lib.ptr(uv), None,
                                                      _lib.ptr(valid), None, None, _lib.ptr(sf), len(sf), 15.0, 0, 0.0, *bounds, 100, 1, _lib.ptr(col),
                                                      _lib.ptr(loc), _lib.ptr(nm), st.cuda_stream)
        assert rc == 0

    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10):
        launch()
    e1.record(st)
    torch.cuda.synchronize()
    tg = e0.elapsed_time(e1) / 10 / P
    c = cnt.cpu().numpy()
    # parity + CPU time on 8 pairs
    tcs = []
    for p in (0, 1, 100, 254):
        n1, n2 = int(c[p, 0]), int(c[p + 1, 0])
        a = np.frombuffer(kps[p, :n1].cpu().numpy().tobytes(), O.KP_DTYPE)
        b = np.frombuffer(kps[p + 1, :n2].cpu().numpy().tobytes(), O.KP_DTYPE)
        da, db = desc[p, :n1].cpu().numpy(), desc[p + 1, :n2].cpu().numpy()
        u = uv[p, :n1].cpu().numpy()
        t = time.perf_counter()
        ref = O.search_by_projection(da, u, None, a["octave"], a["angle"], None, db, b, None, None, sf, 15.0, 0, 0.0, bounds, 100, True)
        tcs.append(time.perf_counter() - t)
        assert int(nm[p]) == ref[0] and col[p, :n1].cpu().numpy().tolist() == ref[1].tolist()
    tc = float(np.mean(tcs)) * 1e3
    return {"row": "N3 SearchByProjection, device batch", "workload": "255 consecutive pairs of a 256-frame VGA batch (~960 x 960 each, th = 15), grid built on the device",
            "gpu_ms_per_pair": tg, "cpu_oracle_ms_per_pair": tc, "speedup": tc / tg, "includes": "k_assign_grid + k_proj_topk + k_proj_resolve, HIP events"}


if __name__ == "__main__":
    main()
