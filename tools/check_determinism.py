"""Determinism check (and race hunt that needs no fault): the stereo workload of bench.py (three lanes, two handles each, timing events and stage profiling as
the bench has them) runs the SAME batch every step, and every step's outputs are compared on the device with the first step's.  Any
difference is a race (the path is deterministic by contract); the report names the output, the step and the frames.

usage: check_determinism.py [steps] [sync_every] [rows cols nfeatures B]
MODE=mono: bench.py's default step instead -- one extractor per lane on the synthetic stream and the brute-force matcher on consecutive
frames (match lists and counts compared as well)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from orbslam3_cpp_fork_amd import synth, _lib
from orbslam3_cpp_fork_amd.extractor import ORBextractor
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 180
SYNC_EVERY = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows, cols, nf, B = (int(x) for x in sys.argv[3:7]) if len(sys.argv) > 6 else (376, 1241, 2000, 128)
NL = int(os.environ.get("LANES", "3"))
MONO = os.environ.get("MODE") == "mono"
STEREO = os.environ.get("NO_STEREO") != "1"
PROFILING = os.environ.get("NO_PROFILING") != "1"
if MONO:
    from orbslam3_cpp_fork_amd.matcher import ORBmatcher
    frames = synth.make_stream(rows, cols, B)
    dF = torch.from_numpy(frames).cuda()
    d_pairs = torch.from_numpy(np.stack([(np.arange(B) - 1) % B, np.arange(B)], 1).astype(np.int32)).cuda()
    mlanes = []
    for li in range(NL):
        ex, mt = ORBextractor(nf), ORBmatcher(0.7, True)
        cap = ex.max_keypoints(rows, cols)
        t = dict(k=torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"), d=torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"),
                 c=torch.zeros((B, 2), dtype=torch.int32, device="cuda"), m=torch.full((B, cap), -1, dtype=torch.int32, device="cuda"), nm=torch.zeros(B, dtype=torch.int32, device="cuda"))
        st = _lib.lib().orbx_stream_create(0)
        mlanes.append(dict(ex=ex, mt=mt, t=t, st=st, ts=torch.cuda.ExternalStream(st), cap=cap, ref=None, bad={k: torch.zeros(B, dtype=torch.int32, device="cuda") for k in t}))
    def mstep(ln):
        t, st, cap = ln["t"], ln["st"], ln["cap"]
        with torch.cuda.stream(ln["ts"]):
            ln["ex"].extract_batch_device(dF, B, rows, cols, cols, rows * cols, t["k"], t["d"], t["c"], cap, (0, 1000), stream=st)
            ln["mt"].match_pairs_device(t["d"], t["k"], t["c"], cap, d_pairs, B, t["m"], t["nm"], stream=st)
            if ln["ref"] is None:
                ln["ref"] = {k: v.clone() for k, v in t.items()}
                return
            ref = ln["ref"]
            valid = torch.arange(cap, device="cuda")[None, :] < ref["c"][:, 0:1]
            qvalid = valid[d_pairs[:, 0].long()]  # match lists are indexed by the query frame's features
            for k in t:
                a, b = t[k], ref[k]
                if k in ("k", "d"):
                    d = ((a.view(torch.int32) if a.dtype == torch.float32 else a) != (b.view(torch.int32) if b.dtype == torch.float32 else b)).any(dim=2)
                    d = (d & valid).sum(dim=1)
                elif k == "m":
                    d = ((a != b) & qvalid).sum(dim=1)
                elif k == "nm":
                    d = (a != b).int()
                else:
                    d = (a != b).any(dim=1).int()
                ln["bad"][k] += d.int()
    def mreport(i):
        torch.cuda.synchronize()
        msgs = []
        for li, ln in enumerate(mlanes):
            for k in ln["t"]:
                b = ln["bad"][k].cpu().numpy()
                if b.any():
                    msgs.append("lane%d.%s: %d differing records over frames %s" % (li, k, int(b.sum()), np.nonzero(b)[0][:12].tolist()))
            f = ln["ex"].debug_fault()
            if f:
                msgs.append("lane%d count fault %s" % (li, f))
        print("step %d: %s" % (i, "; ".join(msgs) if msgs else "all outputs identical to the first step's"), flush=True)
        return bool(msgs)
    for ln in mlanes:
        mstep(ln)
    torch.cuda.synchronize()
    bad = False
    for i in range(steps):
        mstep(mlanes[i % NL])
        if i % SYNC_EVERY == SYNC_EVERY - 1 and mreport(i + 1):
            bad = True
            break
    bad = mreport(steps) or bad
    print("done", flush=True)
    sys.exit(1 if bad else 0)
pairs = [synth.make_stereo_pair(rows, cols, camera_idx=i % 8, frame_idx=i // 8) for i in range(B)]
dL = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda(); dR = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
KEYS = ("cl", "cr", "kl", "kr", "dl", "dr", "kept", "ur", "dp")
lanes = []
for li in range(NL):
    gl, gr = ORBextractor(nf), ORBextractor(nf)
    cap = gl.max_keypoints(rows, cols)
    t = dict(kl=z((B, cap, 7), torch.float32), kr=z((B, cap, 7), torch.float32), dl=z((B, cap, 32), torch.uint8), dr=z((B, cap, 32), torch.uint8), cl=z((B, 2), torch.int32),
             cr=z((B, 2), torch.int32), ur=z((B, cap), torch.float32), dp=z((B, cap), torch.float32), kept=z(B, torch.int32))
    st = _lib.lib().orbx_stream_create(0)
    lanes.append(dict(gl=gl, gr=gr, t=t, st=st, ts=torch.cuda.ExternalStream(st), cap=cap, ref=None, bad={k: z(B, torch.int32) for k in KEYS}, first={k: None for k in KEYS}))
kept_events = []
def valid_mask(counts, cap):  # only the first count[f] records of a frame are defined
    return torch.arange(cap, device="cuda")[None, :] < counts[:, 0:1]
def step(ln, i):
    t, st, cap = ln["t"], ln["st"], ln["cap"]
    with torch.cuda.stream(ln["ts"]):
        ln["gl"].extract_batch_device(dL, B, rows, cols, cols, rows * cols, t["kl"], t["dl"], t["cl"], cap, (0, 0), stream=st)
        ln["gr"].extract_batch_device(dR, B, rows, cols, cols, rows * cols, t["kr"], t["dr"], t["cr"], cap, (0, 0), stream=st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if STEREO:
            rc = _lib.lib().orbx_stereo_matches_batch_device(ln["gl"]._h, ln["gr"]._h, _lib.ptr(t["kl"]), _lib.ptr(t["dl"]), _lib.ptr(t["cl"]), _lib.ptr(t["kr"]), _lib.ptr(t["dr"]), _lib.ptr(t["cr"]),
                                                             B, cap, 0.5372, 386.1448, _lib.ptr(t["ur"]), _lib.ptr(t["dp"]), _lib.ptr(t["kept"]), st)
            assert rc == 0
        e1.record(); kept_events.append((e0, e1))
        if ln["ref"] is None:
            ln["ref"] = {k: v.clone() for k, v in t.items()}
            return
        ref = ln["ref"]
        ml, mr = valid_mask(ref["cl"], cap), valid_mask(ref["cr"], cap)
        for k in KEYS:
            a, b = t[k], ref[k]
            if k in ("kl", "dl", "ur", "dp"):
                d = ((a.view(torch.int32) if a.dtype == torch.float32 else a) != (b.view(torch.int32) if b.dtype == torch.float32 else b))
                d = d if d.dim() == 2 else d.any(dim=2)
                d = (d & ml).sum(dim=1)
            elif k in ("kr", "dr"):
                d = ((a.view(torch.int32) if a.dtype == torch.float32 else a) != (b.view(torch.int32) if b.dtype == torch.float32 else b)).any(dim=2)
                d = (d & mr).sum(dim=1)
            elif k == "kept":
                d = (a != b).int()
            else:
                d = (a != b).any(dim=1).int()
            ln["bad"][k] += d.int()
def report(i):
    torch.cuda.synchronize()
    msgs = []
    for li, ln in enumerate(lanes):
        for k in KEYS:
            b = ln["bad"][k].cpu().numpy()
            if b.any():
                fr = np.nonzero(b)[0]
                msgs.append("lane%d.%s: %d differing records over frames %s" % (li, k, int(b.sum()), fr[:12].tolist()))
        for nm in ("gl", "gr"):
            f = ln[nm].debug_fault()
            if f:
                msgs.append("lane%d.%s count fault %s" % (li, nm, f))
    print("step %d: %s" % (i, "; ".join(msgs) if msgs else "all outputs identical to the first step's"), flush=True)
    return bool(msgs)
def detail():
    """the first differing (lane, left/right, frame): keypoints per octave then and now, and the records only one of the two holds"""
    for li, ln in enumerate(lanes):
        for side in ("l", "r"):
            b = (ln["bad"]["c" + side] + ln["bad"]["k" + side]).cpu().numpy()
            if not b.any():
                continue
            f = int(np.nonzero(b)[0][0])
            n0, n1 = int(ln["ref"]["c" + side][f, 0]), int(ln["t"]["c" + side][f, 0])
            k0, k1 = ln["ref"]["k" + side][f, :n0].cpu().numpy(), ln["t"]["k" + side][f, :n1].cpu().numpy()
            print("lane%d %s frame %d: count %d -> %d (the last step's)" % (li, side, f, n0, n1))
            print("  per octave then:", np.bincount(np.ascontiguousarray(k0).view(np.int32)[:, 5], minlength=8).tolist())
            print("  per octave now :", np.bincount(np.ascontiguousarray(k1).view(np.int32)[:, 5], minlength=8).tolist())
            rec = lambda k: {(float(r[0]), float(r[1]), float(r[3]), float(r[4]), int(o)) for r, o in zip(k.tolist(), np.ascontiguousarray(k).view(np.int32)[:, 5].tolist())}
            s0, s1 = rec(k0), rec(k1)
            only0, only1 = sorted(s0 - s1, key=lambda r: (r[4], r[1], r[0])), sorted(s1 - s0, key=lambda r: (r[4], r[1], r[0]))
            print("  only then (%d):" % len(only0)); [print("   ", r) for r in only0[:16]]
            print("  only now  (%d):" % len(only1)); [print("   ", r) for r in only1[:16]]
            return
for ln in lanes:
    step(ln, -1)
torch.cuda.synchronize()
if PROFILING:
    for ln in lanes:
        ln["gl"].set_profiling(True)
for i in range(steps):
    step(lanes[i % NL], i)
    if i % SYNC_EVERY == SYNC_EVERY - 1:
        if report(i + 1) and os.environ.get("STOP_ON_DIFF", "1") == "1":
            detail()
            break
torch.cuda.synchronize()
bad = report(steps)
print("done", flush=True)
sys.exit(1 if bad else 0)
