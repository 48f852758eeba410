"""Single-call latency of the host-pointer SearchByBoW entry point against the CPU restatement (GPU box: python tests/measure_match_latency.py)."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import oracle_py as O
from orbslam3_cpp_fork_amd import synth
from orbslam3_cpp_fork_amd.matcher import ORBmatcher, flatten_feature_vectors
frames = synth.make_stream(480, 640, 2, camera_idx=4)
oex = O.OracleExtractor(1000, 1.2, 8, 20, 7)
_, kA, dA = oex(frames[0], (0, 1000)); _, kB, dB = oex(frames[1], (0, 1000))
voc = synth.make_vocabulary(10, 4, seed=11); ov = O.OracleVocabulary(10, 4, *voc)
def fv(d):
    _, _, ni, no, fi, _, _ = ov.transform(d, 2)
    return {int(ni[i]): fi[no[i]:no[i + 1]].tolist() for i in range(len(ni))}
fA, fB = fv(dA), fv(dB)
m = ORBmatcher(0.7, True)
def t(fn, n=30):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
g = flatten_feature_vectors(fA, fB)
print("SearchByBoW brute force : gpu %.3f ms  cpu %.3f ms" % (t(lambda: m.SearchByBoW(dA, kA["angle"], None, None, dB, kB["angle"], None)), t(lambda: O.match_groups(dA, dB, kA["angle"], kB["angle"], 50, 0.7, False, True), 5)))
print("SearchByBoW grouped     : gpu %.3f ms  cpu %.3f ms" % (t(lambda: m.SearchByBoW(dA, kA["angle"], fA, None, dB, kB["angle"], fB)), t(lambda: O.match_groups(dA, dB, kA["angle"], kB["angle"], 50, 0.7, False, True, g), 20)))
