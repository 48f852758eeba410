#!/bin/bash
# SQ_INSTS_VALU per launch per kernel for the default bench with one chunk in flight (deterministic: the metric for instruction-count
# work on a chip that is VALU-issue-bound as a whole) plus the kernel times of a second, traced run.  Usage: tools/quick_valu.sh <tag>
set -u
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/qv_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $OUT/pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-single-call --lanes 1 > $OUT/bench_pmc.log 2>&1 || echo "step failed (rc $?)"
[ -n "${QUICK_VALU_NO_TRACE:-}" ] || timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-pcie --no-single-call --lanes 1 > $OUT/bench.log 2>&1 || echo "step failed (rc $?)"
cd $R
python3 - "$OUT" <<'PY'
import collections, csv, glob, re, sys
d = sys.argv[1]
f = glob.glob(d + "/pmc/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "SQ_INSTS_VALU":
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        if m:
            agg[m.group(1)].append(float(r["Counter_Value"]))
t = {}
ff = glob.glob(d + "/trace/**/*kernel_stats.csv", recursive=True)
for r in (csv.DictReader(open(ff[0])) if ff else []):
    m = re.search(r"(k_\w+)", r["Name"])
    if m:
        t[m.group(1)] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    per = sum(v) / len(v)
    n = 7 if k == "k_resize_level" else 1
    tot += per * n
    print("%-20s %8.2f M wave-instr per launch   %8.1f us" % (k, per / 1e6, t.get(k, (0, 0))[1]))
print("per step: %.1f M" % (tot / 1e6))
PY
