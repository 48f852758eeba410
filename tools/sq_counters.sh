#!/bin/bash
# SQ counter pass over bench.py (one rocprofv3 --pmc run, 8 SQ slots): tools/sq_counters.sh <tag> [counters...]
TAG=${1:-sq}; shift
CNT=${@:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $CNT --output-format csv -d $OUT -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > $OUT/log.txt 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, re
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m = re.search(r"\b(k_[a-z0-9_]+)", r["Kernel_Name"])
    if m:
        agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    w = m.get("SQ_WAVES", 1.0) or 1.0
    print("%-20s waves=%8.0f " % (k, w) + " ".join("%s=%.1f" % (n.replace("SQ_", ""), v / w) for n, v in sorted(m.items()) if n != "SQ_WAVES"))
PY
