#!/bin/bash
# Phase ablation of k_fast_tiles with counters (tuning build): per ORBX_DBG_SKIP_FAST mode (1: staging only, 8: + dense pre-test loop,
# 2: + compaction, 4: + score, 0: all) the kernel's duration, VALU instructions, LDS instructions, LDS active cycles and bank-conflict cycles.
# Usage (on the GPU box): tools/fast_ablation_pmc.sh <tag>
TAG=${1:-abl}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fastabl_$TAG
mkdir -p $OUT
export ORBX_BUILD_TUNING=1
cd $R && python3 -c "from orbslam3_cpp_fork_amd import build; build.build(force=True)" || exit 1
cd /tmp && export TMPDIR=/tmp
for m in 1 8 2 4 0; do
  export ORBX_DBG_SKIP_FAST=$m
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_$m -- python3 $R/bench.py --steps 3 --warmup 1 --no-pcie --no-cpu-baseline --no-single-call --lanes 1 > $OUT/pmc_$m.log 2>&1 || echo "pmc $m failed"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$m -- python3 $R/bench.py --steps 6 --warmup 2 --no-pcie --no-cpu-baseline --no-single-call --lanes 1 > $OUT/tr_$m.log 2>&1 || echo "trace $m failed"
done
unset ORBX_DBG_SKIP_FAST
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for m in (1, 8, 2, 4, 0):
    acc = collections.defaultdict(list)
    for f in glob.glob("%s/pmc_%d/**/*counter_collection.csv" % (out, m), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fast_tiles" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = None
    for f in glob.glob("%s/tr_%d/**/*kernel_stats.csv" % (out, m), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fast_tiles" in r["Name"]:
                dur = float(r["AverageNs"]) / 1e3
    print("skip=%d  dur %s us  " % (m, "%.1f" % dur if dur else "?") + "  ".join("%s %.1fM" % (k.replace("SQ_", ""), sum(v) / len(v) / 1e6) for k, v in sorted(acc.items())))
PY
unset ORBX_BUILD_TUNING
python3 -c "from orbslam3_cpp_fork_amd import build; build.build(force=True)"
