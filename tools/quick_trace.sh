#!/bin/bash
# One-chunk-in-flight kernel trace of the default bench; prints the per-kernel averages.  Usage (on the GPU box): tools/quick_trace.sh <tag> [env assignments are inherited]
set -u
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/qt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-pcie --lanes 1 > $OUT/bench.log 2>&1 || echo "step failed (rc $?)"
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print("%-70s %5s %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
