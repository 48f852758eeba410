#!/bin/bash
# One-chunk-in-flight kernel trace of the 8-camera 1080p rig workload (256 frames per step); prints the per-kernel averages.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/qt_rig
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload rig_8cam_1920x1080_8lvl_1000feat_extract+match --steps 5 --warmup 2 --no-cpu-baseline --no-pcie --lanes 1 > $OUT/bench.log 2>&1 || echo "step failed (rc $?)"
cd $R
tail -1 $OUT/bench.log | cut -c1-400
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print("%-70s %5s %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
