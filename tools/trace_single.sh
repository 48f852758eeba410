#!/bin/bash
# Kernel timeline of ONE synchronous orbx_extract call (the last of 40): start offset, duration and gap before every kernel.
# Usage (on the GPU box): tools/trace_single.sh <tag> [rows cols nfeatures [eager]]
set -u
TAG=${1:-s}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/st_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/dbg_single_calls.py "$@" > $OUT/run.log 2>&1 || echo "step failed (rc $?)"
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]) for r in csv.DictReader(open(f))))
# calls are separated by host-side gaps > 30 us
calls, cur = [], [rows[0]]
for x in rows[1:]:
    if x[0] - cur[-1][1] > 30000:
        calls.append(cur)
        cur = []
    cur.append(x)
calls.append(cur)
last = calls[-1]
t0 = last[0][0]
print("kernels of the last call (%d calls seen): span %.1f us, sum of kernels %.1f us" % (len(calls), (last[-1][1] - t0) / 1e3, sum(b - a for a, b, _ in last) / 1e3))
prev = t0
for a, b, n in last:
    print("  +%7.1f us  gap %5.1f  dur %6.1f  %s" % ((a - t0) / 1e3, (a - prev) / 1e3, (b - a) / 1e3, n))
    prev = b
PY
