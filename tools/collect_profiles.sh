#!/bin/bash
# Collect the round's profiling evidence on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of the default bench, FETCH_SIZE and WRITE_SIZE passes (separate --pmc runs, as
#   MI355X_MICROARCH.md prescribes), the access-width calibration, and a JSON summary.
# Usage: tools/collect_profiles.sh <tag>      -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib -- python3 $R/tools/pmc_calibrate.py > $OUT/calib.log 2>&1 || exit 1
cd $R
python3 tools/summarize_profiles.py $OUT > $OUT/summary.json && cat $OUT/summary.json
