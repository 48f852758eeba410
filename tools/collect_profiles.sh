#!/bin/bash
# Collect the round's profiling evidence on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of the default bench (resident pass only), FETCH_SIZE and WRITE_SIZE passes (separate --pmc runs, as
#   MI355X_MICROARCH.md prescribes), the access-width calibration, a VALU-instruction pass (SQ_INSTS_VALU per dispatch -> achieved
#   wave-instructions per second against the measured issue ceiling of tools/valu_calibrate.py), an L2 hit / miss pass
#   (TCC_HIT_sum, TCC_MISS_sum: north_star's "L2 hit rate for the BRIEF pass"), and JSON summaries.
# Usage: tools/collect_profiles.sh <tag>      -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# The default command keeps 3 chunks in flight: its kernel durations contain the share of the other chunks' kernels running beside them
# (trace_default = the summary the bench line's roofline.avg_launch_ms must agree with).  Everything that prices ONE kernel -- solo
# duration, HBM bytes, VALU instructions -- is collected with one chunk in flight.
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 $R/bench.py --no-cpu-baseline --no-pcie --no-single-call > $OUT/bench_trace_default.log 2>&1 || echo "step failed (rc $?)"
B="--no-cpu-baseline --no-pcie --no-single-call --lanes 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $B > $OUT/bench_trace.log 2>&1 || echo "step failed (rc $?)"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 $B > $OUT/bench_fetch.log 2>&1 || echo "step failed (rc $?)"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 $B > $OUT/bench_write.log 2>&1 || echo "step failed (rc $?)"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 $B > $OUT/bench_sq.log 2>&1 || echo "step failed (rc $?)"
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 $R/bench.py --steps 3 --warmup 1 $B > $OUT/bench_tcc.log 2>&1 || echo "step failed (rc $?)"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib -- python3 $R/tools/pmc_calibrate.py > $OUT/calib.log 2>&1 || echo "step failed (rc $?)"
python3 $R/tools/valu_calibrate.py > $OUT/valu_calibration.json 2> $OUT/valu_calibration.err || echo "step failed (rc $?)"
cd $R
python3 tools/summarize_profiles.py $OUT > $OUT/summary.json && python3 tools/valu_mix.py $OUT/valu_calibration.json > $OUT/valu_mix.json 2>/dev/null && python3 tools/valu_summary.py $OUT $OUT/valu_mix.json > $OUT/valu_summary.json && cat $OUT/valu_summary.json
