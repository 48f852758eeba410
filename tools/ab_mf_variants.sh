#!/bin/bash
# tuning builds of k_bf_mfma with parts switched off (ORBX_MF_VARIANT bits: 1 = no epilogue, 2 = no tile expansion, 4 = no LDS operand reads, 8 = hits found but not parked,
# 16 = parked but never merged): where its time goes (kernel durations from a trace; the results of the variants are wrong by design)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export ORBX_BUILD_TUNING=1
for v in ${VARIANTS:-0 1 8 16}; do
  export ORBX_TUNING_DEFINES="-DORBX_MF_VARIANT=$v"
  python3 -c "from orbslam3_cpp_fork_amd import build; build.build(force=True)" || exit 1
  echo "variant $v"; tools/quick_trace.sh mfv$v 2>&1 | grep "k_bf\|resolve"
done
unset ORBX_BUILD_TUNING ORBX_TUNING_DEFINES
python3 -c "from orbslam3_cpp_fork_amd import build; build.build(force=True)"
