#!/bin/bash
# Phase ablation of k_fast_tiles (tuning build: -DORBX_TUNING, ORBX_DBG_SKIP_FAST = 1: staging only, 8: + dense pre-test loop, 2: + compaction, 4: + score, 0: all).
# Usage (on the GPU box): tools/fast_ablation.sh [workload args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export ORBX_BUILD_TUNING=1
python3 -c "from orbslam3_cpp_fork_amd import build; build.build(force=True)" || exit 1
for m in 1 8 2 4 0; do
  ORBX_DBG_SKIP_FAST=$m python3 bench.py --no-pcie --no-cpu-baseline --no-single-call --lanes 1 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('skip=$m fast_ms', (d.get('stage_ms_one_chunk_in_flight') or d['stage_ms'])['fast'], 'value', d['value'])"
done
unset ORBX_BUILD_TUNING
python3 -c "from orbslam3_cpp_fork_amd import build; build.build(force=True)"
