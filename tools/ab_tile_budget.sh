#!/bin/bash
# tuning build: throughput of the default bench against the LDS budget of a FAST tile and the candidate-list size
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export ORBX_BUILD_TUNING=1
python3 -c "from orbslam3_cpp_fork_amd import build; build.build(force=True)" || exit 1
run() { python3 bench.py --no-pcie --no-cpu-baseline --no-single-call "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); s=d.get('stage_ms_one_chunk_in_flight') or d['stage_ms']; print('$TAG', d['value'], d['ms_per_step'], 'fast_solo', s['fast'])"; }
for tb in ${TBS:-8192 10240 12288 16384 8192}; do for cc in ${CCS:-2048}; do
  export ORBX_FAST_TILE_BYTES=$tb ORBX_FAST_CAND_CAP=$cc; TAG="tile=$tb cand=$cc" run
done; done
unset ORBX_BUILD_TUNING ORBX_FAST_TILE_BYTES ORBX_FAST_CAND_CAP
python3 -c "from orbslam3_cpp_fork_amd import build; build.build(force=True)"
