#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export ORBX_BUILD_TUNING=1
python3 -c "from orbslam3_cpp_fork_amd import build; build.build(force=True)" || exit 1
for tb in 2560 3072 4096 6144 8192; do
  export ORBX_FAST_TILE_BYTES_LATENCY=$tb
  echo "latency tile bytes $tb"; tools/trace_single.sh lt$tb 480 640 1000 2>&1 | grep "fast_tiles\|span"
  tools/trace_single.sh lk$tb 376 1241 2000 2>&1 | grep "fast_tiles\|span"
done
unset ORBX_BUILD_TUNING ORBX_FAST_TILE_BYTES_LATENCY
python3 -c "from orbslam3_cpp_fork_amd import build; build.build(force=True)"
