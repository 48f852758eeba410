#!/bin/bash
# same-box A/B of two builds of liborbx.so: new (in tree) vs old (gpurun_old_liborbx.so.bin)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cp orbslam3_cpp_fork_amd/liborbx.so /tmp/new.so
run() { python3 bench.py --no-pcie --no-cpu-baseline --no-single-call "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); s=d.get('stage_ms_one_chunk_in_flight') or d['stage_ms']; print('$TAG', d['value'], d['ms_per_step'], 'fast_solo', s['fast'])"; }
for rep in 1 2 3; do
  cp /tmp/new.so orbslam3_cpp_fork_amd/liborbx.so; TAG=new run
  cp gpurun_old_liborbx.so.bin orbslam3_cpp_fork_amd/liborbx.so; TAG=old run
done
cp /tmp/new.so orbslam3_cpp_fork_amd/liborbx.so
