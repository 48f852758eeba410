#!/bin/bash
# End-of-round evidence that needs no profiler: the default bench line, the other workloads, the single-call figures, rows N1-N3.
# Usage (GPU box): tools/final_runs.sh   -> gpurun_out/*.log (tools/install_profiles.sh copies them into profiles/)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
python3 bench.py --workload rig_8cam_1920x1080_8lvl_1000feat_extract+match --no-single-call > gpurun_out/rig.log 2> gpurun_out/rig.err
python3 bench.py --workload kitti_stereo_1241x376_8lvl_2000feat_extract+stereo > gpurun_out/stereo.log 2> gpurun_out/stereo.err
python3 bench.py --workload euroc_stream_752x480_8lvl_1000feat_extract+match --no-single-call --no-cpu-baseline > gpurun_out/euroc.log 2> gpurun_out/euroc.err
python3 bench.py --bow --no-single-call --no-cpu-baseline --no-pcie > gpurun_out/bow.log 2> gpurun_out/bow.err
python3 bench.py --rows > gpurun_out/rows.log 2> gpurun_out/rows.err
python3 tools/dbg_single_frame.py > gpurun_out/single_frame.txt 2>&1
tools/trace_single.sh final 480 640 1000 > gpurun_out/trace_single_vga.txt 2>&1
python3 tools/lds_calibrate.py > gpurun_out/lds_calibration.json 2> gpurun_out/lds_calibration.err
for f in bench_default rig stereo euroc bow; do python3 - $f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("gpurun_out/%s.log" % sys.argv[1]) if l.startswith('{"metric')][-1])
    print(sys.argv[1], d["value"], d["unit"], "ms/step", d["ms_per_step"], "pcie", (d.get("pcie_inclusive") or {}).get("value"), "single", d.get("single_call_ms"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cut -c1-140 gpurun_out/single_frame.txt
