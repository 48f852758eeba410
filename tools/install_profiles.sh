#!/bin/bash
# Copy the summaries of one tools/collect_profiles.sh run (gpurun_out/prof_<tag>/, scratch) and of the workload runs made beside it
# (gpurun_out/bench_default.log, rig.log, stereo.log, single_frame.txt) into profiles/ under the round's names.
# Usage: tools/install_profiles.sh <tag> [round prefix, default r02]
set -eu
TAG=$1
R=${2:-r03}
P=gpurun_out/prof_$TAG
newest() { ls -t $1 | head -1; }  # gpurun merges runs into the same scratch tree: an earlier run's files may lie beside the latest
cp "$(newest ''$P'/trace_default/runc/*_kernel_stats.csv')" profiles/${R}_default_kernel_stats.csv
cp "$(newest ''$P'/trace/runc/*_kernel_stats.csv')" profiles/${R}_one_chunk_kernel_stats.csv
grep '^{"metric' $P/bench_trace_default.log | tail -1 > profiles/${R}_default_bench.json
grep '^{"metric' $P/bench_trace.log | tail -1 > profiles/${R}_one_chunk_bench.json
cp "$(newest ''$P'/pmc_fetch/runc/*_counter_collection.csv')" profiles/${R}_pmc_fetch_size.csv
cp "$(newest ''$P'/pmc_write/runc/*_counter_collection.csv')" profiles/${R}_pmc_write_size.csv
cp "$(newest ''$P'/pmc_sq/runc/*_counter_collection.csv')" profiles/${R}_pmc_sq.csv
cp "$(newest ''$P'/pmc_tcc/runc/*_counter_collection.csv')" profiles/${R}_pmc_tcc_hit_miss.csv
cp "$(newest ''$P'/calib/runc/*_counter_collection.csv')" profiles/${R}_pmc_calibration.csv
cp $P/valu_calibration.json profiles/${R}_valu_calibration.json
cp $P/summary.json profiles/${R}_pmc_summary.json
python3 tools/valu_mix.py profiles/${R}_valu_calibration.json > profiles/${R}_valu_mix.json 2>/dev/null
python3 tools/valu_summary.py $P profiles/${R}_valu_mix.json > profiles/${R}_valu_summary.json
[ -f gpurun_out/single_frame.txt ] && grep "ms" gpurun_out/single_frame.txt | cut -c1-420 > profiles/${R}_single_frame.txt
[ -f gpurun_out/pmcx_${R}/summary.txt ] && cp gpurun_out/pmcx_${R}/summary.txt profiles/${R}_pmc_pipes_and_spi.txt
[ -f gpurun_out/lds_calibration.json ] && cp gpurun_out/lds_calibration.json profiles/${R}_lds_calibration.json
[ -f gpurun_out/trace_single_vga.txt ] && cp gpurun_out/trace_single_vga.txt profiles/${R}_single_call_timeline.txt
[ -f gpurun_out/bench_default.log ] && grep '^{"metric' gpurun_out/bench_default.log | tail -1 > profiles/${R}_bench_untraced.json
[ -f gpurun_out/rig.log ] && grep '^{"metric' gpurun_out/rig.log | tail -1 > profiles/${R}_bench_rig.json
[ -f gpurun_out/stereo.log ] && grep '^{"metric' gpurun_out/stereo.log | tail -1 > profiles/${R}_bench_stereo.json
echo installed
