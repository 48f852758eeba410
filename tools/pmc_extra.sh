#!/bin/bash
# Three more counter passes over the one-chunk-in-flight bench: what each kernel keeps busy (VALU / LDS / VMEM / scalar issue, texture
# addresser, LDS conflicts) and what stops the dispatcher from placing more of its workgroups (SPI resource-allocation stalls).
# Usage (GPU box): tools/pmc_extra.sh <tag>  -> gpurun_out/pmcx_<tag>/summary.txt
set -u
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcx_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-pcie --no-single-call --lanes 1 --steps 3 --warmup 1"
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $OUT/a -- python3 $R/bench.py $B > $OUT/a.log 2>&1 || echo "pass a failed"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TA_BUSY_avr TD_TD_BUSY_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $OUT/b -- python3 $R/bench.py $B > $OUT/b.log 2>&1 || echo "pass b failed"
timeout -k 10 300 rocprofv3 --pmc SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_BAR_CU_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN SPI_CSN_BUSY --output-format csv -d $OUT/c -- python3 $R/bench.py $B > $OUT/c.log 2>&1 || echo "pass c failed"
cd $R
python3 - "$OUT" > $OUT/summary.txt <<'PY'
import collections, csv, glob, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\b(k_[a-z0-9_]+)", r["Kernel_Name"])
        if m:
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for n in sorted(c):
        print("   %-32s %14.1f" % (n, sum(c[n]) / len(c[n])))
PY
cat $OUT/summary.txt
