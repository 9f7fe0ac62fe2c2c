#!/bin/bash
# SQ / memory counters of every hot kernel (rocprofv3 --pmc, kernel trace only; one group of counters per pass).
#   gpurun --timeout 1500 -- 'bash tools/profile_counters.sh r02'
# Writes gpurun_out/<tag>_pmc_kernels.txt (mean per launch and kernel); copy it into profiles/.
set -u
TAG=${1:-r03}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for GROUP in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d /tmp/pmc_$i -- python "$REPO/tools/profile_targets.py" > /tmp/pmc_$i.log 2>&1
done
python "$REPO/tools/pmc_summary.py" $(find /tmp/pmc_1 /tmp/pmc_2 /tmp/pmc_3 /tmp/pmc_4 -name '*counter_collection.csv') \
  | awk '/^[^ ]/ {keep = ($0 ~ /^k_/)} keep' > "$OUT/${TAG}_pmc_kernels.txt"
wc -l "$OUT/${TAG}_pmc_kernels.txt"
