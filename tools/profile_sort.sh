#!/bin/bash
# SQ / LDS counters of the sorted-row search kernels (rocprofv3 --pmc, kernel trace only; one group of counters per pass).
#   gpurun --timeout 900 -- 'bash tools/profile_sort.sh r06'
set -u
TAG=${1:-r06}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for GROUP in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "GRBM_GUI_ACTIVE FETCH_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmcs_$i
  rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d /tmp/pmcs_$i -- python "$REPO/tools/profile_sort_target.py" > /tmp/pmcs_$i.log 2>&1
done
python "$REPO/tools/pmc_summary.py" $(find /tmp/pmcs_1 /tmp/pmcs_2 /tmp/pmcs_3 -name '*counter_collection.csv') \
  | awk '/^[^ ]/ {keep = ($0 ~ /^k_search_sorted|^k_sort/)} keep' > "$OUT/${TAG}_pmc_sort.txt"
cat "$OUT/${TAG}_pmc_sort.txt"
