#!/bin/bash
# Wide fuzz run on an MI355X (the -m gpu suite runs the same tests with a handful of seeds):
#   gpurun -- 'bash tools/fuzz_campaign.sh 1500 120 300'
# $1 seeds for test_fuzz_every_launch_form_long_rows (10 tensors each, every launch form against the oracle),
# $2 seeds for the arbitrary-codebook / arbitrary-shape fuzz tests, $3 seeds (5 calibrations each) for the calibration fuzz,
# $4 seeds (random models) for the calibration-pass fuzz.  Summary -> gpurun_out/fuzz_campaign.log
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
# ONLY_SORT=1: just the sections of the sorted-row search (round 6)
{
  if [ "${ONLY_SORT:-0}" != "1" ]; then
  echo "== test_fuzz_every_launch_form_long_rows, ${1:-1500} seeds"
  ANTQ_FUZZ_SEEDS=${1:-1500} timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line -s \
      -k fuzz_every_launch_form 2>&1 | grep -E "MISMATCH|Error|passed|failed" | cut -c1-900 | head -60
  echo "== test_random_grids_fuzz, test_random_grids_fuzz_other_entry_points, test_random_shapes_fuzz, ${2:-120} seeds each"
  ANTQ_FUZZ_SEEDS=${2:-120} timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line \
      -k "random_grids_fuzz or random_shapes_fuzz" 2>&1 | grep -E "Error|passed|failed" | cut -c1-600
  echo "== test_fp16_io_fuzz_every_launch_form, ${2:-120} seeds"
  ANTQ_FUZZ_SEEDS=${2:-120} timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line \
      -k fp16_io_fuzz 2>&1 | grep -E "Error|passed|failed" | cut -c1-600
  echo "== test_calibration_fuzz_random_shapes_vs_oracle, ${3:-300} seeds"
  ANTQ_FUZZ_SEEDS=${3:-300} timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line \
      -k calibration_fuzz 2>&1 | grep -E "Error|passed|failed" | cut -c1-600
  echo "== test_calibration_pass_fuzz_fast_schedule_equals_step_by_step, ${4:-150} seeds"
  ANTQ_FUZZ_SEEDS=${4:-150} timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line \
      -k calibration_pass_fuzz 2>&1 | grep -E "Error|passed|failed" | cut -c1-600
  echo "== the calibration fuzz tests again with the histogram clip search forced for every eligible tensor (ANTQ_DEBUG_KNOBS=14=2)"
  ANTQ_DEBUG_KNOBS=14=2 ANTQ_FUZZ_SEEDS=${3:-300} timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line \
      -k "calibration_fuzz or quantizer_end_to_end or type_selection_on_one_read or sharded_per_tensor or bert_base_real_shapes" 2>&1 | grep -E "Error|passed|failed" | cut -c1-600
  ANTQ_DEBUG_KNOBS=14=2 ANTQ_FUZZ_SEEDS=${4:-150} timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line \
      -k calibration_pass_fuzz 2>&1 | grep -E "Error|passed|failed" | cut -c1-600
  echo "== round 6: the calibration fuzz tests with the threshold-sweep clip search forced for every eligible launch (ANTQ_DEBUG_KNOBS=19=2: rows from 256 elements, one-scale fp32 tensors from 4096, OliVe pairs included)"
  ANTQ_DEBUG_KNOBS=19=2 ANTQ_FUZZ_SEEDS=${3:-300} timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line \
      -k "calibration_fuzz or quantizer_end_to_end or type_selection_on_one_read or sharded_per_tensor or bert_base_real_shapes or calibrate_one_call" 2>&1 | grep -E "Error|passed|failed" | cut -c1-600
  ANTQ_DEBUG_KNOBS=19=2 ANTQ_FUZZ_SEEDS=${4:-150} timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line \
      -k calibration_pass_fuzz 2>&1 | grep -E "Error|passed|failed" | cut -c1-600
  fi
  echo "== round 6: the calibration fuzz tests with the SORTED-ROW clip search forced for every eligible launch (ANTQ_DEBUG_KNOBS=20=2: rows from 128 elements, one-scale fp32 tensors from 4096, OliVe pairs included)"
  ANTQ_DEBUG_KNOBS=20=2 ANTQ_FUZZ_SEEDS=${3:-300} timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line \
      -k "calibration_fuzz or quantizer_end_to_end or type_selection_on_one_read or sharded_per_tensor or bert_base_real_shapes or calibrate_one_call" 2>&1 | grep -E "Error|passed|failed" | cut -c1-600
  ANTQ_DEBUG_KNOBS=20=2 ANTQ_FUZZ_SEEDS=${4:-150} timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line \
      -k calibration_pass_fuzz 2>&1 | grep -E "Error|passed|failed" | cut -c1-600
  echo "== ... and under its default rule (rows >= 128 (ANT) / 256 (OliVe), fp32 one-scale tensors >= 1 M elements)"
  ANTQ_FUZZ_SEEDS=${3:-300} timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line \
      -k "calibration_fuzz" 2>&1 | grep -E "Error|passed|failed" | cut -c1-600
} | tee gpurun_out/fuzz_campaign.log
