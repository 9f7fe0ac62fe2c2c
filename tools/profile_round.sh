#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box.  Run from the repo root through gpurun:
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh r03'
# Everything is written under gpurun_out/<tag>/ (small text files only); copy what should be judged into profiles/.
set -u
TAG=${1:-r04}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
PY="python"

# the driver's exact command line first (--steps 20 --warmup 5: three plain runs, one under rocprofv3 --kernel-trace with the
# per-launch durations kept), then the long run
# (SKIP_DRIVER=1: the part below the kernel trace only -- round 6: the traced runs of a command that times ten configs make
#  rocprofv3 write tens of thousands of launches; the long runs therefore pass --configs none)
if [ "${SKIP_DRIVER:-0}" != "1" ]; then
bash tools/driver_cmd.sh "$TAG" > "$OUT/driver_cmd.log" 2>&1
tail -32 "$OUT/driver_cmd.log" | head -30
$PY bench.py --steps 200 --warmup 20 > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.stderr"
tail -1 "$OUT/${TAG}_bench.json" | cut -c1-400

# the launch form the driver uses for N > 1 (one rank per GPU over RCCL), on the one GPU of this box
$PY -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 \
    --no-cpu-baseline 2> "$OUT/bench_torchrun.stderr" | tail -1 > "$OUT/${TAG}_bench_torchrun_n1.json"
cut -c1-200 "$OUT/${TAG}_bench_torchrun_n1.json"

# kernel trace of the same command (no PMC in this pass)
( cd /tmp && rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- \
    $PY "$REPO/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-traffic --configs none > "$OUT/bench_profiled.json" 2> /tmp/prof_kt.err )
KS=$(find /tmp/prof_kt -name '*kernel_stats.csv' | head -1)
[ -n "$KS" ] && cp "$KS" "$OUT/${TAG}_bench_kernel_stats.csv"
tail -1 "$OUT/bench_profiled.json" | cut -c1-200
fi

# HBM traffic of the bench command: one counter per pass, kernel trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section)
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/prof_$C && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -- \
      $PY "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --configs headline_olive > /dev/null 2> /tmp/prof_$C.err )
done
$PY tools/pmc_summary.py $(find /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE -name '*counter_collection.csv') \
    > "$OUT/${TAG}_pmc_summary.txt" 2>&1

# SQ / LDS / memory counters of every hot kernel
bash tools/profile_counters.sh "$TAG" > "$OUT/counters.log" 2>&1
[ -f "$REPO/gpurun_out/${TAG}_pmc_kernels.txt" ] && mv "$REPO/gpurun_out/${TAG}_pmc_kernels.txt" "$OUT/"

$PY tools/bench_configs.py       > "$OUT/${TAG}_configs.log" 2>&1
$PY tools/probe_grid_types.py    > "$OUT/${TAG}_grid_types.log" 2>&1
$PY tools/probe_operator_level.py > "$OUT/${TAG}_operator_level.log" 2>&1
$PY tools/probe_weight_bank.py 2>&1 | grep -v "golden,\|bit \s" > "$OUT/${TAG}_weight_bank.log"
$PY tools/probe_graph_forward.py 2>&1 | grep -v "bit \s\|amdgpu" > "$OUT/${TAG}_graph_forward.log"
$PY tools/probe_size_sweep.py 2>&1 | grep -v amdgpu > "$OUT/${TAG}_size_sweep.log"
$PY tools/probe_group_sweep.py 2>&1 | grep -v amdgpu > "$OUT/${TAG}_group_sweep.log"
$PY tools/probe_lane_u.py 2>&1 | grep -v amdgpu > "$OUT/${TAG}_lane_task_u.log"
$PY tools/probe_codec.py 2>&1 | grep -v amdgpu > "$OUT/${TAG}_codec.log"
$PY tools/probe_search.py 2>&1 | grep -v amdgpu > "$OUT/${TAG}_search.log"
( $PY tools/probe_absmax.py; $PY tools/probe_affine.py ) 2>&1 | grep -v amdgpu > "$OUT/${TAG}_aux_kernels.log"
( $PY tools/probe_lane_rows.py; $PY tools/probe_lane_rows_np2.py; $PY tools/probe_batch_lane.py ) 2>&1 | grep -v amdgpu > "$OUT/${TAG}_lane_rows.log"
( $PY tools/bench_sharded.py --model opt6.7b; $PY tools/bench_sharded.py --model llama70b --inplace ) 2>&1 | grep "^{" > "$OUT/${TAG}_sharded.log"
# round 3: launch shapes (wavefronts per workgroup, task size, unordered launches), footprint, call overhead
$PY tools/probe_launch_shapes.py 3 2>&1 | grep -v "amdgpu\|^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" > "$OUT/${TAG}_launch_shapes.log"
$PY tools/probe_footprint.py 2 2>&1 | grep -v "amdgpu\|^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" > "$OUT/${TAG}_footprint.log"
( $PY tools/probe_call_overhead.py; ANTQ_NO_EXT=1 $PY tools/probe_call_overhead.py ) 2>&1 | grep -v "amdgpu\|^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" > "$OUT/${TAG}_call_overhead.log"
$PY tools/probe_box.py 2>&1 | grep -v "amdgpu\|^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" > "$OUT/${TAG}_box.log"
[ -x tools/exp_lane ] && ./tools/exp_lane > "$OUT/${TAG}_exp_lane.log" 2>&1
# round 4: the 16-bit-domain kernels -- launch-shape experiment, row-length A/B against the round-3 kernels, one launch per tensor
[ -x tools/exp_hrow ] && ( ./tools/exp_hrow > "$OUT/${TAG}_exp_hrow_ant.log" 2>&1; ./tools/exp_hrow olive > "$OUT/${TAG}_exp_hrow_olive.log" 2>&1 )
( $PY tools/probe_hrow_rows.py; $PY tools/probe_hrow_rows.py olive ) 2>&1 | grep -v amdgpu > "$OUT/${TAG}_hrow_rows.log"
$PY tools/probe_per_tensor.py 2>&1 | grep -v amdgpu > "$OUT/${TAG}_per_tensor_shapes.log"
$PY tools/probe_transient.py 40 1.5 2>&1 | grep -v amdgpu > "$OUT/${TAG}_transient.log"
( $PY tools/probe_first_forward.py; $PY tools/probe_first_forward.py ant-int-pot-flint ) 2>&1 | grep "forward" > "$OUT/${TAG}_first_forward.log"
( $PY tools/probe_precalibrate.py; $PY tools/probe_precalibrate.py flint ) 2>&1 | grep "forward\|precalibrate" > "$OUT/${TAG}_precalibrate.log"
$PY tools/probe_search_split.py 2>&1 | grep -v amdgpu > "$OUT/${TAG}_search_split.log"
$PY tools/probe_fp32_occupancy.py 2>&1 | grep -v amdgpu > "$OUT/${TAG}_fp32_occupancy.log"
[ -x tools/stream_shapes ] && ./tools/stream_shapes all > "$OUT/${TAG}_stream_shapes.log" 2>&1
[ -x tools/launch_anatomy ] && ./tools/launch_anatomy 2>&1 | cut -c1-70 > "$OUT/${TAG}_launch_anatomy.log"
[ -x tools/valu_rates ] && ./tools/valu_rates > "$OUT/${TAG}_valu_rates.log" 2>&1
[ -x tools/ubench ] && ( cd tools && ./ubench 2>&1 | head -38; $PY probe_lane_rows.py 2>&1 | grep -v "amdgpu\|OliVe" | head -5 ) > "$OUT/${TAG}_ubench_copy_variants.log"
ls -la "$OUT"
