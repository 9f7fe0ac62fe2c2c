#!/bin/bash
# The driver's exact bench command: three plain runs + one under rocprofv3 --kernel-trace with the
# per-launch trace kept.   gpurun --timeout 900 -- 'bash tools/driver_cmd.sh r04'
set -u
TAG=${1:-r04}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for i in 1 2 3; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_driver_cmd_$i.json"
  python3 - "$OUT/${TAG}_bench_driver_cmd_$i.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
r = j["roofline"]; p = j["config"]["per_tensor_launches"]
print("driver cmd: frac %.4f launch_us %.2f copy %.1f GB/s x%.4f | per-tensor unordered %.4f ordered %.4f" % (
    r["frac"], r["launch_us"], r["copy_ceiling"]["antq_copy_GBps"], r["copy_ceiling"]["frac_of_copy_ceiling"], p["frac"], p["ordered"]["frac"]))
PY
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_driver_cmd.json"
( cd /tmp && rm -rf /tmp/prof_drv && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_drv -- \
    python3 "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/${TAG}_bench_driver_cmd_profiled.json" 2> /tmp/prof_drv.err )
KT=$(find /tmp/prof_drv -name '*kernel_trace.csv' | head -1)
KS=$(find /tmp/prof_drv -name '*kernel_stats.csv' | head -1)
[ -n "$KS" ] && cp "$KS" "$OUT/${TAG}_bench_driver_cmd_kernel_stats.csv"
python3 - "$KT" > "$OUT/${TAG}_bench_driver_cmd_launches.txt" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_fq_hbatch" in r["Kernel_Name"] or "k_fq_batch" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
print("# per-launch durations of %s under rocprofv3 --kernel-trace, bench.py --gpus 1 --steps 20 --warmup 5" % rows[0]["Kernel_Name"][:60])
print("# launch  duration_us  gap_since_previous_end_us")
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%3d  %8.2f  %s" % (i + 1, (e - s) / 1e3, "%.2f" % ((s - prev_end) / 1e3) if prev_end else "-"))
    prev_end = e
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("# warm-up launches 1-5 mean %.2f us; timed launches 6-25 mean %.2f us = %.4f of 8 TB/s" % (
    sum(d[:5]) / 5, sum(d[5:25]) / 20, 2147483648 / (sum(d[5:25]) / 20 * 1e-6) / 8e12))
PY
cat "$OUT/${TAG}_bench_driver_cmd_launches.txt"
tail -1 "$OUT/${TAG}_bench_driver_cmd_profiled.json" | cut -c1-300
