#!/bin/bash
# The driver's exact bench command: three plain runs + one under rocprofv3 --kernel-trace with the
# per-launch trace kept.   gpurun --timeout 900 -- 'bash tools/driver_cmd.sh r04'
set -u
TAG=${1:-r06}
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
for e in j["config"].get("configs", []):
    if "pass_ms" in e:       # a calibration pass: compute-bound, no HBM fraction
        print("   %-34s %8.3f ms  %8.1f G candidate-evals/s  %s" % (e["name"], e["pass_ms"], e["gcand_evals_per_s"], e["kernel"]))
        continue
    print("   %-34s %8.2f us  frac %.4f  %s%s" % (e["name"], e["launch_us"], e["frac"], e["kernel"],
          "  traffic x%.4f" % e["traffic_over_algorithmic"] if e.get("traffic_over_algorithmic") else ""))
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
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_fq_hbatch<antq::bf16_tag, false>" in r["Kernel_Name"]]     # the headline kernel
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

# every antq kernel of the profiled run by (kernel, grid): the config.configs[] entries use distinct grids, so each entry's
# launch_us can be checked against the trace (the stats CSV pools launches of one kernel name over all configs)
python3 - "$KT" > "$OUT/${TAG}_bench_driver_cmd_kernels_by_grid.txt" <<'PY'
import csv, sys, collections
g = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "antq::" not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("antq::bf16_tag", "bf16").replace("antq::f16_tag", "f16")
    g[(name, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# kernel, grid (work-items), launches, mean us, min us, max us, mean of the LAST half (steady state) -- rocprofv3 --kernel-trace of the driver's bench command")
for (name, grid), d in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    h = d[len(d) // 2:]
    print("%-70s grid %10d  n=%5d  mean %10.2f  min %10.2f  max %10.2f  last-half mean %10.2f" % (name[:70], grid, len(d), sum(d) / len(d), min(d), max(d), sum(h) / len(h)))
PY
head -40 "$OUT/${TAG}_bench_driver_cmd_kernels_by_grid.txt"
