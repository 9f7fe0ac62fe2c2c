"""GPU busy time and idle gaps inside the calibrating forward, from a rocprofv3 kernel trace of tools/probe_first_forward.py:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/ff -- python tools/probe_first_forward.py ant-int-pot-flint bf16
    python tools/first_forward_gaps.py /tmp/ff/**/*kernel_trace.csv
The calibrating forward is the burst of launches that contains the calibration kernels (k_calib_*, k_hist*, k_search_*): its
window is taken from the first to the last of them; busy = union of kernel intervals inside it."""
import csv, sys, re

rows = []
for p in [a for a in sys.argv[1:] if not a.startswith('--')]:
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
cal = [i for i, r in enumerate(rows) if re.search(r"k_calib|k_hist|k_search|k_sort", r[2])]
# the probe runs two repetitions: split the calibration launches at the largest pause between them
gaps = sorted(((rows[cal[i + 1]][0] - rows[cal[i]][1], i) for i in range(len(cal) - 1)), reverse=True)
cut = gaps[0][1]
for name, idx in (("first repetition", cal[:cut + 1]), ("second repetition (under cProfile)", cal[cut + 1:])):
    lo, hi = idx[0], idx[-1]
    w = rows[lo:hi + 1]
    t0, t1 = w[0][0], max(r[1] for r in w)
    busy, end = 0, t0
    idle = []
    for s, e, _ in w:
        if s > end:
            idle.append(s - end)
            busy += e - s
        else:
            busy += max(0, e - end)
        end = max(end, e)
    idle.sort()
    n = len(idle)
    print("%s: window %.2f ms, %d kernels, GPU busy %.2f ms (%.0f %%); %d idle gaps: median %.1f us, 90th percentile %.1f us, max %.1f us, sum %.2f ms" % (
        name, (t1 - t0) / 1e6, len(w), busy / 1e6, 100.0 * busy / (t1 - t0), n, idle[n // 2] / 1e3 if n else 0.0,
        idle[int(n * 0.9)] / 1e3 if n else 0.0, idle[-1] / 1e3 if n else 0.0, sum(idle) / 1e6))
    if "--names" in sys.argv or True:
        import collections
        cnt = collections.Counter(re.sub(r"\(.*", "", r[2]).replace("void ", "")[:90] for r in w)
        dur = collections.Counter()
        for s_, e_, nme in w:
            dur[re.sub(r"\(.*", "", nme).replace("void ", "")[:90]] += e_ - s_
        for nme, c in cnt.most_common(28):
            print("    %5d x %-92s %8.2f ms" % (c, nme, dur[nme] / 1e6))
