#!/usr/bin/env python3
"""Condense rocprofv3 counter_collection.csv files into mean counter values per (short) kernel name and grid size, plus
the ratios that matter for a streaming kernel:
    VALU / wave        SQ_INSTS_VALU / SQ_WAVES            (instructions one wavefront executes)
    LDS conflicts      SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
    busy / wait        SQ_ACTIVE_INST_ANY, SQ_WAIT_ANY (parked at s_waitcnt / barrier), SQ_WAIT_INST_ANY (issue stall),
                       each over SQ_WAVE_CYCLES
    HBM bytes          FETCH_SIZE x 2 (gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes
"""
import collections
import csv
import re
import statistics as st
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("antq::", "").replace("bf16_tag", "bf16").replace("f16_tag", "f16")
    return name[:70]


def main(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            key = (short(r["Kernel_Name"]), r.get("Grid_Size", "?"))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        c = {n: st.mean(v) for n, v in acc[k].items()}
        n = max(len(v) for v in acc[k].values())
        print("%-72s grid %-10s n=%d" % (k[0], k[1], n))
        for name, v in sorted(c.items()):
            print("    %-28s %14.1f" % (name, v))
        d = []
        if c.get("SQ_WAVES") and "SQ_INSTS_VALU" in c:
            d.append("VALU / wave %.0f" % (c["SQ_INSTS_VALU"] / c["SQ_WAVES"]))
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d.append("LDS conflict cycles %.1f %%" % (100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]))
        if c.get("SQ_WAVE_CYCLES"):
            w = c["SQ_WAVE_CYCLES"]
            d.append("of wave cycles: issuing %.1f %%, parked (waitcnt / barrier) %.1f %%, issue-stalled %.1f %%" % (
                100 * c.get("SQ_ACTIVE_INST_ANY", 0) / w, 100 * c.get("SQ_WAIT_ANY", 0) / w, 100 * c.get("SQ_WAIT_INST_ANY", 0) / w))
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            d.append("HBM bytes %.0f (FETCH_SIZE x 2 + WRITE_SIZE)" % ((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024))
        for line in d:
            print("    => " + line)


if __name__ == "__main__":
    main(sys.argv[1:])
