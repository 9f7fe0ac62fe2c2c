#!/usr/bin/env python3
"""Condense a rocprofv3 counter_collection.csv into mean counter values per (short) kernel name."""
import collections
import csv
import re
import statistics as st
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("antq::", "")
    return name[:70]


def main(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        n = max(len(v) for v in acc[k].values())
        print("%-72s n=%d" % (k, n))
        for c, v in sorted(acc[k].items()):
            print("    %-28s %14.1f" % (c, st.mean(v)))


if __name__ == "__main__":
    main(sys.argv[1:])
