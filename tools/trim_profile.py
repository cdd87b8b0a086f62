#!/usr/bin/env python3
"""Keep only this library's kernels (names starting with k_) from a rocprofv3
kernel_stats / kernel_trace / counter CSV, so the committed summary stays small.
usage: trim_profile.py in.csv out.csv"""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
with open(src, newline="") as f, open(dst, "w", newline="") as g:
    r = csv.reader(f)
    w = csv.writer(g, quoting=csv.QUOTE_ALL)
    hdr = next(r)
    w.writerow(hdr)
    col = hdr.index("Name") if "Name" in hdr else hdr.index("Kernel_Name")
    for row in r:
        if row[col].startswith(("k_", "void k_")):
            w.writerow(row)
