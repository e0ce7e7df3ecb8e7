#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (units as reported: KB for
FETCH_SIZE / WRITE_SIZE). Usage: pmc_summary.py FETCH.csv WRITE.csv > profiles/rNN_pmc.txt
FETCH_SIZE on gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM);
the 'x2' column applies that correction."""
import collections
import csv
import sys


def agg(path):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        c = out.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += float(r["Counter_Value"])
    return out


fetch, write = agg(sys.argv[1]), agg(sys.argv[2])
print("%-58s %6s %14s %14s %14s" % ("kernel", "calls", "FETCH MB/launch", "FETCHx2 MB", "WRITE MB/launch"))
for k in fetch:
    n, v = fetch[k]
    wn, wv = write.get(k, [1, 0.0])
    print("%-58s %6d %14.1f %14.1f %14.1f" % (k[:58], n, v / n / 1024, 2 * v / n / 1024, wv / max(wn, 1) / 1024))
