#!/usr/bin/env python3
"""Keep the raw material of a rocprofv3 --pmc pass in an auditable, small form: one row per (kernel, counter) with the launch count and
the summed counter value, gzip-compressed (the per-dispatch CSV is tens of MB).  usage: pmc_compact.py <counter_collection.csv> <out.csv.gz>"""
import collections
import csv
import gzip
import sys


def main():
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(sys.argv[1])):
        k = (r["Kernel_Name"], r["Counter_Name"])
        tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    with gzip.open(sys.argv[2], "wt", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "dispatches", "sum"])
        for k in sorted(tot):
            w.writerow([k[0], k[1], cnt[k], repr(tot[k])])


if __name__ == "__main__":
    main()
