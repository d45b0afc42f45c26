#!/usr/bin/env python3
"""Per-kernel means of every counter in a rocprofv3 --pmc counter_collection.csv (dispatches of one kernel name pooled, first launch of
each dropped), plus derived shares when the SQ wave-cycle counters are present.
usage: pmc_kernel_table.py <counter_collection.csv> [name filter regex]"""
import collections
import csv
import re
import sys

rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
per = collections.defaultdict(lambda: collections.defaultdict(float))
name = {}
for r in csv.DictReader(open(sys.argv[1])):
    per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    name[int(r["Dispatch_Id"])] = r["Kernel_Name"]
byk = collections.defaultdict(list)
for d in sorted(per):
    if rx is None or rx.search(name[d]):
        byk[name[d]].append(per[d])
for k, rows in byk.items():
    rows = rows[1:] if len(rows) > 1 else rows
    keys = sorted(rows[0])
    mean = {c: sum(r[c] for r in rows) / len(rows) for c in keys}
    print(f"{k[:110]}  ({len(rows)} launches)")
    for c in keys:
        print(f"    {c:32s} {mean[c]:16.0f}")
    wc = mean.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_FLAT"):
            if c in mean:
                print(f"    {c + ' / SQ_WAVE_CYCLES':40s} {100 * mean[c] / wc:6.1f} %")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean and mean["GRBM_GUI_ACTIVE"] > 0:
        print(f"    MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) = {100 * mean['SQ_VALU_MFMA_BUSY_CYCLES'] / (mean['GRBM_GUI_ACTIVE'] / 8 * 1024):.1f} %")
    if "SQ_LDS_BANK_CONFLICT" in mean and mean.get("SQ_LDS_IDX_ACTIVE"):
        print(f"    LDS bank-conflict cycles / LDS active cycles = {100 * mean['SQ_LDS_BANK_CONFLICT'] / mean['SQ_LDS_IDX_ACTIVE']:.1f} %")
