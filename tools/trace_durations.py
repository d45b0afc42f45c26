#!/usr/bin/env python3
"""Mean GPU duration per consecutive run of identical (kernel, grid) launches in a rocprofv3 kernel trace (launch order)."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2] if len(sys.argv) > 2 else "conv_"
cur, n, tot, first = None, 0, 0, 0
def flush():
    if cur and n >= 5: print(f"{cur[0][:60]:60s} grid={cur[1]:>6s} n={n:3d} mean={tot / n / 1e3:7.1f} us  first={first / 1e3:7.1f} us")
for r in rows:
    if pat not in r["Kernel_Name"]: continue
    k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], str(int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if k != cur:
        flush(); cur, n, tot, first = k, 0, 0, d
    n += 1; tot += d
flush()
