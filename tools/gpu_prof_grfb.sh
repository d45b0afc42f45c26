#!/bin/bash
out=gpurun_out/$1; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o p -- python tools/prof_grfb.py > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, collections
rows = sorted(csv.DictReader(open("$f")), key=lambda r: int(r["Start_Timestamp"]))
# last iteration only: split at the marker fills is fragile; take the last 1/6 of the rows
n = len(rows) // 6
last = rows[-n:]
agg = collections.OrderedDict()
tot = 0
for r in last:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d; tot += d
print(f"launches {len(last)} total {tot:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"{v[1]:8.1f} us  x{v[0]:3d}  {k}")
PY
rm -rf $out/prof
