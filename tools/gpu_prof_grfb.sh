#!/bin/bash
out=gpurun_out/$1; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o p -- python tools/prof_grfb.py > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python - > $out/grfb.txt <<PY
import csv, collections
rows = sorted(csv.DictReader(open("$f")), key=lambda r: int(r["Start_Timestamp"]))
# sections start at arange markers (elementwise kernel named ...arange...); keep the LAST iteration of each (level, phase)
sec, cur = {}, None
for r in rows:
    k = r["Kernel_Name"]
    if "arange" in k.lower():
        cur = []
        key = len([1 for _ in sec])  # placeholder, replaced below
        sec.setdefault("order", []).append(cur)
        continue
    if cur is not None: cur.append(r)
order = sec.get("order", [])
# 3 iterations x (fwd, bwd) per level, levels in sequence
nl = len(order) // 6
for lvl in range(nl):
    for ph, name in ((0, "fwd"), (1, "bwd")):
        last = order[lvl * 6 + 4 + ph]
        agg = collections.OrderedDict(); tot = 0
        for r in last:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64]
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d; tot += d
        print(f"== level {lvl} {name}: launches {len(last)} total {tot:.1f} us")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40 if lvl == 0 else 12]: print(f"{v[1]:8.1f} us  x{v[0]:3d}  {k}")
PY
rm -rf $out/prof
tail -3 $out/prof.log
