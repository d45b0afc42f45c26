#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and the heaviest (kernel, grid) shapes.
usage: prof_summary.py <kernel_trace.csv> <steps_in_trace> [out.md [timeline.tsv]]"""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    # steps == 0: use only the hipGraph replays of the training step -- the windows between consecutive loss_fwd_kernel launches
    # whose launch count is the most common one (warm-up, capture and the instrumented eager step of bench.py differ)
    if steps == 0:
        marks = [i for i, r in enumerate(rows) if "loss_fwd_kernel" in r["Kernel_Name"]]
        wins = [(a, b) for a, b in zip(marks, marks[1:])]
        mode = collections.Counter(b - a for a, b in wins).most_common(1)[0][0]
        wins = [(a, b) for a, b in wins if b - a == mode]
        if len(sys.argv) > 4:                        # timeline: launch order of one step, durations averaged over the windows
            with open(sys.argv[4], "w") as tl:
                for j in range(mode):
                    rs = [rows[a + j] for a, b in wins]
                    d = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / len(rs) / 1e3
                    g = int(rs[0]["Grid_Size_X"]) // max(1, int(rs[0]["Workgroup_Size_X"]))
                    tl.write(f"{j}\t{short(rs[0]['Kernel_Name'])}\t{g}\t{d:.1f}\n")
        rows = [r for a, b in wins for r in rows[a:b]]
        steps = len(wins)
    per_k, per_s = collections.defaultdict(lambda: [0, 0]), collections.defaultdict(lambda: [0, 0])
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = short(r["Kernel_Name"])
        g = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        per_k[k][0] += 1; per_k[k][1] += d
        per_s[(k, g)][0] += 1; per_s[(k, g)][1] += d
    tot = sum(v[1] for v in per_k.values())
    out = [f"# rocprofv3 kernel-trace summary ({path.split('/')[-1]}; {steps} steps in trace)", "",
           f"total kernel time {tot / steps / 1e6:.3f} ms/step over {len(rows) / steps:.0f} launches/step", "",
           "| kernel | launches/step | avg us | ms/step | % |", "|---|---|---|---|---|"]
    for k, v in sorted(per_k.items(), key=lambda kv: -kv[1][1])[:int(__import__('os').environ.get('PROF_TOP', 40))]:
        out.append(f"| {k} | {v[0] / steps:.1f} | {v[1] / v[0] / 1e3:.1f} | {v[1] / steps / 1e6:.3f} | {100 * v[1] / tot:.1f} |")
    out += ["", "## heaviest (kernel, grid-in-workgroups) shapes", "", "| kernel | grid | launches/step | avg us | ms/step |", "|---|---|---|---|---|"]
    for (k, g), v in sorted(per_s.items(), key=lambda kv: -kv[1][1])[:40]:
        out.append(f"| {k} | {g} | {v[0] / steps:.1f} | {v[1] / v[0] / 1e3:.1f} | {v[1] / steps / 1e6:.3f} |")
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
