#!/usr/bin/env python3
"""MFMA utilisation per kernel family from one rocprofv3 counter pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; no tracing
flags): MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE * SIMDs) as in rocprofv3's own derived metric (ROCm 7.2 has no
gfx950 section; this is the gfx94x formula: GRBM_GUI_ACTIVE is summed over the 8 XCDs, 1024 SIMDs).
usage: pmc_mfma.py <counter_collection.csv | compact per-kernel .csv.gz of tools/pmc_compact.py> <out.json>
(from the compact form only the time-weighted figure exists: it holds per-kernel sums, not per-dispatch values)"""
import collections
import csv
import json
import re
import sys

FAMILIES = [
    ("conv fwd/dgrad 3x3, 8-wave LDS-DMA tile kernel, 16 rows x 128 couts (conv3x3_tile_kernel<4,2,4,2>)", re.compile(r"conv3x3_tile_kernel<4, 2, 4, 2")),
    ("conv fwd/dgrad 3x3, tile kernel, 8 rows x 128 couts (conv3x3_tile_kernel<2,2,4,2>)", re.compile(r"conv3x3_tile_kernel<2, 2, 4, 2")),
    ("conv fwd/dgrad 3x3, tile kernel, 32 rows x 64 couts (conv3x3_tile_kernel<4,2,8,1>)", re.compile(r"conv3x3_tile_kernel<4, 2, 8, 1")),
    ("conv fwd/dgrad 3x3, tile kernel, 16 rows x 64 couts (conv3x3_tile_kernel<2,2,8,1>)", re.compile(r"conv3x3_tile_kernel<2, 2, 8, 1")),
    ("conv fwd/dgrad 3x3, tile kernel, all shapes", re.compile(r"conv3x3_tile_kernel<")),
    ("conv fwd/dgrad 3x3, 64-wide cout tiles (conv_igemm_pipe_kernel<2,3,3,2>)", re.compile(r"conv_igemm_pipe_kernel<2, 3, 3, 2")),
    ("conv fwd/dgrad 3x3, tall tiles (conv_igemm_pipe_kernel<1,3,3,4>)", re.compile(r"conv_igemm_pipe_kernel<1, 3, 3, 4")),
    ("conv fwd/dgrad 1x1 (conv_igemm_pipe_kernel<*,1,1,2>)", re.compile(r"conv_igemm_pipe_kernel<\d, 1, 1, 2")),
    ("conv wgrad 3x3, wave-specialised (conv_wgrad_ws_kernel<9>)", re.compile(r"conv_wgrad_ws_kernel<9")),
    ("conv wgrad 3x3, wave-specialised, merged launches (conv_wgrad_ws_multi_kernel<9, *>)", re.compile(r"conv_wgrad_ws_multi_kernel<9")),
    ("conv wgrad 3x3, merged launches of 2x2-block layers (conv_wgrad_ws_multi_kernel<9, 1>)", re.compile(r"conv_wgrad_ws_multi_kernel<9, 1")),
    ("conv 3x3 32->32, weights in registers (conv3x3_wreg_kernel)", re.compile(r"conv3x3_wreg_kernel")),
    ("1x1 backward, dx + dW in one pass (conv1x1_bwd_kernel)", re.compile(r"conv1x1_bwd_kernel")),
    ("conv wgrad 7-tap rows, wave-specialised (conv_wgrad_ws_kernel<7>)", re.compile(r"conv_wgrad_ws_kernel<7")),
    ("conv wgrad 1x1 (conv_wgrad_kernel<bf16,1>)", re.compile(r"conv_wgrad_kernel<.*bf16_t, 1")),
]


def main_compact():
    import gzip
    per = collections.defaultdict(lambda: collections.defaultdict(float))      # kernel -> counter -> sum
    n = {}
    for r in csv.DictReader(gzip.open(sys.argv[1], "rt")):
        per[r["Kernel_Name"]][r["Counter_Name"]] += float(r["sum"])
        n[r["Kernel_Name"]] = int(r["dispatches"])
    out = {}
    for fam, rx in FAMILIES:
        ks = [k for k in per if rx.search(k) and per[k]["GRBM_GUI_ACTIVE"] > 0]
        if not ks:
            continue
        busy, gui = sum(per[k]["SQ_VALU_MFMA_BUSY_CYCLES"] for k in ks), sum(per[k]["GRBM_GUI_ACTIVE"] for k in ks)
        out[fam] = {"launches": sum(n[k] for k in ks), "mfma_util_pct_time_weighted": round(100.0 * busy / (gui / 8.0 * 1024.0), 1)}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))


def main():
    if sys.argv[1].endswith(".gz"):
        return main_compact()
    per = collections.defaultdict(lambda: collections.defaultdict(float))      # dispatch -> counter -> value
    name = {}
    for r in csv.DictReader(open(sys.argv[1])):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = r["Kernel_Name"]
    out = {}
    for fam, rx in FAMILIES:
        rows = [(c["SQ_VALU_MFMA_BUSY_CYCLES"], c["GRBM_GUI_ACTIVE"]) for d, c in per.items() if rx.search(name[d]) and c["GRBM_GUI_ACTIVE"] > 0]
        if not rows:
            continue
        utils = sorted(100.0 * b / (g / 8.0 * 1024.0) for b, g in rows)          # GRBM_GUI_ACTIVE: sum over 8 XCDs; 1024 SIMDs
        tot = 100.0 * sum(b for b, _ in rows) / (sum(g for _, g in rows) / 8.0 * 1024.0)
        out[fam] = {"launches": len(rows), "mfma_util_pct_time_weighted": round(tot, 1), "best_launch_pct": round(utils[-1], 1),
                    "median_launch_pct": round(utils[len(utils) // 2], 1)}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
