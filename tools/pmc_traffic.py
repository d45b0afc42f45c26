#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 counter-collection runs of the same command (one --pmc FETCH_SIZE, one --pmc
WRITE_SIZE; never combined with tracing flags).  FETCH_SIZE is doubled (gfx950: 128-B requests tallied at 64 B for 16-B/lane
streaming reads, /opt/skills/guides/MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as reported.  Both are in KB.
The output has one entry per kernel INSTANTIATION, keyed by the kernel name as egm_conv_kernel_name() / a kernel trace spell it (template
arguments included, "void", "(anonymous namespace)::" and the parameter list stripped), so that bench.py can put the dominant kernel's own
bytes beside its own algorithmic bytes; the coarser family table of rounds 1-2 is kept under "_families".
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import re
import sys

FAMILIES = [
    ("conv_fwd[3x3] (conv_igemm_pipe_kernel<*,3,3,*>)", re.compile(r"conv_igemm_pipe_kernel<\d+, 3, 3, \d+")),
    ("conv_fwd[1x1] (conv_igemm_pipe_kernel<*,1,1,*>)", re.compile(r"conv_igemm_pipe_kernel<\d+, 1, 1, \d+")),
    ("conv_wgrad[3x3] (conv_wgrad_ws_kernel<9> / conv_wgrad_kernel<bf16,9>)", re.compile(r"conv_wgrad_ws_kernel<9|conv_wgrad_kernel<.*bf16_t, 9")),
    ("conv_wgrad[1x1] (conv_wgrad_kernel<bf16,1> / conv_wgrad_ws_kernel<1>)", re.compile(r"conv_wgrad_kernel<.*bf16_t, 1|conv_wgrad_ws_kernel<1")),
    ("bn_ew (fused BatchNorm + element-wise, fwd / bwd reduce / bwd apply)", re.compile(r"bn_ew_")),
    ("bn multi-tensor passes (GRFB branches)", re.compile(r"bn_\w*multi_kernel")),
    ("bn_act_bwd_apply", re.compile(r"bn_act_bwd_apply_kernel")),
    ("bn_act_bwd_reduce (channel_partials_kernel<bf16,1>)", re.compile(r"channel_partials_kernel<.*bf16_t, 1>")),
    ("bn_act_fwd", re.compile(r"bn_act_fwd_kernel")),
    ("wgrad_reduce_multi", re.compile(r"wgrad_reduce_multi_kernel")),
    ("conv_wgrad[dilated rows] (conv_wgrad_kernel<bf16,3>)", re.compile(r"conv_wgrad_kernel<.*bf16_t, 3")),
    ("mca_fused_fwd", re.compile(r"mca_fused_fwd_kernel")),
    ("upcat_fwd (in place: upsampled half only)", re.compile(r"upcat_fwd_kernel")),
    ("maxpool2_fwd", re.compile(r"maxpool2_fwd_kernel")),
]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    return m.group(1) if m else name


def collect_by_kernel(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    return tot, cnt


def collect(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        for fam, rx in FAMILIES:
            if rx.search(r["Kernel_Name"]):
                tot[fam] += float(r["Counter_Value"]); cnt[fam] += 1
                break
    return tot, cnt


def main():
    f_tot, f_cnt = collect(sys.argv[1], "FETCH_SIZE")
    w_tot, w_cnt = collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for fam, _ in FAMILIES:
        if not f_cnt[fam] or not w_cnt[fam]:
            continue
        fetch_kb, write_kb = f_tot[fam] / f_cnt[fam], w_tot[fam] / w_cnt[fam]
        out[fam] = {"launches_in_trace": f_cnt[fam], "fetch_kb_raw": round(fetch_kb, 1), "write_kb": round(write_kb, 1),
                    "hbm_bytes_per_launch_corrected": int(round((2.0 * fetch_kb + write_kb) * 1024))}
    kf_tot, kf_cnt = collect_by_kernel(sys.argv[1], "FETCH_SIZE")
    kw_tot, kw_cnt = collect_by_kernel(sys.argv[2], "WRITE_SIZE")
    per = {}
    for k in kf_tot:
        if not kw_cnt[k]:
            continue
        fetch_kb, write_kb = kf_tot[k] / kf_cnt[k], kw_tot[k] / kw_cnt[k]
        per[k] = {"launches_in_trace": kf_cnt[k], "fetch_kb_raw": round(fetch_kb, 1), "write_kb": round(write_kb, 1),
                  "hbm_bytes_per_launch_corrected": int(round((2.0 * fetch_kb + write_kb) * 1024))}
    per = dict(sorted(per.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch_corrected"] * kv[1]["launches_in_trace"]))
    per["_families"] = out
    json.dump(per, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: v for k, v in list(per.items())[:12]}, indent=1))


if __name__ == "__main__":
    main()
