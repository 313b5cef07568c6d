#!/usr/bin/env python3
"""Per-step kernel time of the training example from a rocprofv3 kernel trace: the steps are told
apart by the RGB forward raster launch (one per step), and only the last `--steps` of them are
counted so that library start-up and tuning kernels stay out.
usage: train_step_breakdown.py <s_kernel_trace.csv> [--steps 30] [--top 25]"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]{0,60}>)?)", name)
    return (m.group(1) if m else name)[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(a.trace))]
    rows.sort()
    marks = [s for s, e, n in rows if "raster_forward_kernel<3" in n or "raster_forward_kernel<1" in n]
    marks = sorted(marks)
    # one mark per step: keep marks at least 2 ms apart
    step_t = []
    for m in marks:
        if not step_t or m - step_t[-1] > 2_000_000:
            step_t.append(m)
    step_t = step_t[-(a.steps + 1):]
    lo, hi = step_t[0], step_t[-1]
    n = len(step_t) - 1
    sel = [(s, e, k) for s, e, k in rows if lo <= s < hi]
    busy = sum(e - s for s, e, k in sel)
    # union of busy intervals (kernels can overlap across streams)
    union, cur_s, cur_e = 0, None, None
    for s, e, k in sel:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        union += cur_e - cur_s
    print("%d steps, %.2f ms per step wall (first raster launch to the next), %d kernels per step" % (n, (hi - lo) / n / 1e6, len(sel) // n))
    print("kernel time per step %.2f ms (sum), %.2f ms (GPU busy), idle %.2f ms" % (busy / n / 1e6, union / n / 1e6, ((hi - lo) - union) / n / 1e6))
    tot, cnt = collections.Counter(), collections.Counter()
    for s, e, k in sel:
        tot[short(k)] += e - s
        cnt[short(k)] += 1
    for k, t in tot.most_common(a.top):
        print("%7.1f us/step %5.1f%%  x%-5.1f %s" % (t / n / 1e3, 100.0 * t / busy, cnt[k] / n, k))
    groups = collections.Counter()
    for k, t in tot.items():
        g = ("sv3d raster" if "raster_" in k else "sv3d other" if re.search(r"chamfer|laplacian|flatten|texture_|edt_|distance|project_|face_|mesh_|spmv", k)
             else "conv / gemm libraries" if re.search(r"conv|Conv|gemm|Cijk|igemm|Im2d|transpose|ck::", k)
             else "batch norm" if "BatchNorm" in k else "optimizer (multi_tensor)" if "multi_tensor" in k else "torch elementwise / reduce / other")
        groups[g] += t
    print("by group:")
    for g, t in groups.most_common():
        print("%7.1f us/step %5.1f%%  %s" % (t / n / 1e3, 100.0 * t / busy, g))


if __name__ == "__main__":
    main()
