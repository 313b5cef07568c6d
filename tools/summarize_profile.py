#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/profile_round.sh into one table per kernel.

HBM bytes follow the gfx950 rule for FETCH_SIZE: the counter is in KiB-like units of 1024 B as
rocprofv3 reports it, and reads of wide coalesced streams are tallied at half their size, so
the read side is reported both raw and doubled."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def load(sub):
    files = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(name):
    for k in ("raster_forward_tiled_shared_kernel", "raster_forward_tiled_kernel", "raster_tile_lists_kernel", "raster_setup_lists_kernel", "raster_forward_kernel", "raster_backward_kernel", "raster_setup_kernel", "raster_prep_kernel",
              "chamfer_nn_small_kernel", "chamfer_nn_kernel", "chamfer_grad_kernel", "face_gather_kernel",
              "face_scatter_kernel"):
        if k in name:
            return k
    return None


rows = collections.defaultdict(dict)
for sub in ("sq", "fetch", "write", "lds"):
    for kern, ctrs in load(sub).items():
        s = short(kern)
        if s is None:
            continue
        for c, vals in ctrs.items():
            rows[s][c] = sum(vals) / len(vals)
stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
for f in stats:
    for r in csv.DictReader(open(f)):
        s = short(r["Name"])
        if s:
            rows[s]["avg_ns"] = float(r["AverageNs"])
            rows[s]["calls"] = int(r["Calls"])
summary = {}
for k, v in rows.items():
    d = dict(v)
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes_raw"] = d["FETCH_SIZE"] * 1024
        d["hbm_read_bytes_x2"] = 2 * d["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
    if "SQ_INSTS_VALU" in d and "avg_ns" in d:
        d["valu_inst_per_ns_per_simd"] = d["SQ_INSTS_VALU"] / d["avg_ns"] / 1024.0
    if "SQ_ACTIVE_INST_VALU" in d and "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"]:
        # SQ_ACTIVE_INST_VALU counts cycles (x4 per quad-cycle tick) a VALU instruction is executing,
        # summed over SIMDs; SQ_BUSY_CYCLES is per shader engine.  Reported as the raw ratio.
        d["valu_active_per_busy_cycle"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_BUSY_CYCLES"]
    if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"]:
        d["wait_any_frac"] = d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]
        d["wait_issue_frac"] = d.get("SQ_WAIT_INST_ANY", 0) / d["SQ_WAVE_CYCLES"]
    summary[k] = d
    print(k)
    for c in sorted(d):
        print("    %-28s %s" % (c, ("%.4g" % d[c]) if isinstance(d[c], float) else d[c]))
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
