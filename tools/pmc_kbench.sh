#!/bin/bash
# SQ counters of one kbench configuration: tools/pmc_kbench.sh <tag> <kbench arguments...>
# (one rocprofv3 pass; prints per-kernel averages of the counters)
set -o pipefail
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $out -o c -- python3 tools/kbench.py --rounds 2 --iters 5 "$@" > $out/kbench.log 2> $out/err.log || exit 1
python3 - "$out" <<'P'
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"raster_\w+(<[^>]*>)?", r["Kernel_Name"])
        if m:
            acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    print(k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())}, "launches", len(next(iter(c.values()))))
P
