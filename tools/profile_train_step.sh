#!/bin/bash
# rocprofv3 kernel-trace statistics of the synthetic CMR-style training step
# (examples/cmr_synthetic.py, batch 4 per rank, part maps on).  Outputs under gpurun_out/prof_train_<tag>/.
# usage: profile_train_step.sh <tag> [extra example flags, e.g. --cuda_graph 1]
set -o pipefail
tag=${1:-eager}
shift
out=gpurun_out/prof_train_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- \
    python examples/cmr_synthetic.py --batch_size 4 --seg_train 1 --print_scalars 0 --checkpoint_dir /tmp/sv3d_prof_$tag \
    --name prof --dataset_size 1024 --save_latest_freq 1000000 --warmup_iter 8 --num_iter 40 "$@" > $out/run.log 2> $out/run.err || exit 1
cat $out/run.log
f=$(ls $out/stats/*/s_kernel_stats.csv $out/stats/s_kernel_stats.csv 2>/dev/null | head -n 1)
cp "$f" $out/kernel_stats.csv
python - "$out/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over the run; top kernels:" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%6.2f%%  calls %6s  avg %9.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"],
          float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
