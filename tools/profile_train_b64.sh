#!/bin/bash
# rocprofv3 kernel trace of the training example at 64 dataset images per step, one camera hypothesis
# (the batch-64 reading of the reference's training configuration).  Outputs under gpurun_out/prof_train_b64/.
set -o pipefail
out=gpurun_out/prof_train_b64
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- \
    python examples/cmr_synthetic.py --batch_size 64 --seg_train 0 --num_hypo_cams 1 --flip_pair 0 --num_parts 0 \
    --print_scalars 0 --checkpoint_dir /tmp/sv3d_prof_b64 --name prof --dataset_size 1024 --save_latest_freq 1000000 \
    --warmup_iter 6 --num_iter 16 "$@" > $out/run.log 2> $out/run.err || exit 1
cat $out/run.log
t=$(ls $out/stats/*/s_kernel_trace.csv $out/stats/s_kernel_trace.csv 2>/dev/null | head -n 1)
python tools/train_step_breakdown.py "$t" --steps 8 --top 40 > $out/breakdown.txt
cat $out/breakdown.txt
