#!/bin/bash
# Rehearsal of the multi-rank training step on ONE GPU: two ranks share it, gradients travel through
# gloo (RCCL refuses two ranks on one device).  gloo moves CUDA buckets at some tens of MB/s, so use
# a small network (--subdivide 1 --img_size 64 --tex_size 2: ~2 s per step); the default one
# (~100 M parameters) takes minutes per step.  usage: rehearse_ddp.sh [extra cmr_synthetic.py flags]
export SV3D_DIST_BACKEND=gloo+cuda
exec timeout -k 5 ${REHEARSE_TIMEOUT:-120} python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port ${REHEARSE_PORT:-29533} examples/cmr_synthetic.py --num_iter 4 --batch_size 2 \
  --print_scalars 0 --checkpoint_dir /tmp/sv3d_rehearse "$@"
