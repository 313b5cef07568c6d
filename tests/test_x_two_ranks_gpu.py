"""The synthetic training step under DistributedDataParallel on the GPU box: two ranks sharing the
one GPU.  This file starts processes and depends on their timing, so it is collected after the
parity tests (pytest collects files in name order)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_stay_in_lockstep(tmp_path):
    """The whole step under DistributedDataParallel with device-bound wrappers, two ranks sharing
    this GPU (gradients through gloo: RCCL wants a GPU per rank): both DDP-wrapped modules -- the
    network and the part basis with its own optimizer -- end with the same weights on both ranks."""
    root = ROOT
    env = dict(os.environ, SV3D_DIST_BACKEND="gloo+cuda")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29577", os.path.join(root, "examples", "cmr_synthetic.py"), "--num_iter", "3",
           "--batch_size", "2", "--print_scalars", "0", "--checkpoint_dir", str(tmp_path), "--seg_train", "1",
           "--tex_size", "2", "--subdivide", "1", "--img_size", "64", "--enc_width", "8", "--print_checksum", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=170, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    sums = dict(re.findall(r"rank (\d) weight checksum (\S+)", out.stdout))
    assert set(sums) == {"0", "1"} and sums["0"] == sums["1"], out.stdout[-2000:]
    assert "over 2 rank(s)" in out.stdout
