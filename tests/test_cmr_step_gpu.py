"""GPU test of the synthetic CMR-style training step (examples/cmr_synthetic.py): the render /
Chamfer call pattern of one reference training iteration through the Trainer loop."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


def test_training_steps_run_and_learn(tmp_path):
    import cmr_synthetic
    argv = ["--num_iter", "6", "--batch_size", "2", "--img_size", "64", "--subdivide", "2", "--tex_size", "4",
            "--num_parts", "3", "--seg_train", "1", "--checkpoint_dir", str(tmp_path), "--print_freq", "2",
            "--dataset_size", "32", "--learning_rate", "1e-3"]
    o = cmr_synthetic.parse(argv)
    o.epoch_fraction = 1.0
    t = cmr_synthetic.ShapeTrainer(o)
    t.init_training()
    before = {k: v.detach().clone() for k, v in t.model.state_dict().items() if v.dtype.is_floating_point}
    w2 = t.part_basis_generator.w.detach().clone()
    assert t.train() == 6
    assert torch.isfinite(t.total_loss)
    after = t.model.state_dict()
    moved = [k for k in before if not torch.equal(before[k], after[k])]
    assert any(k.startswith("shape") for k in moved) and any(k.startswith("cam.") for k in moved)
    assert any(k.startswith("tex") for k in moved) and any(k.startswith("encoder") for k in moved)
    assert not torch.equal(w2, t.part_basis_generator.w)           # second optimizer stepped
    # render batch shape of the reference recipe: Nc * 2B images per render call
    assert t.imgs.shape[0] == 2


def test_two_ranks_on_one_gpu_stay_in_lockstep(tmp_path):
    """The whole step under DistributedDataParallel with device-bound wrappers, two ranks sharing
    this GPU (gradients through gloo: RCCL wants a GPU per rank): both DDP-wrapped modules -- the
    network and the part basis with its own optimizer -- end with the same weights on both ranks."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SV3D_DIST_BACKEND="gloo+cuda")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29577", os.path.join(root, "examples", "cmr_synthetic.py"), "--num_iter", "3",
           "--batch_size", "2", "--print_scalars", "0", "--checkpoint_dir", str(tmp_path), "--seg_train", "1",
           "--tex_size", "2", "--subdivide", "1", "--img_size", "64", "--print_checksum", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=170, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    sums = dict(re.findall(r"rank (\d) weight checksum (\S+)", out.stdout))
    assert set(sums) == {"0", "1"} and sums["0"] == sums["1"], out.stdout[-2000:]
    assert "over 2 rank(s)" in out.stdout
