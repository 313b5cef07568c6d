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


def test_captured_step_matches_the_eager_step(tmp_path):
    """--cuda_graph 1: after the eager steps the trainer replays forward + loss + backward from a
    HIP graph.  On one batch and one set of weights the replayed step must give the loss and the
    gradients of the eager step (fp32 here, so the only difference is the summation order of
    library kernels), and training must keep moving the weights."""
    import cmr_synthetic
    argv = ["--num_iter", "6", "--batch_size", "2", "--img_size", "64", "--subdivide", "2", "--tex_size", "4",
            "--num_parts", "3", "--seg_train", "1", "--checkpoint_dir", str(tmp_path), "--print_scalars", "0",
            "--dataset_size", "32", "--learning_rate", "1e-3", "--cuda_graph", "1", "--amp", "0"]
    o = cmr_synthetic.parse(argv)
    o.epoch_fraction = 1.0
    t = cmr_synthetic.ShapeTrainer(o)
    t.init_training()
    before = t.model.shape.weight.detach().clone() if hasattr(t.model, "shape") and hasattr(t.model.shape, "weight") else None
    assert t.train() == 6                                   # 3 eager steps, capture, 3 replayed steps
    assert t._graph is not None and torch.isfinite(t.total_loss)
    if before is not None:
        assert not torch.equal(before, t.model.shape.weight)
    params = [p for m in (t.model, t.part_basis_generator) for p in m.parameters() if p.requires_grad]
    batch = next(iter(t.dataloader))
    scaler = t._scaler
    # replayed
    t.set_input(batch)
    t._forward_backward_graphed(0, scaler)
    loss_g = float(t.total_loss.detach())
    grads_g = [p.grad.detach().clone() if p.grad is not None else None for p in params]
    # eager, same weights and batch (fresh gradient tensors: the captured ones belong to the graph)
    saved = [p.grad for p in params]

    def eager():
        t.set_input(batch)
        for p in params:
            p.grad = None
        t._forward_backward(0, scaler)
        return float(t.total_loss.detach()), [p.grad.detach().clone() if p.grad is not None else None for p in params]

    def rel(ga, gb):
        num = sum(float((a - b).double().pow(2).sum()) for a, b in zip(ga, gb) if a is not None and b is not None)
        den = sum(float(b.double().pow(2).sum()) for b in gb if b is not None)
        assert den > 0
        return (num / den) ** 0.5

    loss_e, grads_e = eager()
    for p, g in zip(params, saved):
        p.grad = g
    assert abs(loss_g - loss_e) <= 1e-4 * abs(loss_e), (loss_g, loss_e)
    assert all((a is None) == (b is None) for a, b in zip(grads_g, grads_e))
    # The step is not bit-repeatable from call to call: the convolution library is free to pick
    # another algorithm (its workspace differs under capture), the encoder's output then moves in
    # its last bits, and the 1/sigma slope of the silhouette fragments turns that into 1e-3 .. 1e-2
    # of the gradient.  The same weights and batch with the encoder's output perturbed by one part
    # in 1e6 gives the size of that effect; the replayed step must stay within it.
    with torch.no_grad():
        for p in t.model.shape.parameters():
            p.mul_(1.0 + 1e-6)
    _, grads_p = eager()
    for p, g in zip(params, saved):
        p.grad = g
    floor = rel(grads_p, grads_e)
    got = rel(grads_g, grads_e)
    print("captured vs eager gradient %.3e, one-in-1e6 perturbation %.3e" % (got, floor))
    assert got < max(10.0 * floor, 1e-4) and got < 3e-2, (got, floor)
    # a batch of another shape is refused, not silently replayed
    t.set_input((batch[0][:1], batch[1][:1]))
    with pytest.raises(RuntimeError):
        t._forward_backward_graphed(0, scaler)


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
