#!/usr/bin/env python3
"""Host-side profile (cProfile) of the eager training step of examples/cmr_synthetic.py at the
bench configuration (batch 4, part maps on): where the host spends the step it is bound by."""
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-view-3d-reconstruction_amd"))
sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch  # noqa: E402


def main():
    import cmr_synthetic as cmr
    o = cmr.parse(["--batch_size", "4", "--seg_train", "1", "--print_scalars", "0", "--checkpoint_dir",
                   tempfile.mkdtemp(prefix="sv3d_prof_"), "--name", "prof", "--dataset_size", "1024",
                   "--save_latest_freq", "1000000"] + sys.argv[1:])
    o.epoch_fraction = 1.0
    t = cmr.ShapeTrainer(o)
    t.init_training()
    o.num_iter = 8
    t.train()
    torch.cuda.synchronize()
    o.num_iter = 30
    t0 = time.perf_counter()
    t.train()
    torch.cuda.synchronize()
    print("eager step %.2f ms" % ((time.perf_counter() - t0) / 30 * 1e3))
    # phases of one step on the host, each followed by a device sync (so: host + device time of the phase)
    batch = next(iter(t.dataloader))
    scaler = t._scaler
    for rep in range(2):
        marks = []
        torch.cuda.synchronize(); a = time.perf_counter()
        t.set_input(batch); t._zero_grads()
        with torch.autocast("cuda", dtype=torch.float16):
            t.forward(0)
        b = time.perf_counter(); torch.cuda.synchronize(); b2 = time.perf_counter()
        scaler.scale(t.total_loss).backward()
        c = time.perf_counter(); torch.cuda.synchronize(); c2 = time.perf_counter()
        t._optimizer_steps(scaler)
        d = time.perf_counter(); torch.cuda.synchronize(); d2 = time.perf_counter()
        print("host issue time: forward %.2f ms (+%.2f to drain), backward %.2f (+%.2f), optimizer %.2f (+%.2f)"
              % ((b - a) * 1e3, (b2 - b) * 1e3, (c - b2) * 1e3, (c2 - c) * 1e3, (d - c2) * 1e3, (d2 - d) * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    t.train()
    torch.cuda.synchronize()
    pr.disable()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
        print(s.getvalue()[:6000])


if __name__ == "__main__":
    main()
