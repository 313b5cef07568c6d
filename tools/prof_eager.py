import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-view-3d-reconstruction_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from nnutils.softras import SoftNeuralRenderer
from sv3d import meshgen
verts, faces = meshgen.icosphere(3)
B = 8
dev = torch.device("cuda:0")
v = (torch.from_numpy(verts)[None].repeat(B, 1, 1) + 0.02 * torch.randn(B, verts.shape[0], 3)).to(dev).requires_grad_(True)
f = torch.from_numpy(faces).int()[None].repeat(B, 1, 1).to(dev)
cams = torch.tensor([[0.8, 0.05, -0.02, 0.9, 0.1, 0.3, -0.2]]).repeat(B, 1).to(dev).requires_grad_(True)
r = SoftNeuralRenderer(img_size=256)
g = torch.rand(B, 256, 256, device=dev)
def step():
    v.grad = None; cams.grad = None
    m = r(v, f, cams)
    m.backward(g)
for _ in range(50): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(500): step()
torch.cuda.synchronize()
print("eager mask call fwd+bwd: %.1f us" % ((time.perf_counter() - t0) / 500 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(500): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
