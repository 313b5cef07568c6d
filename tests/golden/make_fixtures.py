"""Generate the golden fixtures under tests/golden/ from the reference checkout.

Run from the repo root in a container that has /root/reference mounted:

    python tests/golden/make_fixtures.py

It imports only the reference modules that are pure numpy/torch
(utils/meshzoo.py, utils/mesh.py, nnutils/geom_utils.py) and stores their
*outputs* for seeded inputs, plus the teapot mesh / silhouette data files of
misc/examples/data.  Nothing here runs on the GPU box; the fixtures do.

Fixtures written:
  mesh_sub{1,2,3,4}.npz  icosphere verts/faces (utils/mesh.py:12-16)
  mesh_sub3_sym.npz      make_symmetric ordering (utils/mesh.py:19-60)
  proj_vectors.npz       orthographic_proj_withz / orthographic_proj / quat_rotate /
                         hamilton_product outputs (nnutils/geom_utils.py:49-161)
  teapot.npz             misc/examples/data/teapot.obj vertices+faces,
                         example2_ref.png as a float mask
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("SV3D_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, REF)
    from utils import mesh as ref_mesh            # noqa: E402
    from nnutils import geom_utils as ref_geom    # noqa: E402

    for sub in (1, 2, 3, 4):
        v, f = ref_mesh.create_sphere(sub)
        np.savez_compressed(os.path.join(HERE, "mesh_sub%d.npz" % sub),
                            verts=v.astype(np.float32), faces=f.astype(np.int32))
        print("sub", sub, v.shape, f.shape)

    v, f = ref_mesh.create_sphere(3)
    sv, sf, n_ind, n_sym, nf_ind, nf_sym = ref_mesh.make_symmetric(v, f)
    np.savez_compressed(os.path.join(HERE, "mesh_sub3_sym.npz"),
                        verts=sv.astype(np.float32), faces=sf.astype(np.int32),
                        counts=np.array([n_ind, n_sym, nf_ind, nf_sym], dtype=np.int32))
    print("sym", n_ind, n_sym, nf_ind, nf_sym)

    g = torch.Generator().manual_seed(1234)
    B, N = 4, 37
    X = torch.randn(B, N, 3, generator=g)
    cams = torch.empty(B, 7)
    cams[:, 0] = 0.6 + 0.3 * torch.rand(B, generator=g)
    cams[:, 1:3] = 0.2 * torch.rand(B, 2, generator=g) - 0.1
    q = torch.randn(B, 4, generator=g)
    cams[:, 3:] = q / q.norm(dim=1, keepdim=True)
    qa = torch.randn(B, N, 4, generator=g)
    qb = torch.randn(B, N, 4, generator=g)
    with torch.no_grad():
        withz = ref_geom.orthographic_proj_withz(X, cams, offset_z=5.)
        withz0 = ref_geom.orthographic_proj_withz(X, cams)
        ortho = ref_geom.orthographic_proj(X, cams)
        rot = ref_geom.quat_rotate(X, cams[:, 3:])
        ham = ref_geom.hamilton_product(qa, qb)
    np.savez_compressed(os.path.join(HERE, "proj_vectors.npz"),
                        X=X.numpy(), cams=cams.numpy(), qa=qa.numpy(), qb=qb.numpy(),
                        withz_off5=withz.numpy(), withz_off0=withz0.numpy(),
                        ortho=ortho.numpy(), quat_rotate=rot.numpy(), hamilton=ham.numpy())

    # teapot data files (plain-text OBJ: v / f records only)
    verts, faces = [], []
    with open(os.path.join(REF, "misc/examples/data/teapot.obj")) as fh:
        for line in fh:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                verts.append([float(t) for t in p[1:4]])
            elif p[0] == "f":
                idx = [int(t.split("/")[0]) - 1 for t in p[1:]]
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    verts = np.asarray(verts, dtype=np.float32)
    faces = np.asarray(faces, dtype=np.int32)
    from PIL import Image
    ref_img = np.asarray(Image.open(os.path.join(REF, "misc/examples/data/example2_ref.png")).convert("RGB"),
                         dtype=np.float32).mean(-1) / 255.
    np.savez_compressed(os.path.join(HERE, "teapot.npz"), verts=verts, faces=faces,
                        example2_ref=(ref_img > 0.5).astype(np.uint8))
    print("teapot", verts.shape, faces.shape, ref_img.shape, ref_img.mean())


if __name__ == "__main__":
    main()
