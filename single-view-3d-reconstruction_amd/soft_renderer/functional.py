"""`soft_renderer.functional` subset: OBJ I/O and the functional rasterizer entry."""
import numpy as np
import torch

import sv3d


def load_obj(filename_obj, normalization=False, load_texture=False, texture_res=4,
             texture_type='surface'):
    """Vertices [V,3] float32 and faces [F,3] int32 of a Wavefront OBJ (polygons are fanned
    into triangles); tensors are placed on the current GPU when there is one."""
    if load_texture:
        raise NotImplementedError('load_texture=True is not supported')
    verts, faces = [], []
    with open(filename_obj) as fh:
        for line in fh:
            p = line.split()
            if not p:
                continue
            if p[0] == 'v':
                verts.append([float(t) for t in p[1:4]])
            elif p[0] == 'f':
                idx = [int(t.split('/')[0]) - 1 for t in p[1:]]
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    device = 'cuda' if torch.cuda.is_available() else 'cpu'
    vertices = torch.tensor(np.asarray(verts, dtype=np.float32), device=device)
    faces = torch.tensor(np.asarray(faces, dtype=np.int32), device=device)
    if normalization:
        vertices = vertices - vertices.min(0).values[None]
        vertices = vertices / vertices.abs().max()
        vertices = vertices * 2
        vertices = vertices - vertices.max(0).values[None] / 2
    return vertices, faces


def save_obj(filename, vertices, faces):
    vertices = torch.as_tensor(vertices).detach().cpu().numpy()
    faces = torch.as_tensor(faces).detach().cpu().numpy()
    with open(filename, 'w') as fh:
        for v in vertices:
            fh.write('v %.8f %.8f %.8f\n' % (v[0], v[1], v[2]))
        fh.write('\n')
        for f in faces:
            fh.write('f %d %d %d\n' % (f[0] + 1, f[1] + 1, f[2] + 1))


def face_vertices(vertices, faces):
    """vertices [B,V,3], faces [B,F,3] -> [B,F,3,3]."""
    return sv3d.face_gather(vertices, faces)


def soft_rasterize(face_vertices, textures, image_size=256, background_color=(0, 0, 0), near=1,
                   far=100, fill_back=True, eps=1e-3, sigma_val=1e-5, dist_func='euclidean',
                   dist_eps=1e-4, gamma_val=1e-4, aggr_func_rgb='softmax', aggr_func_alpha='prod',
                   texture_type='surface'):
    """Functional form of the rasterizer: RGBA [B,4,image_size,image_size] (no anti-aliasing)."""
    if dist_func != 'euclidean' or aggr_func_rgb != 'softmax' or aggr_func_alpha != 'prod':
        raise NotImplementedError('only euclidean distance, softmax colour and product alpha are supported')
    return sv3d.soft_rasterize(face_vertices, textures, image_size, anti_aliasing=False,
                               texture_type=texture_type, background_color=background_color,
                               near=near, far=far, eps=eps, sigma_val=sigma_val, dist_eps=dist_eps,
                               gamma_val=gamma_val, fill_back=fill_back)
