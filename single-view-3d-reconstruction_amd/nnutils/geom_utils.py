"""Projection and quaternion helpers with the names and argument meaning of the
reference's nnutils/geom_utils.py (reference lines cited per function).

The projection helpers are plain torch (any device, differentiable): the small O(B*V) input
stage of the staged renderer call.  The texture samplers use the HIP sampler on GPU tensors.
"""
import torch


def hamilton_product(qa, qb):
    """Quaternion product qa * qb over the last axis (w, x, y, z); geom_utils.py:108-133."""
    aw, ax, ay, az = qa.unbind(-1)
    bw, bx, by, bz = qb.unbind(-1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz,
                        aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw), dim=-1)


def quat_conj(q):
    """geom_utils.py:135-136."""
    return q * q.new_tensor([1.0, -1.0, -1.0, -1.0])


def quat2ang(q):
    """Rotation angle of a [B,N,4] quaternion, shape [B,N,1]; geom_utils.py:138-141."""
    return 2 * torch.acos(torch.clamp(q[:, :, 0], min=-1 + 1e-6, max=1 - 1e-6)).unsqueeze(-1)


def cross_product(qa, qb):
    """geom_utils.py:83-105."""
    return torch.cross(qa, qb, dim=-1)


def quat_rotate(X, q):
    """Rotate points X [B,N,3] by quaternions q [B,4]: the vector part of q (0,X) q*, as
    geom_utils.py:143-161 computes with two Hamilton products.  q is NOT normalised there,
    so the result carries the factor |q|^2; the closed form below keeps that."""
    w = q[:, None, 0:1]
    u = q[:, None, 1:4]
    uu = (u * u).sum(-1, keepdim=True)
    ux = (u * X).sum(-1, keepdim=True)
    return (w * w - uu) * X + 2.0 * ux * u + 2.0 * w * torch.cross(u.expand_as(X), X, dim=-1)


def orthographic_proj(X, cam):
    """X [B,N,3], cam [B,7] = [scale, tx, ty, quaternion] -> [B,N,2]; geom_utils.py:49-60."""
    rot = quat_rotate(X, cam[:, -4:])
    return cam[:, 0, None, None] * rot[:, :, :2] + cam[:, None, 1:3]


def orthographic_proj_withz(X, cam, offset_z=0.):
    """Like orthographic_proj but keeps (scaled) depth plus offset_z; geom_utils.py:63-80."""
    proj = cam[:, 0, None, None] * quat_rotate(X, cam[:, -4:])
    shift = torch.cat((cam[:, 1:3], cam.new_full((cam.shape[0], 1), float(offset_z))), dim=1)
    return proj + shift[:, None, :]


def _sample(texture_flow, images):
    """[B,F,T*T,C] samples of `images` at `texture_flow`: bilinear, align_corners=True, zero
    padding.  GPU tensors go through the HIP sampler, which writes the channels-last texel layout
    directly; CPU tensors, more than 4 channels or images that need a gradient take torch's
    grid_sample + permute as the reference does."""
    T = texture_flow.size(-2)
    F = texture_flow.size(1)
    grid = texture_flow.reshape(-1, F, T * T, 2)
    if images.is_cuda and images.shape[1] <= 4 and not images.requires_grad:
        import sv3d.texture
        return sv3d.texture.texture_sample(images, grid.reshape(-1, F * T * T, 2)).view(-1, F, T * T, images.shape[1])
    return torch.nn.functional.grid_sample(images, grid, align_corners=True).permute(0, 2, 3, 1)


def sample_textures(texture_flow, images):
    """texture_flow [B,F,T,T,2] in [-1,1], images [B,3,N,N] -> [B,F,T*T,3]; geom_utils.py:11-26."""
    return _sample(texture_flow, images)


def sample_textures_2(texture_flow, images):
    """As sample_textures but shaped [B,F,T,T,C]; geom_utils.py:29-46."""
    T = texture_flow.size(-2)
    F = texture_flow.size(1)
    return _sample(texture_flow, images).reshape(-1, F, T, T, images.shape[1])
