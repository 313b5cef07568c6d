"""The subset of the third-party `soft_renderer` (SoftRas) object model that the reference
touches (nnutils/softras.py:33-37,45,48,51,64,72-77; nnutils/mesh_net.py:636-638), rebuilt
over the `sv3d` HIP ops.

Pipeline of `SoftRenderer.__call__`: Mesh -> Lighting -> Transform -> SoftRasterizer, as in
SoftRas.  What is supported is what the hot path needs: euclidean distance, product alpha,
z-softmax colour, 'look_at' / 'look' cameras, orthographic or perspective projection,
surface or vertex textures.  Anything else raises instead of silently doing something else.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import sv3d


def _as_batched(x, dtype, device=None):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    if not torch.is_tensor(x):
        x = torch.as_tensor(x)
    if x.dim() == 2:
        x = x[None]
    if device is not None:
        x = x.to(device)
    return x.to(dtype) if x.dtype != dtype else x


def face_normals(fv):
    """Unit normals [B,F,3] of face_vertices [B,F,3,3]."""
    v10 = fv[:, :, 0] - fv[:, :, 1]
    v12 = fv[:, :, 2] - fv[:, :, 1]
    return F.normalize(torch.cross(v12, v10, dim=-1), p=2, dim=2, eps=1e-6)


class Mesh(object):
    """Batched triangle mesh: vertices [B,V,3], faces [B,F,3], textures [B,F,R*R,3]
    ('surface') or [B,V,3] ('vertex')."""

    def __init__(self, vertices, faces, textures=None, texture_res=1, texture_type='surface'):
        if texture_type not in ('surface', 'vertex'):
            raise ValueError('texture_type should be surface or vertex')
        self.vertices = _as_batched(vertices, torch.float32)
        self.device = self.vertices.device
        # [B,F,3], or [1,F,3] for one topology shared by the batch (read once per image)
        self.faces = _as_batched(faces, torch.int32, self.device).contiguous()
        self.texture_type = texture_type
        self.batch_size = self.vertices.shape[0]
        self.num_vertices = self.vertices.shape[1]
        self.num_faces = self.faces.shape[1]
        self.texture_res = texture_res
        if textures is None:
            shape = (self.batch_size, self.num_faces, texture_res ** 2, 3) if texture_type == 'surface' \
                else (self.batch_size, self.num_vertices, 3)
            self.textures = torch.ones(shape, dtype=torch.float32, device=self.device)
        else:
            if not torch.is_tensor(textures):
                textures = torch.as_tensor(np.asarray(textures))
            if textures.dim() == (3 if texture_type == 'surface' else 2):
                textures = textures[None]
            self.textures = textures.to(self.device, torch.float32)
            if texture_type == 'surface':
                self.texture_res = int(round(math.sqrt(self.textures.shape[2])))

    @classmethod
    def from_obj(cls, filename_obj, normalization=False, load_texture=False, texture_res=1,
                 texture_type='surface'):
        from .functional import load_obj
        if load_texture:
            raise NotImplementedError('load_texture=True is not supported (no .mtl / image reader here)')
        vertices, faces = load_obj(filename_obj, normalization=normalization)
        return cls(vertices, faces, texture_res=texture_res, texture_type=texture_type)

    def _topology(self):
        # one shared topology is passed as [1,F,3] so the kernels read it once per image
        return self.faces

    @property
    def face_vertices(self):
        return sv3d.face_gather(self.vertices, self._topology())

    @property
    def face_textures(self):
        if self.texture_type == 'surface':
            return self.textures
        return sv3d.face_gather(self.textures, self._topology())

    @property
    def surface_normals(self):
        return face_normals(self.face_vertices)


# ---- lighting -----------------------------------------------------------------------
def _rgb_scaled(x, color, k):
    """x[..., None] * k * color as [..., 3], from Python scalars only: no host-to-device copy, so
    the lighting can be captured in a HIP graph.  A tensor colour is used as it is."""
    if torch.is_tensor(color):
        return x[..., None] * (k * color.to(x.device, x.dtype))
    c = [float(v) * float(k) for v in np.asarray(color, dtype=np.float64).reshape(-1)]
    return torch.stack((x * c[0], x * c[1], x * c[2]), dim=-1)


class AmbientLighting(nn.Module):
    def __init__(self, light_intensity=0.5, light_color=(1, 1, 1)):
        super(AmbientLighting, self).__init__()
        self.light_intensity = light_intensity
        self.light_color = light_color

    def forward(self, light):
        if torch.is_tensor(self.light_color):
            return light + self.light_intensity * self.light_color.to(light.device, light.dtype)
        c = [float(v) * float(self.light_intensity) for v in np.asarray(self.light_color, dtype=np.float64).reshape(-1)]
        if c[0] == c[1] == c[2]:
            return light + c[0]
        return torch.stack((light[..., 0] + c[0], light[..., 1] + c[1], light[..., 2] + c[2]), dim=-1)


class DirectionalLighting(nn.Module):
    def __init__(self, light_intensity=0.5, light_color=(1, 1, 1), light_direction=(0, 1, 0)):
        super(DirectionalLighting, self).__init__()
        self.light_intensity = light_intensity
        self.light_color = light_color
        self.light_direction = light_direction

    def forward(self, light, normals):
        if torch.is_tensor(self.light_direction):
            d = self.light_direction.to(normals.device, normals.dtype)
            cosine = F.relu((normals * d).sum(-1))
        else:
            d = [float(v) for v in np.asarray(self.light_direction, dtype=np.float64).reshape(-1)]
            cosine = F.relu(normals[..., 0] * d[0] + normals[..., 1] * d[1] + normals[..., 2] * d[2])
        return light + _rgb_scaled(cosine, self.light_color, self.light_intensity)


class Lighting(nn.Module):
    """Ambient + directional lights multiplied into the textures (per face for
    light_mode 'surface')."""

    def __init__(self, light_mode='surface', intensity_ambient=0.5, color_ambient=(1, 1, 1),
                 intensity_directionals=0.5, color_directionals=(1, 1, 1), directions=(0, 1, 0)):
        super(Lighting, self).__init__()
        if light_mode not in ('surface', 'vertex'):
            raise ValueError('Lighting mode only support surface and vertex')
        self.light_mode = light_mode
        self.ambient = AmbientLighting(intensity_ambient, color_ambient)
        self.directionals = nn.ModuleList([DirectionalLighting(intensity_directionals,
                                                               color_directionals, directions)])

    def is_identity(self):
        amb = np.asarray(self.ambient.light_color, dtype=np.float64) * self.ambient.light_intensity
        return bool(np.all(amb == 1.0)) and all(float(d.light_intensity) == 0.0 for d in self.directionals)

    def surface_light(self, face_vertices):
        """Per-face light colour [B,F,3] for face_vertices [B,F,3,3] (normals do not depend on
        a translation of the mesh, so camera-space corners serve as well)."""
        B, nf = face_vertices.shape[0], face_vertices.shape[1]
        light = torch.zeros(B, nf, 3, dtype=torch.float32, device=face_vertices.device)
        light = self.ambient(light)
        if any(float(d.light_intensity) != 0.0 for d in self.directionals):
            normals = face_normals(face_vertices)
            for directional in self.directionals:
                light = directional(light, normals)
        return light

    def forward(self, mesh):
        if self.is_identity():
            return mesh                             # textures * 1: skip a full pass over the texels
        if self.light_mode == 'surface':
            light = self.surface_light(mesh.face_vertices)
            mesh.textures = mesh.textures * light[:, :, None, :]
        else:
            if any(float(d.light_intensity) != 0.0 for d in self.directionals):
                raise NotImplementedError('directional lighting on vertex textures is not supported')
            light = self.ambient(torch.zeros(mesh.batch_size, mesh.num_vertices, 3,
                                             dtype=torch.float32, device=mesh.device))
            mesh.textures = mesh.textures * light
        return mesh


# ---- camera transform ------------------------------------------------------------------
def _look_rotation(z_axis, up, dtype, device):
    up = torch.as_tensor(up, dtype=dtype, device=device)
    z_axis = F.normalize(z_axis, eps=1e-5, dim=-1)
    x_axis = F.normalize(torch.cross(up.expand_as(z_axis), z_axis, dim=-1), eps=1e-5, dim=-1)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=-1), eps=1e-5, dim=-1)
    return torch.stack((x_axis, y_axis, z_axis), dim=-2)          # rows


class Transform(nn.Module):
    """Camera transform on mesh vertices: 'look_at' (towards the origin) or 'look' (along a
    direction), then perspective or orthogonal projection."""

    def __init__(self, camera_mode='projection', P=None, dist_coeffs=None, orig_size=512,
                 perspective=True, viewing_angle=30, viewing_scale=1.0, eye=None,
                 camera_direction=(0, 0, 1)):
        super(Transform, self).__init__()
        if camera_mode not in ('look_at', 'look'):
            raise NotImplementedError("camera_mode %r is not supported (only 'look_at' and 'look')" % camera_mode)
        self.camera_mode = camera_mode
        self.perspective = perspective
        self.viewing_angle = viewing_angle
        self.viewing_scale = viewing_scale
        self.camera_direction = camera_direction
        if eye is None:
            eye = [0, 0, -(1. / math.tan(math.radians(viewing_angle)) + 1)]
        self.eye = eye

    def set_eyes(self, eyes):
        self.eye = eyes

    def transform_vertices(self, v):
        eye = torch.as_tensor(self.eye, dtype=v.dtype, device=v.device)
        if eye.dim() == 1:
            eye = eye[None]
        if self.camera_mode == 'look_at':
            z_axis = -eye                                             # at = origin
        else:
            z_axis = torch.as_tensor(self.camera_direction, dtype=v.dtype, device=v.device).expand_as(eye)
        r = _look_rotation(z_axis, (0.0, 1.0, 0.0), v.dtype, v.device)   # [1 or B,3,3]
        v = torch.matmul(v - eye[:, None, :], r.transpose(1, 2))
        if self.perspective:
            width = math.tan(math.radians(self.viewing_angle))
            z = v[:, :, 2:3]
            v = torch.cat((v[:, :, :2] / (z * width), z), dim=2)
        elif self.viewing_scale != 1.0:
            v = torch.cat((v[:, :, :2] * self.viewing_scale, v[:, :, 2:3]), dim=2)
        return v

    def forward(self, mesh):
        mesh.vertices = self.transform_vertices(mesh.vertices)
        return mesh


# ---- rasterizer ------------------------------------------------------------------------
class SoftRasterizer(nn.Module):
    def __init__(self, image_size=256, background_color=(0, 0, 0), near=1, far=100,
                 anti_aliasing=False, fill_back=False, eps=1e-3, sigma_val=1e-5,
                 dist_func='euclidean', dist_eps=1e-4, gamma_val=1e-4, aggr_func_rgb='softmax',
                 aggr_func_alpha='prod', texture_type='surface'):
        super(SoftRasterizer, self).__init__()
        if dist_func != 'euclidean':
            raise NotImplementedError("dist_func %r is not supported (only 'euclidean')" % dist_func)
        if aggr_func_rgb != 'softmax':
            raise NotImplementedError("aggr_func_rgb %r is not supported (only 'softmax')" % aggr_func_rgb)
        if aggr_func_alpha != 'prod':
            raise NotImplementedError("aggr_func_alpha %r is not supported (only 'prod')" % aggr_func_alpha)
        if texture_type not in ('surface', 'vertex'):
            raise ValueError('Texture type only support surface and vertex')
        self.image_size = image_size
        self.background_color = background_color
        self.near = near
        self.far = far
        self.anti_aliasing = anti_aliasing
        self.eps = eps
        self.fill_back = fill_back
        self.sigma_val = sigma_val
        self.dist_func = dist_func
        self.dist_eps = dist_eps
        self.gamma_val = gamma_val
        self.aggr_func_rgb = aggr_func_rgb
        self.aggr_func_alpha = aggr_func_alpha
        self.texture_type = texture_type

    def forward(self, mesh, mode=None):
        """RGBA [B,4,H,W]; mode 'silhouettes' evaluates coverage only (RGB = background)."""
        textures = None if mode == 'silhouettes' else mesh.face_textures
        return sv3d.soft_rasterize(mesh.face_vertices, textures, self.image_size,
                                   anti_aliasing=self.anti_aliasing, texture_type=self.texture_type,
                                   background_color=self.background_color, near=self.near,
                                   far=self.far, eps=self.eps, sigma_val=self.sigma_val,
                                   dist_eps=self.dist_eps, gamma_val=self.gamma_val,
                                   fill_back=self.fill_back)


class SoftRenderer(nn.Module):
    """`sr.SoftRenderer(...)(vertices, faces, textures=None) -> [B,4,H,W]`."""

    def __init__(self, image_size=256, background_color=(0, 0, 0), near=1, far=100,
                 anti_aliasing=False, fill_back=True, eps=1e-3, sigma_val=1e-5,
                 dist_func='euclidean', dist_eps=1e-4, gamma_val=1e-4, aggr_func_rgb='softmax',
                 aggr_func_alpha='prod', texture_type='surface', camera_mode='projection', P=None,
                 dist_coeffs=None, orig_size=512, perspective=True, viewing_angle=30,
                 viewing_scale=1.0, eye=None, camera_direction=(0, 0, 1), light_mode='surface',
                 light_intensity_ambient=0.5, light_color_ambient=(1, 1, 1),
                 light_intensity_directionals=0.5, light_color_directionals=(1, 1, 1),
                 light_directions=(0, 1, 0)):
        super(SoftRenderer, self).__init__()
        self.lighting = Lighting(light_mode, light_intensity_ambient, light_color_ambient,
                                 light_intensity_directionals, light_color_directionals,
                                 light_directions)
        self.transform = Transform(camera_mode, P, dist_coeffs, orig_size, perspective,
                                   viewing_angle, viewing_scale, eye, camera_direction)
        self.rasterizer = SoftRasterizer(image_size, background_color, near, far, anti_aliasing,
                                         fill_back, eps, sigma_val, dist_func, dist_eps, gamma_val,
                                         aggr_func_rgb, aggr_func_alpha, texture_type)

    def set_sigma(self, sigma):
        self.rasterizer.sigma_val = sigma

    def set_gamma(self, gamma):
        self.rasterizer.gamma_val = gamma

    def set_texture_mode(self, mode):
        if mode not in ('surface', 'vertex'):
            raise ValueError('Mode only support surface and vertex')
        self.lighting.light_mode = mode
        self.rasterizer.texture_type = mode

    def render_mesh(self, mesh, mode=None):
        self.set_texture_mode(mesh.texture_type)
        mesh = self.lighting(mesh)
        mesh = self.transform(mesh)
        return self.rasterizer(mesh, mode)

    def forward(self, vertices, faces, textures=None, mode=None, texture_type='surface'):
        mesh = Mesh(vertices, faces, textures=textures, texture_type=texture_type)
        return self.render_mesh(mesh, mode)
