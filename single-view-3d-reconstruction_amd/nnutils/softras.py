"""Drop-in for the reference's nnutils/softras.py: `SoftNeuralRenderer`.

Same constructor, attributes and methods (softras.py:25-77): weak-perspective projection
with offset_z 5, y flip, then one of three renders -- silhouette (textures None), lit RGB
from surface textures, or unlit per-vertex colours for the part-segmentation maps.  The
renderer underneath is the `soft_renderer` drop-in over the sv3d HIP kernels.
"""
import torch

import soft_renderer as sr
import sv3d

from . import geom_utils


def convert_as(src, trg):
    return src.to(trg.device).type_as(trg)


class SoftNeuralRenderer(torch.nn.Module):
    """forward(vertices [B,V,3], faces [B,F,3], cams [B,7], textures=None) ->
    masks [B,H,W] when textures is None, else RGB [B,3,H,W].

    `textures` may hold fewer sets than there are images: TB sets with B a multiple of TB, image b
    using set b % TB -- what `torch.cat([textures] * num_cams)` (main.py:415) builds, without the
    copies (fused camera stage only)."""

    def __init__(self, img_size=256, texture_type='surface'):
        super(SoftNeuralRenderer, self).__init__()
        # constants of softras.py:33-37
        self.renderer = sr.SoftRenderer(image_size=img_size, camera_mode='look_at',
                                        aggr_func_rgb='softmax', anti_aliasing=True,
                                        sigma_val=1e-5, gamma_val=1e-4, dist_eps=1e-10,
                                        perspective=False, light_intensity_ambient=0.8,
                                        eye=[0, 0, -2.732], texture_type=texture_type,
                                        light_mode=texture_type)
        self.proj_fn = geom_utils.orthographic_proj_withz
        self.offset_z = 5.
        self.part_seg = texture_type == 'vertex'

    def ambient_light_only(self):
        """softras.py:43-45: ambient 1, no directional light (textures pass unchanged)."""
        self.renderer.lighting = sr.Lighting(intensity_ambient=1., intensity_directionals=0.)

    def set_bgcolor(self, color):
        self.renderer.rasterizer.background_color = color

    def set_light_dir(self, direction):
        self.renderer.lighting.directionals[0].light_direction = direction

    def project_points(self, verts, cams):
        """softras.py:53-55: xy of the projection, no flip, no depth offset."""
        return self.proj_fn(verts, cams)[:, :, :2]

    def _camera_stage(self):
        """(scale, shift) that express the y flip and the renderer's camera transform when
        that transform is a pure depth shift (orthographic look_at from an eye on the -z axis,
        as configured above); None for any other transform."""
        t = self.renderer.transform
        eye = t.eye
        if self.proj_fn is not geom_utils.orthographic_proj_withz or t.camera_mode != 'look_at' \
                or t.perspective or t.viewing_scale != 1.0 or torch.is_tensor(eye):
            return None
        eye = [float(e) for e in eye]
        if len(eye) != 3 or eye[0] != 0.0 or eye[1] != 0.0 or not eye[2] < 0.0:
            return None
        return (1.0, -1.0, 1.0), (0.0, 0.0, -eye[2])

    def forward(self, vertices, faces, cams, textures=None, part_seg=False):
        stage = self._camera_stage()
        if stage is not None:
            return self._forward_fused(vertices, faces, cams, textures, *stage)
        return self._forward_staged(vertices, faces, cams, textures)

    def _forward_fused(self, vertices, faces, cams, textures, scale, shift):
        """The renderer call with camera, flip, depth shift, gather and face setup inside the
        HIP ops (one launch) instead of some twenty elementwise launches."""
        if textures is None:
            return self._rgba_fused(vertices, faces, cams, None, scale, shift, alpha_only=True)
        return self._rgba_fused(vertices, faces, cams, textures, scale, shift)[:, :3]

    def _packed(self, r, scale, shift, mode, res, alpha_only):
        """The operators' packed configuration for the rasterizer's current settings; packed again
        only when a setting changed (the mutators above, or a caller editing the rasterizer)."""
        from sv3d import ops
        key = (r.image_size, r.anti_aliasing, mode, res, r.fill_back, alpha_only, tuple(r.background_color), r.near,
               r.far, r.eps, r.sigma_val, r.dist_eps, r.gamma_val, self.offset_z, scale, shift, ops._backward_enumeration)
        hit = self.__dict__.get('_packed_cfg')
        if hit is None or hit[0] != key:
            icfg, fcfg = ops.raster_config(r.image_size, r.anti_aliasing, mode, res, r.fill_back, alpha_only,
                                           r.background_color, r.near, r.far, r.eps, r.sigma_val, r.dist_eps, r.gamma_val)
            hit = (key, ops._mesh_cfg(self.offset_z, scale, shift), icfg, fcfg)
            self.__dict__['_packed_cfg'] = hit
        return hit[1], hit[2], hit[3]

    def _rgba_fused(self, vertices, faces, cams, textures, scale, shift, alpha_only=False):
        r = self.renderer.rasterizer
        if textures is None:
            from sv3d import _lib, ops
            self.__dict__['mask_only'] = True
            mesh_cfg, icfg, fcfg = self._packed(r, scale, shift, _lib.SV3D_TEX_NONE, 1, alpha_only)
            return ops.render_packed(vertices, cams, faces, None, mesh_cfg, icfg, fcfg)
        kw = dict(offset_z=self.offset_z, scale=scale, shift=shift, anti_aliasing=r.anti_aliasing,
                  background_color=r.background_color, near=r.near, far=r.far, eps=r.eps,
                  sigma_val=r.sigma_val, dist_eps=r.dist_eps, gamma_val=r.gamma_val, fill_back=r.fill_back)
        self.mask_only = False
        if self.part_seg:
            self.renderer.set_texture_mode('vertex')
            colours = sv3d.face_gather(textures, faces if faces.dtype == torch.int32 else faces.to(torch.int32))
            return sv3d.render(vertices, cams, faces, colours, r.image_size, texture_type='vertex', **kw)
        self.renderer.set_texture_mode('surface')
        lighting = self.renderer.lighting
        if lighting.is_identity():
            return sv3d.render(vertices, cams, faces, textures, r.image_size, **kw)
        fv = sv3d.project_gather(vertices, cams, faces, self.offset_z, scale, shift)
        light = lighting.surface_light(fv)[:, :, None, :]
        if textures.shape[0] != fv.shape[0]:       # one texture set for several cameras
            sets = textures.shape[0]
            if sets < 1 or fv.shape[0] % sets != 0:
                raise ValueError("%d texture sets do not divide %d images" % (sets, fv.shape[0]))
            lit = (textures[None] * light.view((-1, sets) + tuple(light.shape[1:]))).flatten(0, 1)
        else:
            lit = textures * light
        kw.pop('offset_z'), kw.pop('scale'), kw.pop('shift')
        return sv3d.soft_rasterize(fv, lit, r.image_size, **kw)

    # ---- renders that share one rasterizer pass ------------------------------------------
    # The reference's training step renders the same (vertices, faces, cams) ten times per
    # iteration (main.py:417-426): a silhouette, an RGB image, and Np part maps one at a time,
    # each part map copied into three identical colour channels of which one is kept.  The
    # silhouette is the alpha channel every colour pass computes anyway, and the colour channels
    # of a vertex-colour pass are independent, so the same images come out of 1 + ceil(Np / 8)
    # passes (the kernels' eight-channel vertex mode).
    def forward_with_mask(self, vertices, faces, cams, textures):
        """(forward(vertices, faces, cams, textures), forward(vertices, faces, cams)) from one
        pass: RGB [B,3,H,W] and the silhouette [B,H,W]."""
        stage = self._camera_stage()
        if stage is None:
            return self.forward(vertices, faces, cams, textures), self.forward(vertices, faces, cams)
        rgba = self._rgba_fused(vertices, faces, cams, textures, *stage)
        return rgba[:, :3], rgba[:, 3]

    def render_parts(self, vertices, faces, cams, part_maps):
        """part_maps [B,V,Np] -> [B,Np,H,W], equal to stacking
        forward(vertices, faces, cams, part_maps[:, :, k:k+1].repeat(1, 1, 3))[:, 0:1] over k
        (main.py:422-426).  Eight maps per rasterizer pass (the kernels' eight-channel vertex
        mode) when the camera stage is the fused one, else three.  Needs texture_type='vertex'."""
        if not self.part_seg:
            raise ValueError("render_parts needs a renderer built with texture_type='vertex'")
        n = part_maps.shape[2]
        stage = self._camera_stage()
        out = []
        if stage is not None:
            r = self.renderer.rasterizer
            self.mask_only = False
            self.renderer.set_texture_mode('vertex')
            for k in range(0, n, 8):
                group = part_maps[:, :, k:k + 8]
                if group.shape[2] < 8:
                    group = torch.nn.functional.pad(group, (0, 8 - group.shape[2]))
                colours = sv3d.face_gather_n(group, faces)
                img = sv3d.render(vertices, cams, faces, colours, r.image_size, texture_type='vertex8',
                                  offset_z=self.offset_z, scale=stage[0], shift=stage[1],
                                  anti_aliasing=r.anti_aliasing, background_color=r.background_color,
                                  near=r.near, far=r.far, eps=r.eps, sigma_val=r.sigma_val,
                                  dist_eps=r.dist_eps, gamma_val=r.gamma_val, fill_back=r.fill_back)
                out.append(img[:, :min(8, n - k)])
            return out[0] if len(out) == 1 else torch.cat(out, dim=1)
        for k in range(0, n, 3):
            group = part_maps[:, :, k:k + 3]
            if group.shape[2] < 3:
                group = torch.nn.functional.pad(group, (0, 3 - group.shape[2]))
            out.append(self.forward(vertices, faces, cams, group)[:, :min(3, n - k)])
        return torch.cat(out, dim=1)

    def _forward_staged(self, vertices, faces, cams, textures):
        verts = self.proj_fn(vertices, cams, offset_z=self.offset_z)
        vs = verts * verts.new_tensor([1.0, -1.0, 1.0])               # image y points down
        if textures is None:
            self.mask_only = True
            # coverage only: the reference renders RGBA and keeps [:, -1] (softras.py:64)
            mesh = sr.Mesh(vs, faces)
            mesh = self.renderer.transform(mesh)
            return self.renderer.rasterizer(mesh, 'silhouettes')[:, -1]
        self.mask_only = False
        if self.part_seg:
            # vertex colours, no lighting (softras.py:69-75)
            mesh = sr.Mesh(vs, faces, textures=textures, texture_type='vertex')
            self.renderer.set_texture_mode(mesh.texture_type)
            mesh = self.renderer.transform(mesh)
            return self.renderer.rasterizer(mesh, None)[:, :3]
        return self.renderer(vs, faces, textures)[:, :3]
