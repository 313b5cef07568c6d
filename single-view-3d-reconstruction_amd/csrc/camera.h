// Device-side camera stage shared by the mesh glue kernels and the rasterizer's fused setup:
// the weak-perspective camera of nnutils/geom_utils.py:63-80 (quat_rotate :143-161 with an
// un-normalised quaternion, so the rotation carries |q|^2) followed by the per-axis scale and
// shift that express the y flip of nnutils/softras.py:60 and the renderer's look_at z shift.
#ifndef SV3D_CAMERA_H
#define SV3D_CAMERA_H

#include <hip/hip_runtime.h>

namespace sv3d_dev {

struct Affine {
    float s0, s1, s2, t0, t1, t2;
};

// P = s * M X + (tx, ty, tz);  M = (w^2 - |u|^2) I + 2 u u^T + 2 w [u]x,  q = (w, u)
struct Camera {
    float m00, m01, m02, m10, m11, m12, m20, m21, m22;
    float s, tx, ty, tz;
    float w, ux, uy, uz;
};

__device__ inline Camera load_camera(const float *__restrict__ cam, float offset_z)
{
    Camera c;
    c.s = cam[0]; c.tx = cam[1]; c.ty = cam[2]; c.tz = offset_z;
    const float w = cam[3], x = cam[4], y = cam[5], z = cam[6];
    c.w = w; c.ux = x; c.uy = y; c.uz = z;
    const float ww = w * w, xx = x * x, yy = y * y, zz = z * z;
    c.m00 = ww + xx - yy - zz; c.m01 = 2.f * (x * y - w * z); c.m02 = 2.f * (x * z + w * y);
    c.m10 = 2.f * (x * y + w * z); c.m11 = ww - xx + yy - zz; c.m12 = 2.f * (y * z - w * x);
    c.m20 = 2.f * (x * z - w * y); c.m21 = 2.f * (y * z + w * x); c.m22 = ww - xx - yy + zz;
    return c;
}

__device__ inline void camera_rotate(const Camera &c, float X, float Y, float Z, float &rx, float &ry,
                                     float &rz)
{
    rx = c.m00 * X + c.m01 * Y + c.m02 * Z;
    ry = c.m10 * X + c.m11 * Y + c.m12 * Z;
    rz = c.m20 * X + c.m21 * Y + c.m22 * Z;
}

// out = a.s * (s M X + t) + a.t
__device__ inline void camera_project(const Camera &c, const Affine &a, float X, float Y, float Z,
                                      float &ox, float &oy, float &oz)
{
    float rx, ry, rz;
    camera_rotate(c, X, Y, Z, rx, ry, rz);
    ox = fmaf(c.s, rx, c.tx) * a.s0 + a.t0;
    oy = fmaf(c.s, ry, c.ty) * a.s1 + a.t1;
    oz = fmaf(c.s, rz, c.tz) * a.s2 + a.t2;
}

}  // namespace sv3d_dev
#endif
