"""ctypes binding of libsv3d_hip.so (C ABI declared in include/sv3d_hip.h).

The library is the product: there is no CPU or pure-torch fallback.  Loading fails
loudly when the shared object is missing, and every entry point raises on a non-zero
status.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsv3d_hip.so")

SV3D_TEX_NONE, SV3D_TEX_SURFACE, SV3D_TEX_VERTEX, SV3D_TEX_VERTEX8 = 0, 1, 2, 3
SV3D_BWD_AUTO, SV3D_BWD_BOXES, SV3D_BWD_INTERVALS = 0, 1, 2

EXPORTS = (
    "sv3d_strerror", "sv3d_abi_version", "sv3d_raster_sizes", "sv3d_raster_forward",
    "sv3d_raster_backward", "sv3d_raster_forward_timed", "sv3d_raster_backward_timed",
    "sv3d_raster_forward_timed_n", "sv3d_raster_backward_timed_n", "sv3d_raster_forward_kernel_name",
    "sv3d_chamfer_forward", "sv3d_chamfer_backward",
    "sv3d_face_gather", "sv3d_face_scatter",
    "sv3d_project_gather", "sv3d_project_scatter", "sv3d_render_forward",
    "sv3d_face_gather_n", "sv3d_face_scatter_n",
    "sv3d_laplacian_loss", "sv3d_flatten_loss", "sv3d_mesh_spmv",
    "sv3d_texture_sample", "sv3d_texture_sample_backward", "sv3d_distance_transform",
    "sv3d_distance_transform_workspace", "sv3d_distance_transform_static",
)


class RasterParams(ctypes.Structure):
    """Mirror of `sv3d_raster_params`."""
    _fields_ = [
        ("batch", ctypes.c_int32), ("num_faces", ctypes.c_int32),
        ("raster_size", ctypes.c_int32), ("aa", ctypes.c_int32),
        ("texture_mode", ctypes.c_int32), ("texture_res", ctypes.c_int32),
        ("double_side", ctypes.c_int32), ("texture_batch", ctypes.c_int32),
        ("background", ctypes.c_float * 3),
        ("near_plane", ctypes.c_float), ("far_plane", ctypes.c_float), ("eps", ctypes.c_float),
        ("sigma", ctypes.c_float), ("dist_eps", ctypes.c_float), ("gamma", ctypes.c_float),
        ("alpha_only", ctypes.c_int32), ("backward_enumeration", ctypes.c_int32),
    ]


class MeshInput(ctypes.Structure):
    """Mirror of `sv3d_mesh_input`."""
    _fields_ = [
        ("vertices", ctypes.c_void_p), ("cams", ctypes.c_void_p), ("faces", ctypes.c_void_p),
        ("num_vertices", ctypes.c_int32), ("faces_batch", ctypes.c_int32),
        ("offset_z", ctypes.c_float), ("scale", ctypes.c_float * 3), ("shift", ctypes.c_float * 3),
    ]


class Sv3dError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library; raises if it has not been built (`make -C .../csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Sv3dError(
                "libsv3d_hip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C single-view-3d-reconstruction_amd/csrc`; there is no fallback path"
                % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        vp, i32, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_size_t
        pp = ctypes.POINTER(RasterParams)
        L.sv3d_strerror.restype = ctypes.c_char_p
        L.sv3d_strerror.argtypes = [ctypes.c_int]
        L.sv3d_abi_version.restype = ctypes.c_int
        L.sv3d_abi_version.argtypes = []
        L.sv3d_raster_sizes.argtypes = [pp, ctypes.POINTER(sz), ctypes.POINTER(sz), ctypes.POINTER(sz)]
        L.sv3d_raster_forward.argtypes = [pp, vp, vp, vp, vp, vp, vp]
        L.sv3d_raster_backward.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        L.sv3d_raster_forward_timed.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.sv3d_raster_backward_timed.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp]
        L.sv3d_raster_forward_timed_n.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, vp, i32]
        L.sv3d_raster_backward_timed_n.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, i32]
        L.sv3d_chamfer_forward.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]
        L.sv3d_chamfer_backward.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
        f3 = ctypes.POINTER(ctypes.c_float)
        L.sv3d_face_gather.argtypes = [vp, vp, i32, i32, i32, i32, f3, f3, vp, vp]
        L.sv3d_face_scatter.argtypes = [vp, vp, i32, i32, i32, i32, f3, vp, vp]
        L.sv3d_face_gather_n.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
        L.sv3d_face_scatter_n.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
        L.sv3d_laplacian_loss.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
        L.sv3d_flatten_loss.argtypes = [vp, vp, vp, vp, i32, i32, i32, ctypes.c_float, vp, vp, vp]
        L.sv3d_mesh_spmv.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
        L.sv3d_texture_sample.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
        L.sv3d_texture_sample_backward.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
        L.sv3d_distance_transform.argtypes = [vp, i32, i32, i32, i32, ctypes.c_float, vp, vp, ctypes.c_size_t, vp]
        L.sv3d_distance_transform_workspace.argtypes = [i32, i32, i32, ctypes.POINTER(ctypes.c_size_t)]
        L.sv3d_distance_transform_static.argtypes = [vp, i32, i32, i32, i32, ctypes.c_float, vp, vp]
        mp = ctypes.POINTER(MeshInput)
        L.sv3d_project_gather.argtypes = [mp, i32, i32, vp, vp]
        L.sv3d_project_scatter.argtypes = [mp, i32, i32, vp, vp, vp, vp]
        L.sv3d_render_forward.argtypes = [pp, mp, vp, vp, vp, vp, vp, vp]
        L.sv3d_raster_forward_kernel_name.argtypes = [pp, i32]
        L.sv3d_raster_forward_kernel_name.restype = ctypes.c_char_p
        for name in EXPORTS:
            if name not in ("sv3d_strerror", "sv3d_raster_forward_kernel_name"):
                getattr(L, name).restype = ctypes.c_int
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        raise Sv3dError("%s failed: %s (status %d)" % (what, lib().sv3d_strerror(status).decode(), status))


def make_params(batch, num_faces, raster_size, aa, texture_mode, texture_res, double_side,
                background, near, far, eps, sigma, dist_eps, gamma, alpha_only=False,
                backward_enumeration=SV3D_BWD_AUTO):
    p = RasterParams()
    p.batch, p.num_faces, p.raster_size, p.aa = int(batch), int(num_faces), int(raster_size), int(aa)
    p.texture_mode, p.texture_res = int(texture_mode), int(texture_res)
    p.double_side, p.texture_batch = int(bool(double_side)), 0
    p.background[0], p.background[1], p.background[2] = (float(c) for c in background)
    p.near_plane, p.far_plane, p.eps = float(near), float(far), float(eps)
    p.sigma, p.dist_eps, p.gamma = float(sigma), float(dist_eps), float(gamma)
    p.alpha_only = int(bool(alpha_only))
    p.backward_enumeration = int(backward_enumeration)
    return p


def raster_sizes(p):
    a, b, c = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    check(lib().sv3d_raster_sizes(ctypes.byref(p), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)),
          "sv3d_raster_sizes")
    return a.value, b.value, c.value


def make_mesh_input(vertices, cams, faces, offset_z, scale, shift):
    """`sv3d_mesh_input` over three CUDA(HIP) tensors (kept alive by the caller for the call)."""
    m = MeshInput()
    m.vertices, m.cams, m.faces = vertices.data_ptr(), cams.data_ptr(), faces.data_ptr()
    m.num_vertices, m.faces_batch = int(vertices.shape[1]), int(faces.shape[0])
    m.offset_z = float(offset_z)
    for i in range(3):
        m.scale[i] = float(scale[i])
        m.shift[i] = float(shift[i])
    return m


def float3(values):
    return (ctypes.c_float * 3)(*[float(v) for v in values])
