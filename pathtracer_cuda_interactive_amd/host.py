"""Host scene pipeline bindings (libpt_host.so): parse -> flatten -> BVH.

Mirrors the reference's host surface: `parse_scene` (parse_scene.cpp:862-877),
`Scene::Scene` (scene.cpp:11-153) and `compute_camera_ray_data` (camera.cuh:28-43).
"""
import ctypes as C
import os

import numpy as np

from . import _build
from .ctypes_defs import (PT_BVH_SORT_TOTAL, PT_OK, PtCamera, PtError, PtMaterial, PtRenderParams, PtSceneDesc,
                          c_float3)

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.build_host()          # no-op when the in-tree .so is newer than its sources
        L = C.CDLL(path)
        vp = C.c_void_p
        L.pt_host_last_error.restype = C.c_char_p
        L.pt_host_scene_new.argtypes = [C.POINTER(vp)]
        L.pt_host_scene_load_xml.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.pt_host_scene_load_pts.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.pt_host_scene_save_pts.argtypes = [vp, C.c_char_p]
        L.pt_host_scene_destroy.argtypes = [vp]
        L.pt_host_scene_set_camera.argtypes = [vp, C.POINTER(PtCamera)]
        L.pt_host_scene_get_camera.argtypes = [vp, C.POINTER(PtCamera)]
        L.pt_host_scene_set_background.argtypes = [vp, c_float3]
        L.pt_host_scene_add_material.argtypes = [vp, C.POINTER(PtMaterial)]
        L.pt_host_scene_add_point_light.argtypes = [vp, c_float3, c_float3]
        L.pt_host_scene_add_sphere.argtypes = [vp, c_float3, C.c_float, C.c_int, C.POINTER(C.c_float)]
        L.pt_host_scene_add_mesh.argtypes = [vp, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int32), C.c_int,
                                             C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float)]
        L.pt_host_scene_finalize.argtypes = [vp, C.c_int]
        L.pt_host_scene_get_desc.argtypes = [vp, C.POINTER(PtSceneDesc)]
        L.pt_host_scene_bvh_depth.argtypes = [vp]
        L.pt_host_camera_ray_data.argtypes = [C.POINTER(PtCamera), C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.pt_host_camera_ray_data.restype = None
        L.pt_host_default_params.argtypes = [C.POINTER(PtCamera), C.c_int, C.c_int, C.c_int, C.POINTER(PtRenderParams)]
        L.pt_host_default_params.restype = None
        L.pt_host_write_pfm.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int, C.c_int]
        L.pt_host_write_ppm.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int, C.c_int]
        _lib = L
    return _lib


def _check(rc):
    if rc != PT_OK:
        raise PtError(rc, lib().pt_host_last_error().decode())


def _f3(v):
    return c_float3(float(v[0]), float(v[1]), float(v[2]))


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class HostScene:
    """A parsed scene that can be flattened into the `pt_scene_desc` the device library consumes."""

    def __init__(self, handle=None):
        if handle is None:
            handle = C.c_void_p()
            _check(lib().pt_host_scene_new(C.byref(handle)))
        self._h = handle
        self._desc = None

    # ---- constructors -------------------------------------------------
    @classmethod
    def load(cls, path):
        """Load a Mitsuba-style .xml scene or a .pts parsed-scene container."""
        h = C.c_void_p()
        fn = lib().pt_host_scene_load_pts if str(path).endswith(".pts") else lib().pt_host_scene_load_xml
        _check(fn(str(path).encode(), C.byref(h)))
        return cls(h)

    def save_pts(self, path):
        _check(lib().pt_host_scene_save_pts(self._h, str(path).encode()))

    def close(self):
        if self._h:
            lib().pt_host_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- builder (ParsedScene equivalent) ------------------------------
    def set_camera(self, lookfrom, lookat, up, vfov, width, height, spp=16):
        cam = PtCamera(_f3(lookfrom), _f3(lookat), _f3(up), float(vfov), int(width), int(height), int(spp))
        _check(lib().pt_host_scene_set_camera(self._h, C.byref(cam)))

    @property
    def camera(self):
        cam = PtCamera()
        _check(lib().pt_host_scene_get_camera(self._h, C.byref(cam)))
        return cam

    def set_background(self, rgb):
        _check(lib().pt_host_scene_set_background(self._h, _f3(rgb)))

    def add_material(self, mtype, reflectance, eta=1.5, exponent=5.0):
        m = PtMaterial(int(mtype), _f3(reflectance), float(eta), float(exponent))
        r = lib().pt_host_scene_add_material(self._h, C.byref(m))
        if r < 0:
            _check(-r)
        return r

    def add_point_light(self, position, intensity):
        r = lib().pt_host_scene_add_point_light(self._h, _f3(position), _f3(intensity))
        if r < 0:
            _check(-r)
        return r

    def add_sphere(self, center, radius, material_id, radiance=None):
        rad = None if radiance is None else (C.c_float * 3)(*[float(x) for x in radiance])
        r = lib().pt_host_scene_add_sphere(self._h, _f3(center), float(radius), int(material_id), rad)
        if r < 0:
            _check(-r)
        return r

    def add_mesh(self, positions, indices, material_id, normals=None, radiance=None):
        P = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        I = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1, 3)
        N = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        if N is not None and N.shape != P.shape:
            raise ValueError("normals must match positions")
        rad = None if radiance is None else (C.c_float * 3)(*[float(x) for x in radiance])
        r = lib().pt_host_scene_add_mesh(self._h, _fptr(P), P.shape[0], I.ctypes.data_as(C.POINTER(C.c_int32)),
                                         I.shape[0], None if N is None else _fptr(N), int(material_id), rad)
        if r < 0:
            _check(-r)
        return r

    # ---- Scene::Scene ---------------------------------------------------
    def finalize(self, bvh_sort_mode=PT_BVH_SORT_TOTAL):
        _check(lib().pt_host_scene_finalize(self._h, int(bvh_sort_mode)))
        d = PtSceneDesc()
        _check(lib().pt_host_scene_get_desc(self._h, C.byref(d)))
        d._owner = self          # the desc holds raw pointers into the native scene: keep it alive as long as the desc
        self._desc = d
        return d

    @property
    def desc(self):
        if self._desc is None:
            self.finalize()
        return self._desc

    @property
    def bvh_depth(self):
        return lib().pt_host_scene_bvh_depth(self._h)

    def nodes_array(self):
        """BVH node pool as a structured numpy array (copy)."""
        d = self.desc
        dt = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("left", "<i4"), ("right", "<i4"), ("prim", "<i4")])
        buf = C.string_at(d.nodes, d.num_nodes * dt.itemsize)
        return np.frombuffer(buf, dtype=dt).copy()

    # ---- camera ---------------------------------------------------------
    def render_params(self, width=None, height=None, spp=None, seed=1984):
        """pt_render_params for this scene's camera (width/height/spp override the XML's: SURVEY F4)."""
        cam = self.camera
        w = int(width or cam.width)
        h = int(height or cam.height)
        s = int(spp or cam.spp)
        p = PtRenderParams()
        lib().pt_host_default_params(C.byref(cam), w, h, s, C.byref(p))
        p.seed = int(seed)
        return p


def camera_ray_data(cam, width, height):
    out = (C.c_float * 12)()
    lib().pt_host_camera_ray_data(C.byref(cam), int(width), int(height), out)
    return np.array(out, dtype=np.float32).reshape(4, 3)


def write_image(path, img):
    """Save a frame [H, W, 3] float32: .pfm (lossless linear) or .ppm (sqrt-gamma 8 bit, as the reference displays it)."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("expected an [H, W, 3] image")
    fn = lib().pt_host_write_pfm if str(path).endswith(".pfm") else lib().pt_host_write_ppm
    rc = fn(str(path).encode(), _fptr(img), img.shape[1], img.shape[0])
    if rc != PT_OK:
        raise PtError(rc, f"cannot write {path}")


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(v) for v in f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return data[::-1].astype(np.float32)
