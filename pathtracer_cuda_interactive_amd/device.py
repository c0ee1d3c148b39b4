"""Device library bindings (libpt_hip.so): the C ABI of include/pt_api.h.

There is NO CPU fallback: if the HIP library is missing or no GPU is present the
calls raise (PT_ERR_NO_DEVICE) — the product never routes through the oracle.
"""
import ctypes as C
import os

import numpy as np

from . import _build
from .ctypes_defs import (PT_OK, PT_TRAVERSAL_DEFAULT, PtBvhNode, PtCounters, PtError, PtRenderParams, PtSceneDesc)

_lib = None

EXPORTS = [
    "pt_api_version", "pt_last_error", "pt_scene_create", "pt_scene_destroy", "pt_render", "pt_render_async",
    "pt_render_accumulate", "pt_get_counters", "pt_scene_set_option", "pt_scene_get_info", "pt_debug_math",
    "pt_debug_intersect", "pt_debug_math_host", "pt_bvh_build_device", "pt_bvh_build_sweep", "pt_get_frame_times", "pt_bvh_build_sweep_device",
]


def lib():
    global _lib
    if _lib is None:
        path = _build.HIP_LIB
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -m pathtracer_cuda_interactive_amd._build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = C.CDLL(path)
        vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.pt_api_version.restype = C.c_int
        L.pt_last_error.restype = C.c_char_p
        L.pt_scene_create.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(vp)]
        L.pt_scene_destroy.argtypes = [vp]
        L.pt_render.argtypes = [vp, C.POINTER(PtRenderParams), vp, C.c_int]
        L.pt_render_async.argtypes = [vp, C.POINTER(PtRenderParams), vp, vp]
        L.pt_render_accumulate.argtypes = [vp, C.POINTER(PtRenderParams), vp, vp]
        L.pt_get_counters.argtypes = [vp, C.POINTER(PtCounters)]
        L.pt_get_frame_times.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.pt_scene_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
        L.pt_scene_get_info.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
        L.pt_debug_math.argtypes = [C.c_int, fp, fp, fp, fp, C.c_int]
        L.pt_debug_intersect.argtypes = [vp, fp, C.c_int, C.c_int, fp, ip]
        L.pt_debug_math_host.argtypes = [C.c_int, fp, fp, fp, fp, C.c_int]
        L.pt_bvh_build_device.argtypes = [C.POINTER(PtSceneDesc), C.c_int, C.POINTER(PtBvhNode), ip, ip, C.POINTER(C.c_double)]
        L.pt_bvh_build_sweep.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(PtBvhNode), ip, ip, C.POINTER(C.c_double)]
        L.pt_bvh_build_sweep_device.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(PtBvhNode), ip, ip, C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _check(rc):
    if rc != PT_OK:
        raise PtError(rc, lib().pt_last_error().decode())


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class DeviceScene:
    """A scene resident on the current HIP device (pt_scene*)."""

    def __init__(self, desc):
        h = C.c_void_p()
        _check(lib().pt_scene_create(C.byref(desc), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().pt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        _check(lib().pt_scene_set_option(self._h, key.encode(), int(value)))

    def info(self, key):
        v = C.c_int64()
        _check(lib().pt_scene_get_info(self._h, key.encode(), C.byref(v)))
        return v.value

    def render(self, params, traversal=None):
        """Blocking render into a new host array [rows, W, 3] float32."""
        p = params.copy()
        if traversal is not None:
            p.traversal = traversal
        img = np.empty((p.num_rows(), p.width, 3), dtype=np.float32)
        _check(lib().pt_render(self._h, C.byref(p), img.ctypes.data_as(C.c_void_p), 0))
        return img

    def render_into(self, params, dev_ptr, stream=None, traversal=None):
        """Asynchronous render into device memory (e.g. a torch tensor's data_ptr()) on a HIP stream."""
        p = params.copy()
        if traversal is not None:
            p.traversal = traversal
        _check(lib().pt_render_async(self._h, C.byref(p), C.c_void_p(dev_ptr), C.c_void_p(stream or 0)))

    def accumulate_into(self, params, dev_ptr, stream=None):
        _check(lib().pt_render_accumulate(self._h, C.byref(params), C.c_void_p(dev_ptr), C.c_void_p(stream or 0)))

    def counters(self):
        c = PtCounters()
        _check(lib().pt_get_counters(self._h, C.byref(c)))
        return c

    def frame_times(self, max_frames):
        """(kernel_ms[], resolve_ms[]) of the last render calls, oldest first (option "timing_frames" sets how many are kept)."""
        k = (C.c_double * max(max_frames, 1))()
        r = (C.c_double * max(max_frames, 1))()
        n = C.c_int()
        _check(lib().pt_get_frame_times(self._h, int(max_frames), k, r, C.byref(n)))
        return np.array(k[: n.value]), np.array(r[: n.value])

    def intersect(self, rays, traversal=PT_TRAVERSAL_DEFAULT):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        tuv = np.zeros((rays.shape[0], 3), dtype=np.float32)
        prim = np.zeros(rays.shape[0], dtype=np.int32)
        _check(lib().pt_debug_intersect(self._h, _fp(rays), rays.shape[0], traversal, _fp(tuv),
                                        prim.ctypes.data_as(C.POINTER(C.c_int32))))
        return tuv, prim


def debug_math(op, x, y=None, host=False):
    """op 0: sincos, 1: powf, 2: pcg32 (stream/seed = bit patterns of x/y).  host=True runs the host build of pt_math.h."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = x if y is None else np.ascontiguousarray(y, dtype=np.float32)
    o0 = np.zeros_like(x)
    o1 = np.zeros_like(x)
    fn = lib().pt_debug_math_host if host else lib().pt_debug_math
    _check(fn(op, _fp(x), _fp(y), _fp(o0), _fp(o1), x.size))
    return o0, o1


def build_bvh_sweep(desc, on_device=False):
    """The library's internal-tree builder over the leaf boxes of `desc`'s tree: pt_bvh_build_sweep (host code, runs without a
    GPU) or, on_device=True, pt_bvh_build_sweep_device — the same tree byte for byte.  Returns (desc2, info) like build_bvh_device."""
    n_nodes = 2 * desc.num_shapes - 1
    nodes = np.zeros(max(n_nodes, 1), dtype=NODE_DTYPE)
    root, depth, ms = C.c_int32(), C.c_int32(), C.c_double()
    fn = lib().pt_bvh_build_sweep_device if on_device else lib().pt_bvh_build_sweep
    _check(fn(C.byref(desc), nodes.ctypes.data_as(C.POINTER(PtBvhNode)), C.byref(root), C.byref(depth),
                                    C.byref(ms)))
    d2 = PtSceneDesc()
    C.memmove(C.byref(d2), C.byref(desc), C.sizeof(PtSceneDesc))
    d2.nodes = nodes.ctypes.data_as(C.POINTER(PtBvhNode))
    d2.num_nodes = n_nodes
    d2.root = root.value
    d2._keep = (nodes, desc)
    return d2, {"root": root.value, "depth": depth.value, "build_ms": ms.value, "nodes": nodes}


PT_BVH_DEVICE_LBVH, PT_BVH_DEVICE_SAH = 0, 1
NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("left", "<i4"), ("right", "<i4"), ("prim", "<i4")])


def build_bvh_device(desc, method=PT_BVH_DEVICE_SAH):
    """Builds the BVH of `desc`'s primitives on the GPU (pt_bvh_build_device).  Returns (desc2, info): desc2 is a copy of
    `desc` that points at the new node array (kept alive by desc2), info = {"root", "depth", "build_ms", "nodes"}."""
    n_nodes = 2 * desc.num_shapes - 1
    nodes = np.zeros(max(n_nodes, 1), dtype=NODE_DTYPE)
    root, depth, ms = C.c_int32(), C.c_int32(), C.c_double()
    _check(lib().pt_bvh_build_device(C.byref(desc), int(method), nodes.ctypes.data_as(C.POINTER(PtBvhNode)), C.byref(root),
                                     C.byref(depth), C.byref(ms)))
    d2 = PtSceneDesc()
    C.memmove(C.byref(d2), C.byref(desc), C.sizeof(PtSceneDesc))
    d2.nodes = nodes.ctypes.data_as(C.POINTER(PtBvhNode))
    d2.num_nodes = n_nodes
    d2.root = root.value
    d2._keep = (nodes, desc)          # the node array, and whatever keeps the other arrays of `desc` alive
    return d2, {"root": root.value, "depth": depth.value, "build_ms": ms.value, "nodes": nodes}
