"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

from pathtracer_cuda_interactive_amd.ctypes_defs import PT_OK, PtError, PtRenderParams, PtSceneDesc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libpt_oracle.so")

MATH_DET, MATH_LIBM = 0, 1
RNG_PER_SAMPLE, RNG_PER_PIXEL = 0, 1


class OracleOpts(C.Structure):
    _fields_ = [("math_mode", C.c_int32), ("rng_mode", C.c_int32), ("threads", C.c_int32), ("accumulate", C.c_int32)]


class OracleCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("paths", "segments", "inner_pops", "leaf_tri", "leaf_sphere", "valid_hits", "closer_hits", "closer_tri",
                 "rng_draws", "emit", "term_miss", "term_rr", "term_absorb", "term_maxdepth", "max_stack",
                 "stack_overflow", "shadow_rays", "nee_hits")] + [("seconds", C.c_double), ("threads_used", C.c_int32), ("pad", C.c_int32),
                                       ("trace", C.c_void_p), ("trace_len", C.c_uint64), ("trace_cap", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n not in ("pad", "trace", "trace_len", "trace_cap")}

    def bytes_per_segment(self):
        """Algorithmic bytes per segment of the REFERENCE layout:
        inner pop 60 B, triangle leaf 120 B, sphere leaf 32 B, +60 B per closer TRIANGLE hit (3 normals + 3 uvs), 40 B shade."""
        s = float(self.segments)
        return (60.0 * self.inner_pops + 120.0 * self.leaf_tri + 32.0 * self.leaf_sphere + 60.0 * self.closer_tri) / s + 40.0


_lib = None


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)
    return ORACLE_LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            build()
        L = C.CDLL(ORACLE_LIB)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.pt_oracle_render.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(PtRenderParams), C.POINTER(OracleOpts), fp,
                                       C.POINTER(OracleCounters)]
        L.pt_oracle_render_pixels.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(PtRenderParams), C.POINTER(OracleOpts),
                                              ip, C.c_int, fp, C.POINTER(OracleCounters)]
        L.pt_oracle_intersect.argtypes = [C.POINTER(PtSceneDesc), fp, C.c_int, C.c_int, fp, ip]
        L.pt_oracle_trace_pixels.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(PtRenderParams), ip, C.c_int, C.c_char_p,
                                             C.c_uint64, C.POINTER(C.c_uint64)]
        L.pt_oracle_math.argtypes = [C.c_int, C.c_int, fp, fp, fp, fp, C.c_int]
        L.pt_oracle_pcg.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint32), fp, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def _chk(rc, what):
    if rc != PT_OK:
        raise PtError(rc, what)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def render(desc, params, math_mode=MATH_DET, rng_mode=RNG_PER_SAMPLE, threads=0, accumulate=False):
    """Returns (image[rows,W,3] float32, OracleCounters)."""
    opts = OracleOpts(math_mode, rng_mode, threads if threads > 0 else usable_cores(), 1 if accumulate else 0)
    img = np.zeros((params.num_rows(), params.width, 3), dtype=np.float32)
    cnt = OracleCounters()
    _chk(lib().pt_oracle_render(C.byref(desc), C.byref(params), C.byref(opts), _fp(img), C.byref(cnt)), "oracle render")
    return img, cnt


def render_pixels(desc, params, xy, math_mode=MATH_DET, rng_mode=RNG_PER_SAMPLE, threads=0):
    xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
    out = np.zeros((xy.shape[0], 3), dtype=np.float32)
    opts = OracleOpts(math_mode, rng_mode, threads if threads > 0 else usable_cores(), 0)
    cnt = OracleCounters()
    _chk(lib().pt_oracle_render_pixels(C.byref(desc), C.byref(params), C.byref(opts),
                                       xy.ctypes.data_as(C.POINTER(C.c_int32)), xy.shape[0], _fp(out), C.byref(cnt)),
         "oracle render_pixels")
    return out, cnt


def intersect(desc, rays, math_mode=MATH_DET):
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
    tuv = np.zeros((rays.shape[0], 3), dtype=np.float32)
    prim = np.zeros(rays.shape[0], dtype=np.int32)
    _chk(lib().pt_oracle_intersect(C.byref(desc), _fp(rays), rays.shape[0], math_mode, _fp(tuv),
                                   prim.ctypes.data_as(C.POINTER(C.c_int32))), "oracle intersect")
    return tuv, prim


def intersect_work(desc, rays, math_mode=MATH_DET):
    """Per ray: (inner pops, leaves reached, order-independent hash of the primitive ids of those leaves)."""
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
    n = rays.shape[0]
    inner = np.zeros(n, dtype=np.uint32)
    leaf = np.zeros(n, dtype=np.uint32)
    leaf_set = np.zeros(n, dtype=np.uint64)
    lib().pt_oracle_intersect_work.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    _chk(lib().pt_oracle_intersect_work(C.byref(desc), _fp(rays), n, math_mode, inner.ctypes.data, leaf.ctypes.data,
                                        leaf_set.ctypes.data), "oracle intersect_work")
    return inner, leaf, leaf_set


def sincos(x, math_mode=MATH_DET):
    x = np.ascontiguousarray(x, dtype=np.float32)
    s = np.zeros_like(x)
    c = np.zeros_like(x)
    _chk(lib().pt_oracle_math(0, math_mode, _fp(x), _fp(x), _fp(s), _fp(c), x.size), "oracle math")
    return s, c


def powf(x, y, math_mode=MATH_DET):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    o = np.zeros_like(x)
    _chk(lib().pt_oracle_math(1, math_mode, _fp(x), _fp(y), _fp(o), _fp(o), x.size), "oracle math")
    return o


def pcg(stream, seed, n):
    u = np.zeros(n, dtype=np.uint32)
    f = np.zeros(n, dtype=np.float32)
    si = (C.c_uint64 * 2)()
    _chk(lib().pt_oracle_pcg(int(stream), int(seed), n, u.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(f), si), "pcg")
    return u, f, (int(si[0]), int(si[1]))


def trace_pixels(desc, params, xy, cap=1 << 26):
    """Event string (I inner visit, L leaf test, S shaded hit, M miss, E path end) of every path of the listed pixels."""
    xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
    buf = C.create_string_buffer(cap)
    n = C.c_uint64()
    _chk(lib().pt_oracle_trace_pixels(C.byref(desc), C.byref(params), xy.ctypes.data_as(C.POINTER(C.c_int32)), xy.shape[0],
                                      buf, cap, C.byref(n)), "oracle trace")
    return buf.raw[: n.value].decode()
