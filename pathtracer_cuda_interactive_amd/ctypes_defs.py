"""ctypes mirrors of the PODs in include/pt_api.h and include/pt_host.h.

Kept free of any library loading so both the product bindings (host.py, device.py)
and the test-only oracle binding (tests/oracle_binding.py) can share them.
"""
import ctypes as C

c_float3 = C.c_float * 3

PT_OK = 0
PT_ERR_INVALID_ARG, PT_ERR_BAD_SCENE, PT_ERR_DEVICE, PT_ERR_NO_DEVICE = 1, 2, 3, 4
PT_ERR_IO, PT_ERR_PARSE, PT_ERR_UNSUPPORTED = 5, 6, 7
PT_SHAPE_SPHERE, PT_SHAPE_TRIANGLE = 0, 1
PT_MAT_DIFFUSE, PT_MAT_MIRROR, PT_MAT_PLASTIC, PT_MAT_PHONG = 0, 1, 2, 3
PT_LIGHT_POINT, PT_LIGHT_DIFFUSE_AREA = 0, 1
PT_TRAVERSAL_DEFAULT, PT_TRAVERSAL_EXACT, PT_TRAVERSAL_PRUNED = 0, 1, 2
PT_BVH_SORT_TOTAL, PT_BVH_SORT_REFERENCE = 0, 1
PT_RENDER_NEE = 1

STATUS_NAMES = {
    0: "PT_OK", 1: "PT_ERR_INVALID_ARG", 2: "PT_ERR_BAD_SCENE", 3: "PT_ERR_DEVICE", 4: "PT_ERR_NO_DEVICE",
    5: "PT_ERR_IO", 6: "PT_ERR_PARSE", 7: "PT_ERR_UNSUPPORTED",
}


class PtShape(C.Structure):
    _fields_ = [("type", C.c_int32), ("material_id", C.c_int32), ("area_light_id", C.c_int32),
                ("center", c_float3), ("radius", C.c_float), ("face_index", C.c_int32), ("mesh_index", C.c_int32)]


class PtMesh(C.Structure):
    _fields_ = [("material_id", C.c_int32), ("area_light_id", C.c_int32), ("num_vertices", C.c_int32),
                ("num_faces", C.c_int32), ("positions", C.POINTER(C.c_float)), ("indices", C.POINTER(C.c_int32)),
                ("normals", C.POINTER(C.c_float))]


class PtMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("reflectance", c_float3), ("eta", C.c_float), ("exponent", C.c_float)]


class PtLight(C.Structure):
    _fields_ = [("type", C.c_int32), ("shape_id", C.c_int32), ("radiance", c_float3), ("position", c_float3)]


class PtBvhNode(C.Structure):
    _fields_ = [("bmin", c_float3), ("bmax", c_float3), ("left", C.c_int32), ("right", C.c_int32), ("prim", C.c_int32)]


class PtSceneDesc(C.Structure):
    _fields_ = [("num_shapes", C.c_int32), ("shapes", C.POINTER(PtShape)),
                ("num_meshes", C.c_int32), ("meshes", C.POINTER(PtMesh)),
                ("num_materials", C.c_int32), ("materials", C.POINTER(PtMaterial)),
                ("num_lights", C.c_int32), ("lights", C.POINTER(PtLight)),
                ("num_nodes", C.c_int32), ("nodes", C.POINTER(PtBvhNode)),
                ("root", C.c_int32), ("background", c_float3)]


class PtRenderParams(C.Structure):
    _fields_ = [("cam_origin", c_float3), ("cam_top_left", c_float3), ("cam_horizontal", c_float3),
                ("cam_vertical", c_float3), ("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32),
                ("row_begin", C.c_int32), ("row_end", C.c_int32), ("row_stride", C.c_int32),
                ("seed", C.c_uint64), ("max_depth", C.c_int32), ("rr_depth", C.c_int32),
                ("sample_offset", C.c_int32), ("stream_stride", C.c_int32), ("traversal", C.c_int32),
                ("flags", C.c_int32)]

    def copy(self):
        out = PtRenderParams()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(PtRenderParams))
        return out

    def num_rows(self):
        rb, re = self.row_begin, self.row_end
        if rb == 0 and re == 0:
            re = self.height
        step = self.row_stride if self.row_stride > 1 else 1
        return len(range(rb, re, step))


class PtCounters(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("node_visits", C.c_uint64),
                ("leaf_tests", C.c_uint64), ("kernel_ms", C.c_double), ("resolve_ms", C.c_double)]


class PtCamera(C.Structure):
    _fields_ = [("lookfrom", c_float3), ("lookat", c_float3), ("up", c_float3), ("vfov", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32)]


class PtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status
