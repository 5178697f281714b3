"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg.
The product package (raytracer_amd/) never imports this module.  PARITY UNPINNED (see oracle.h).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

PRIM_DT = np.dtype([("type", "<u4"), ("material", "<u4"), ("v", "<f8", (9,))])
MAT_DT = np.dtype([("type", "<u4"), ("pad", "<u4"), ("albedo", "<f8", (3,)), ("param", "<f8")])
NODE_DT = np.dtype([("min", "<f8", (3,)), ("max", "<f8", (3,)), ("is_leaf", "<u4"), ("left", "<u4"), ("right", "<u4"),
                    ("first", "<u4"), ("count", "<u4"), ("pad", "<u4")])
ALGOS = {"naive": 0, "bvh": 1, "grid": 2}
BVH_MODES = {"sah-position": 0, "sah-plane": 1, "naive": 2}


class CameraIn(C.Structure):
    _fields_ = [("aspect_ratio", C.c_double), ("image_width", C.c_uint32), ("pad", C.c_uint32), ("vfov", C.c_double),
                ("look_from", C.c_double * 3), ("look_at", C.c_double * 3), ("v_up", C.c_double * 3),
                ("defocus_angle", C.c_double), ("focus_dist", C.c_double)]


class CameraOut(C.Structure):
    _fields_ = [("image_width", C.c_uint32), ("image_height", C.c_uint32), ("center", C.c_double * 3),
                ("pixel00", C.c_double * 3), ("du", C.c_double * 3), ("dv", C.c_double * 3), ("ddu", C.c_double * 3),
                ("ddv", C.c_double * 3), ("defocus_angle", C.c_double)]


class Counters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("primary_rays", "scatter_rays", "intersection_checks", "traversal_steps",
                                          "overlapping_aabb", "gridbox_intersection_checks", "box_tests", "prim_tests")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} missing: run `make -C oracle`")
        L = C.CDLL(path)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_build_bvh.argtypes = [C.c_void_p, C.c_int]
        L.orc_build_bvh_sweep.argtypes = [C.c_void_p]
        L.orc_set_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_bvh_node_count.argtypes = [C.c_void_p]; L.orc_bvh_node_count.restype = C.c_uint32
        L.orc_get_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_build_grid.argtypes = [C.c_void_p, C.c_double]
        L.orc_grid_cell_count.argtypes = [C.c_void_p]; L.orc_grid_cell_count.restype = C.c_uint64
        L.orc_grid_list_len.argtypes = [C.c_void_p]; L.orc_grid_list_len.restype = C.c_uint64
        L.orc_get_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_set_draw_boxes.argtypes = [C.c_void_p, C.c_int]; L.orc_set_draw_boxes.restype = None
        L.orc_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Counters)]
        L.orc_camera_init.argtypes = [C.POINTER(CameraIn), C.POINTER(CameraOut)]
        L.orc_generate.argtypes = [C.POINTER(CameraOut), C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(CameraOut), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                 C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(Counters), C.c_int]
        L.orc_resolve.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
        L.orc_rng_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]; L.orc_rng_draw.restype = C.c_double
        _lib = L
    return _lib


class OracleScene:
    def __init__(self, prims, centroids, mats):
        self.prims = np.ascontiguousarray(prims, dtype=PRIM_DT)
        self.centroids = np.ascontiguousarray(centroids, dtype=np.float64).reshape(-1, 3)
        self.mats = np.ascontiguousarray(mats, dtype=MAT_DT)
        self._h = lib().orc_scene_create(self.prims.ctypes.data, self.centroids.ctypes.data, len(self.prims), self.mats.ctypes.data, len(self.mats))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_scene_destroy(self._h); self._h = None

    def build_bvh(self, mode="sah-position", sweep=False):
        """sweep=True (sah-position only): the O(n log n)-per-node statement of the same build, for big meshes."""
        if sweep and mode != "sah-position":
            raise ValueError("the sweep builder restates BvhSahPosition only")
        rc = lib().orc_build_bvh_sweep(self._h) if sweep else lib().orc_build_bvh(self._h, BVH_MODES[mode])
        if rc: raise RuntimeError(f"orc_build_bvh {rc}")
        return self

    def set_bvh(self, nodes, order):
        nodes = np.ascontiguousarray(nodes, dtype=NODE_DT); order = np.ascontiguousarray(order, dtype=np.uint32)
        rc = lib().orc_set_bvh(self._h, nodes.ctypes.data, len(nodes), order.ctypes.data, len(order))
        if rc: raise RuntimeError(f"orc_set_bvh {rc}")
        return self

    def get_bvh(self):
        n = lib().orc_bvh_node_count(self._h)
        nodes = np.zeros(n, NODE_DT); order = np.zeros(len(self.prims), np.uint32)
        rc = lib().orc_get_bvh(self._h, nodes.ctypes.data, order.ctypes.data)
        if rc: raise RuntimeError(f"orc_get_bvh {rc}")
        return nodes, order

    def build_grid(self, size):
        rc = lib().orc_build_grid(self._h, float(size))
        if rc: raise RuntimeError(f"orc_build_grid {rc}")
        return self

    def get_grid(self):
        nc = lib().orc_grid_cell_count(self._h); ll = lib().orc_grid_list_len(self._h)
        aabb = np.zeros((nc, 6)); off = np.zeros(nc + 1, np.uint64); pl = np.zeros(max(ll, 1), np.uint32)
        rc = lib().orc_get_grid(self._h, aabb.ctypes.data, off.ctypes.data, pl.ctypes.data)
        if rc: raise RuntimeError(f"orc_get_grid {rc}")
        return aabb, off, pl[:ll]

    def set_draw_boxes(self, on):
        lib().orc_set_draw_boxes(self._h, 1 if on else 0)
        return self

    def trace(self, algorithm, orig, direction, tmin=0.001, tmax=float("inf")):
        orig = np.ascontiguousarray(orig, dtype=np.float64).reshape(-1, 3)
        direction = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
        n = len(orig)
        prim = np.empty(n, np.int32); t = np.empty(n); tp = np.empty(n); hit = np.empty(n, np.uint8)
        c = Counters()
        rc = lib().orc_trace(self._h, ALGOS[algorithm], orig.ctypes.data, direction.ctypes.data, n, tmin, tmax,
                             prim.ctypes.data, t.ctypes.data, tp.ctypes.data, hit.ctypes.data, C.byref(c))
        if rc: raise RuntimeError(f"orc_trace {rc}")
        return {"prim": prim, "t": t, "tp": tp, "hit": (hit & 1).astype(bool), "edge": (hit & 2).astype(bool), "stats": c.as_dict()}

    def render(self, cam, algorithm, s_begin, s_count, max_depth, seed=1, rows=None, nthreads=1):
        H, W = cam.image_height, cam.image_width
        out = np.zeros((H, W, 3))
        c = Counters()
        r0, r1 = rows if rows else (0, H)
        rc = lib().orc_render(self._h, C.byref(cam), ALGOS[algorithm], s_begin, s_count, max_depth, seed, r0, r1,
                              out.ctypes.data, C.byref(c), nthreads)
        if rc: raise RuntimeError(f"orc_render {rc}")
        return out, c.as_dict()


def camera(aspect_ratio, image_width, vfov, look_from, look_at, v_up, defocus_angle, focus_dist):
    ci = CameraIn(aspect_ratio, image_width, 0, vfov, (C.c_double * 3)(*look_from), (C.c_double * 3)(*look_at),
                  (C.c_double * 3)(*v_up), defocus_angle, focus_dist)
    co = CameraOut()
    lib().orc_camera_init(C.byref(ci), C.byref(co))
    return co


def camera_from_rt(rt_cam):
    """Copy the product's rt_camera (identical layout) into the oracle's struct."""
    co = CameraOut()
    C.memmove(C.byref(co), C.byref(rt_cam), C.sizeof(co))
    return co


def generate(cam, seed, sample, pix_i, pix_j):
    pi = np.ascontiguousarray(pix_i, dtype=np.uint32); pj = np.ascontiguousarray(pix_j, dtype=np.uint32)
    o = np.empty((len(pi), 3)); d = np.empty((len(pi), 3))
    lib().orc_generate(C.byref(cam), seed, sample, pi.ctypes.data, pj.ctypes.data, len(pi), o.ctypes.data, d.ctypes.data)
    return o, d


def resolve(summed, spp_total):
    summed = np.ascontiguousarray(summed, dtype=np.float64)
    rgb = np.empty(summed.shape, np.uint8)
    lib().orc_resolve(summed.ctypes.data, summed.size // 3, spp_total, rgb.ctypes.data)
    return rgb


def rng_draw(seed, path_lo, path_hi, stream, k):
    return lib().orc_rng_draw(seed, path_lo, path_hi, stream, k)
