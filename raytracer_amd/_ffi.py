"""ctypes bindings of the two product libraries.

* ``librt_hip.so``  — the HIP kernels behind the C ABI of ``include/rt_hip.h`` (the drop-in boundary).
* ``librt_host.so`` — the C++ host side (scene loaders, BVH / grid builders, camera presets, CLI ``run()``).

Nothing here computes anything: the structs mirror the headers field for field and every call goes
straight through.  Loading ``librt_hip.so`` fails loudly when the library has not been built — there is
no Python or CPU fallback for the device path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

# enums of include/rt_hip.h
RT_SPHERE, RT_TRIANGLE = 0, 1
RT_LAMBERTIAN, RT_METAL, RT_DIELECTRIC = 0, 1, 2
RT_ALGO_NAIVE, RT_ALGO_BVH, RT_ALGO_GRID = 0, 1, 2
BVH_SAH_POSITION, BVH_SAH_PLANE, BVH_NAIVE = 0, 1, 2
FORMAT_NATIVE, FORMAT_PLY = 0, 1
RT_OPT_DRAW_BOXES = 1


class RtPrim(C.Structure):
    _fields_ = [("type", C.c_uint32), ("material", C.c_uint32), ("v", C.c_double * 9)]


class RtMaterial(C.Structure):
    _fields_ = [("type", C.c_uint32), ("pad", C.c_uint32), ("albedo", C.c_double * 3), ("param", C.c_double)]


class RtBvhNode(C.Structure):
    _fields_ = [("min", C.c_double * 3), ("max", C.c_double * 3), ("is_leaf", C.c_uint32), ("left", C.c_uint32),
                ("right", C.c_uint32), ("first", C.c_uint32), ("count", C.c_uint32), ("pad", C.c_uint32)]


class RtGridDesc(C.Structure):
    _fields_ = [("box_size", C.c_double * 3), ("origin", C.c_double * 3), ("end", C.c_double * 3),
                ("total_size", C.c_double * 3), ("n_cells", C.c_uint64)]


class RtCamera(C.Structure):
    _fields_ = [("image_width", C.c_uint32), ("image_height", C.c_uint32), ("center", C.c_double * 3),
                ("pixel00", C.c_double * 3), ("du", C.c_double * 3), ("dv", C.c_double * 3), ("ddu", C.c_double * 3),
                ("ddv", C.c_double * 3), ("defocus_angle", C.c_double)]


class RtStats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("scatter_rays", C.c_uint64), ("intersection_checks", C.c_uint64),
                ("traversal_steps", C.c_uint64), ("overlapping_aabb", C.c_uint64),
                ("gridbox_intersection_checks", C.c_uint64), ("box_tests", C.c_uint64), ("prim_tests", C.c_uint64),
                ("ray_segments", C.c_uint64), ("extend_launches", C.c_uint64), ("extend_ms", C.c_double),
                ("shade_ms", C.c_double), ("generate_ms", C.c_double), ("resolve_ms", C.c_double),
                ("render_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


RT_COMM_ID_BYTES = 128

# every symbol include/rt_hip.h declares (tests check the library exports exactly these)
RT_HIP_SYMBOLS = ["rt_init", "rt_set_profiling", "rt_scene_create", "rt_scene_set_bvh", "rt_scene_set_grid", "rt_scene_set_options", "rt_scene_bvh_layout", "rt_bvh_build", "rt_bvh_build_mode", "rt_scene_build_grid", "rt_scene_get_grid", "rt_trace",
                  "rt_generate", "rt_render", "rt_render_device", "rt_render_device_noalloc", "rt_scene_reserve", "rt_scene_destroy", "rt_last_error",
                  "rt_comm_unique_id", "rt_comm_create", "rt_comm_reduce_sum", "rt_comm_destroy", "rt_render_reduce"]

_hip = None
_host = None


def _need(path):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build the native libraries first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or make -C raytracer_amd/csrc). There is no CPU fallback for the HIP path.")
    return path


def hip_lib():
    """librt_hip.so with argument types set."""
    global _hip
    if _hip is not None:
        return _hip
    # RT_HIP_LIB: diagnostic builds of the same library (e.g. lib/librt_hip_stamps.so); default is the product library
    lib = C.CDLL(_need(os.environ.get("RT_HIP_LIB") or os.path.join(LIB_DIR, "librt_hip.so")), mode=C.RTLD_GLOBAL)
    P = C.POINTER
    lib.rt_init.argtypes = [C.c_int]; lib.rt_init.restype = C.c_int
    lib.rt_set_profiling.argtypes = [C.c_int]; lib.rt_set_profiling.restype = C.c_int
    lib.rt_scene_create.argtypes = [P(RtPrim), C.c_uint32, P(RtMaterial), C.c_uint32]; lib.rt_scene_create.restype = C.c_void_p
    lib.rt_scene_set_bvh.argtypes = [C.c_void_p, P(RtBvhNode), C.c_uint32, P(C.c_uint32), C.c_uint32]; lib.rt_scene_set_bvh.restype = C.c_int
    lib.rt_scene_set_grid.argtypes = [C.c_void_p, P(RtGridDesc), P(C.c_double), P(C.c_uint64), P(C.c_uint32)]; lib.rt_scene_set_grid.restype = C.c_int
    lib.rt_scene_set_options.argtypes = [C.c_void_p, C.c_uint32]; lib.rt_scene_set_options.restype = C.c_int
    lib.rt_bvh_build.argtypes = [P(RtPrim), C.c_void_p, C.c_uint32, P(RtBvhNode), P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]; lib.rt_bvh_build.restype = C.c_int
    lib.rt_bvh_build_mode.argtypes = [P(RtPrim), C.c_void_p, C.c_uint32, C.c_int, P(RtBvhNode), P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]; lib.rt_bvh_build_mode.restype = C.c_int
    lib.rt_scene_build_grid.argtypes = [C.c_void_p, C.c_double]; lib.rt_scene_build_grid.restype = C.c_int
    lib.rt_scene_get_grid.argtypes = [C.c_void_p, P(RtGridDesc), P(C.c_uint64), C.c_void_p, C.c_void_p, C.c_void_p]; lib.rt_scene_get_grid.restype = C.c_int
    lib.rt_scene_bvh_layout.argtypes = [C.c_void_p, P(C.c_uint32), P(C.c_uint32)]; lib.rt_scene_bvh_layout.restype = C.c_int
    lib.rt_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, P(RtStats)]
    lib.rt_trace.restype = C.c_int
    lib.rt_generate.argtypes = [P(RtCamera), C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.rt_generate.restype = C.c_int
    lib.rt_render.argtypes = [C.c_void_p, P(RtCamera), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, P(RtStats)]
    lib.rt_render.restype = C.c_int
    lib.rt_render_device.argtypes = [C.c_void_p, P(RtCamera), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                     C.c_void_p, C.c_void_p, P(RtStats)]
    lib.rt_render_device.restype = C.c_int
    lib.rt_render_device_noalloc.argtypes = lib.rt_render_device.argtypes
    lib.rt_render_device_noalloc.restype = C.c_int
    lib.rt_scene_reserve.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]; lib.rt_scene_reserve.restype = C.c_int
    lib.rt_scene_destroy.argtypes = [C.c_void_p]; lib.rt_scene_destroy.restype = None
    lib.rt_last_error.argtypes = []; lib.rt_last_error.restype = C.c_char_p
    lib.rt_comm_unique_id.argtypes = [C.c_void_p]; lib.rt_comm_unique_id.restype = C.c_int
    lib.rt_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, P(C.c_void_p)]; lib.rt_comm_create.restype = C.c_int
    lib.rt_comm_reduce_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]; lib.rt_comm_reduce_sum.restype = C.c_int
    lib.rt_comm_destroy.argtypes = [C.c_void_p]; lib.rt_comm_destroy.restype = C.c_int
    lib.rt_render_reduce.argtypes = [C.c_void_p, P(RtCamera), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, P(RtStats)]
    lib.rt_render_reduce.restype = C.c_int
    _hip = lib
    return lib


def host_lib():
    """librt_host.so with argument types set (it links librt_hip.so)."""
    global _host
    if _host is not None:
        return _host
    hip_lib()
    lib = C.CDLL(_need(os.path.join(LIB_DIR, "librt_host.so")))
    P = C.POINTER
    lib.rth_last_error.restype = C.c_char_p
    lib.rth_scene_load.argtypes = [C.c_char_p, C.c_int]; lib.rth_scene_load.restype = C.c_void_p
    lib.rth_scene_from_arrays.argtypes = [P(RtPrim), C.c_void_p, C.c_uint32, P(RtMaterial), C.c_uint32]; lib.rth_scene_from_arrays.restype = C.c_void_p
    lib.rth_scene_free.argtypes = [C.c_void_p]; lib.rth_scene_free.restype = None
    lib.rth_scene_subdivide.argtypes = [C.c_void_p, C.c_int]; lib.rth_scene_subdivide.restype = None
    lib.rth_scene_append.argtypes = [C.c_void_p, C.c_void_p]; lib.rth_scene_append.restype = None
    lib.rth_scene_n_prims.argtypes = [C.c_void_p]; lib.rth_scene_n_prims.restype = C.c_uint32
    lib.rth_scene_n_mats.argtypes = [C.c_void_p]; lib.rth_scene_n_mats.restype = C.c_uint32
    lib.rth_scene_prims.argtypes = [C.c_void_p]; lib.rth_scene_prims.restype = P(RtPrim)
    lib.rth_scene_centroids.argtypes = [C.c_void_p]; lib.rth_scene_centroids.restype = P(C.c_double)
    lib.rth_scene_mats.argtypes = [C.c_void_p]; lib.rth_scene_mats.restype = P(RtMaterial)
    lib.rth_bvh_build.argtypes = [C.c_void_p, C.c_int]; lib.rth_bvh_build.restype = C.c_void_p
    lib.rth_bvh_build_gpu.argtypes = [C.c_void_p, C.c_int]; lib.rth_bvh_build_gpu.restype = C.c_void_p
    lib.rth_bvh_free.argtypes = [C.c_void_p]; lib.rth_bvh_free.restype = None
    lib.rth_bvh_n_nodes.argtypes = [C.c_void_p]; lib.rth_bvh_n_nodes.restype = C.c_uint32
    lib.rth_bvh_depth.argtypes = [C.c_void_p]; lib.rth_bvh_depth.restype = C.c_uint32
    lib.rth_bvh_nodes.argtypes = [C.c_void_p]; lib.rth_bvh_nodes.restype = P(RtBvhNode)
    lib.rth_bvh_order.argtypes = [C.c_void_p]; lib.rth_bvh_order.restype = P(C.c_uint32)
    lib.rth_grid_build.argtypes = [C.c_void_p, C.c_double]; lib.rth_grid_build.restype = C.c_void_p
    lib.rth_grid_free.argtypes = [C.c_void_p]; lib.rth_grid_free.restype = None
    lib.rth_grid_desc.argtypes = [C.c_void_p]; lib.rth_grid_desc.restype = P(RtGridDesc)
    lib.rth_grid_cell_aabb.argtypes = [C.c_void_p]; lib.rth_grid_cell_aabb.restype = P(C.c_double)
    lib.rth_grid_cell_offset.argtypes = [C.c_void_p]; lib.rth_grid_cell_offset.restype = P(C.c_uint64)
    lib.rth_grid_prim_list.argtypes = [C.c_void_p]; lib.rth_grid_prim_list.restype = P(C.c_uint32)
    lib.rth_grid_list_len.argtypes = [C.c_void_p]; lib.rth_grid_list_len.restype = C.c_uint64
    lib.rth_camera_preset.argtypes = [C.c_char_p, C.c_int, C.c_int, P(RtCamera), P(C.c_uint32), P(C.c_uint32)]; lib.rth_camera_preset.restype = C.c_int
    lib.rth_resolve.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]; lib.rth_resolve.restype = None
    lib.rth_write_ppm.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]; lib.rth_write_ppm.restype = C.c_int
    lib.rth_run.argtypes = [C.c_int, P(C.c_char_p)]; lib.rth_run.restype = C.c_int
    lib.rth_output_filename.argtypes = [C.c_int, P(C.c_char_p), C.c_char_p, C.c_uint32]; lib.rth_output_filename.restype = C.c_int
    _host = lib
    return lib
