"""Thin Python handles over the C++ host library and the HIP library.

The names follow the reference (Davvos11/raytracer): ``HittableList`` (src/hittable/hittable_list.rs:14-25) is the
scene container with ``algorithm`` / ``options`` / ``init()`` / ``hit()``; ``Camera.render`` (src/camera.rs:53) is the
path loop; ``run`` is ``run(Cli)`` (src/main.rs:31).  Every method forwards to native code — the host library for
loading and acceleration-structure builds, ``librt_hip.so`` for anything that traces rays.
"""
import ctypes as C

import numpy as np

from . import _ffi as F


class RtError(RuntimeError):
    pass


def _hip_check(rc):
    if rc != 0:
        raise RtError(f"rt_hip error {rc}: {F.hip_lib().rt_last_error().decode()}")


_PRIM_DT = np.dtype([("type", "<u4"), ("material", "<u4"), ("v", "<f8", (9,))])
_MAT_DT = np.dtype([("type", "<u4"), ("pad", "<u4"), ("albedo", "<f8", (3,)), ("param", "<f8")])
_NODE_DT = np.dtype([("min", "<f8", (3,)), ("max", "<f8", (3,)), ("is_leaf", "<u4"), ("left", "<u4"), ("right", "<u4"),
                     ("first", "<u4"), ("count", "<u4"), ("pad", "<u4")])
assert _PRIM_DT.itemsize == C.sizeof(F.RtPrim) and _MAT_DT.itemsize == C.sizeof(F.RtMaterial)
assert _NODE_DT.itemsize == C.sizeof(F.RtBvhNode)


def _view(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(C.addressof(ptr.contents))
    return np.frombuffer(buf, dtype=dtype, count=n)


def init(device=0):
    """rt_init: bind this process to one GPU.  Raises when no GPU / no library is present."""
    _hip_check(F.hip_lib().rt_init(int(device)))


def set_profiling(on):
    _hip_check(F.hip_lib().rt_set_profiling(1 if on else 0))


class Bvh:
    """Host-built Bvh (src/acceleration/bvh.rs:12-16) in the reference's own node layout."""

    def __init__(self, handle):
        self._h = handle
        lib = F.host_lib()
        self.n_nodes = lib.rth_bvh_n_nodes(handle)
        self.depth = lib.rth_bvh_depth(handle)

    def nodes(self):
        return _view(F.host_lib().rth_bvh_nodes(self._h), self.n_nodes, _NODE_DT).copy()

    def order(self, n):
        return _view(F.host_lib().rth_bvh_order(self._h), n, np.dtype("<u4")).copy()

    def __del__(self):
        if getattr(self, "_h", None):
            F.host_lib().rth_bvh_free(self._h)
            self._h = None


class Grid:
    """Host-built Grid (src/acceleration/grid.rs:11-18) as flat tables."""

    def __init__(self, handle):
        self._h = handle
        lib = F.host_lib()
        self.desc = lib.rth_grid_desc(handle).contents
        self.n_cells = int(self.desc.n_cells)
        self.list_len = int(lib.rth_grid_list_len(handle))

    def cell_aabb(self):
        return _view(F.host_lib().rth_grid_cell_aabb(self._h), self.n_cells * 6, np.dtype("<f8")).reshape(-1, 6).copy()

    def cell_offset(self):
        return _view(F.host_lib().rth_grid_cell_offset(self._h), self.n_cells + 1, np.dtype("<u8")).copy()

    def prim_list(self):
        return _view(F.host_lib().rth_grid_prim_list(self._h), self.list_len, np.dtype("<u4")).copy()

    def __del__(self):
        if getattr(self, "_h", None):
            F.host_lib().rth_grid_free(self._h)
            self._h = None


class HittableList:
    """The scene: objects in file order plus the chosen acceleration structure.

    ``algorithm`` is one of "naive" | "bvh" | "grid" (IntersectionAlgorithm, src/utils/rtweekend.rs:57-62);
    ``bvh_mode`` one of "sah-position" | "sah-plane" | "naive" (AlgorithmOptions, rtweekend.rs:137-147).
    """

    ALGOS = {"naive": F.RT_ALGO_NAIVE, "bvh": F.RT_ALGO_BVH, "grid": F.RT_ALGO_GRID}
    BVH_MODES = {"sah-position": F.BVH_SAH_POSITION, "sah-plane": F.BVH_SAH_PLANE, "naive": F.BVH_NAIVE}

    def __init__(self, handle, filename=""):
        self._h = handle
        self.filename = filename
        self.algorithm = "bvh"
        self.bvh_mode = "sah-position"
        self.grid_size = 25.0
        self.draw_boxes = False          # Options.draw_boxes (rtweekend.rs:92), `-o draw-boxes`
        self.gpu_build = False           # build the BVH (any mode) / the grid on the GPU (rt_bvh_build_mode, rt_scene_build_grid)
        self.bvh = None
        self.grid = None
        self._dev = None
        self._grid_on_device = False

    # ---- construction
    @classmethod
    def from_json(cls, path):
        h = F.host_lib().rth_scene_load(path.encode(), F.FORMAT_NATIVE)
        if not h:
            raise RtError(F.host_lib().rth_last_error().decode())
        return cls(h, path)

    @classmethod
    def from_ply(cls, path, subdivide=0):
        h = F.host_lib().rth_scene_load(path.encode(), F.FORMAT_PLY)
        if not h:
            raise RtError(F.host_lib().rth_last_error().decode())
        if subdivide:
            F.host_lib().rth_scene_subdivide(h, int(subdivide))
        return cls(h, path)

    @classmethod
    def from_arrays(cls, prims, centroids, mats, filename=""):
        prims = np.ascontiguousarray(prims, dtype=_PRIM_DT)
        mats = np.ascontiguousarray(mats, dtype=_MAT_DT)
        centroids = np.ascontiguousarray(centroids, dtype=np.float64)
        h = F.host_lib().rth_scene_from_arrays(prims.ctypes.data_as(C.POINTER(F.RtPrim)), centroids.ctypes.data, len(prims),
                                               mats.ctypes.data_as(C.POINTER(F.RtMaterial)), len(mats))
        return cls(h, filename)

    def append(self, other):
        F.host_lib().rth_scene_append(self._h, other._h)

    def __len__(self):
        return F.host_lib().rth_scene_n_prims(self._h)

    def prims(self):
        return _view(F.host_lib().rth_scene_prims(self._h), len(self), _PRIM_DT).copy()

    def centroids(self):
        return _view(F.host_lib().rth_scene_centroids(self._h), 3 * len(self), np.dtype("<f8")).reshape(-1, 3).copy()

    def materials(self):
        return _view(F.host_lib().rth_scene_mats(self._h), F.host_lib().rth_scene_n_mats(self._h), _MAT_DT).copy()

    # ---- world.init() (hittable_list.rs:33-53): host build; upload() flattens it onto the GPU
    def init(self):
        lib = F.host_lib()
        if self.algorithm == "bvh":
            if self.gpu_build:
                F.hip_lib()
                h = lib.rth_bvh_build_gpu(self._h, self.BVH_MODES[self.bvh_mode])
            else:
                h = lib.rth_bvh_build(self._h, self.BVH_MODES[self.bvh_mode])
            if not h:
                raise RtError(lib.rth_last_error().decode())
            self.bvh = Bvh(h)
        elif self.algorithm == "grid" and self.gpu_build:
            self.grid = None; self._grid_on_device = True          # binned by upload(), on the GPU
        elif self.algorithm == "grid":
            self._grid_on_device = False
            h = lib.rth_grid_build(self._h, float(self.grid_size))
            if not h:
                raise RtError(lib.rth_last_error().decode())
            self.grid = Grid(h)
        return self

    def upload(self):
        """rt_scene_create + rt_scene_set_bvh / rt_scene_set_grid for whatever init() built."""
        hip, host = F.hip_lib(), F.host_lib()
        if self._dev is None:
            dev = hip.rt_scene_create(host.rth_scene_prims(self._h), len(self), host.rth_scene_mats(self._h),
                                      host.rth_scene_n_mats(self._h))
            if not dev:
                raise RtError(hip.rt_last_error().decode())
            self._dev = dev
        _hip_check(hip.rt_scene_set_options(self._dev, F.RT_OPT_DRAW_BOXES if self.draw_boxes else 0))
        if self.bvh is not None:
            _hip_check(hip.rt_scene_set_bvh(self._dev, host.rth_bvh_nodes(self.bvh._h), self.bvh.n_nodes,
                                            host.rth_bvh_order(self.bvh._h), len(self)))
        if self.grid is not None:
            _hip_check(hip.rt_scene_set_grid(self._dev, host.rth_grid_desc(self.grid._h), host.rth_grid_cell_aabb(self.grid._h),
                                             host.rth_grid_cell_offset(self.grid._h), host.rth_grid_prim_list(self.grid._h)))
        elif self._grid_on_device:
            _hip_check(hip.rt_scene_build_grid(self._dev, float(self.grid_size)))
        return self

    # ---- Hittable::hit for a batch of rays (hittable_list.rs:65-102), on the GPU
    def hit(self, orig, direction, tmin=0.001, tmax=float("inf"), algorithm=None):
        if self._dev is None:
            raise RtError("upload() first")
        orig = np.ascontiguousarray(orig, dtype=np.float64).reshape(-1, 3)
        direction = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
        n = len(orig)
        prim = np.empty(n, np.int32); t = np.empty(n, np.float64); tp = np.empty(n, np.float64); hit = np.empty(n, np.uint8)
        st = F.RtStats()
        algo = self.ALGOS[algorithm or self.algorithm]
        _hip_check(F.hip_lib().rt_trace(self._dev, algo, orig.ctypes.data, direction.ctypes.data, n, tmin, tmax,
                                        prim.ctypes.data, t.ctypes.data, tp.ctypes.data, hit.ctypes.data, C.byref(st)))
        return {"prim": prim, "t": t, "tp": tp, "hit": (hit & 1).astype(bool), "edge": (hit & 2).astype(bool), "stats": st.as_dict()}

    def build_grid_on_device(self, grid_size=None):
        """Grid::new on the GPU (rt_scene_build_grid): bins the uploaded objects and installs the grid; no host tables."""
        if self._dev is None:
            self.upload()
        _hip_check(F.hip_lib().rt_scene_build_grid(self._dev, float(self.grid_size if grid_size is None else grid_size)))
        return self

    def device_grid(self):
        """(desc, cell_aabb, cell_offset, prim_list) of the grid installed on the device, copied back."""
        hip = F.hip_lib(); d = F.RtGridDesc(); ln = C.c_uint64()
        _hip_check(hip.rt_scene_get_grid(self._dev, C.byref(d), C.byref(ln), None, None, None))
        aabb = np.empty((d.n_cells, 6)); off = np.empty(d.n_cells + 1, np.uint64); pl = np.empty(max(ln.value, 1), np.uint32)
        _hip_check(hip.rt_scene_get_grid(self._dev, None, None, aabb.ctypes.data, off.ctypes.data, pl.ctypes.data))
        return d, aabb, off, pl[:ln.value]

    def bvh_layout(self):
        """(bytes per sibling pair, bytes per primitive) the BVH extend kernel reads for this scene."""
        a, b = C.c_uint32(), C.c_uint32()
        _hip_check(F.hip_lib().rt_scene_bvh_layout(self._dev, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self._dev is not None:
            F.hip_lib().rt_scene_destroy(self._dev)
            self._dev = None

    def __del__(self):
        try:
            self.close()
            if getattr(self, "_h", None):
                F.host_lib().rth_scene_free(self._h)
                self._h = None
        except Exception:
            pass


class Camera:
    """Camera (src/camera.rs:14-36) with the presets of run() (src/main.rs:72-104)."""

    def __init__(self, filename, camera=None, image_width=None, samples_per_pixel=None, max_depth=None):
        self.c = F.RtCamera()
        spp, md = C.c_uint32(), C.c_uint32()
        rc = F.host_lib().rth_camera_preset(filename.encode(), -1 if camera is None else int(camera), int(image_width or -1),
                                            C.byref(self.c), C.byref(spp), C.byref(md))
        if rc != 0:
            raise RtError(F.host_lib().rth_last_error().decode())
        self.samples_per_pixel = samples_per_pixel or spp.value
        self.max_depth = md.value if max_depth is None else max_depth
        self.image_width, self.image_height = self.c.image_width, self.c.image_height

    def get_rays(self, pix_i, pix_j, sample=0, seed=1):
        """Camera::get_ray (camera.rs:132-148) for explicit pixels, on the GPU."""
        pi = np.ascontiguousarray(pix_i, dtype=np.uint32); pj = np.ascontiguousarray(pix_j, dtype=np.uint32)
        o = np.empty((len(pi), 3)); d = np.empty((len(pi), 3))
        _hip_check(F.hip_lib().rt_generate(C.byref(self.c), seed, sample, pi.ctypes.data, pj.ctypes.data, len(pi), o.ctypes.data, d.ctypes.data))
        return o, d

    def render(self, world, seed=1, s_begin=0, s_count=None, d_out=None, stream=None, noalloc=False):
        """Camera::render (camera.rs:53-82) minus the text output: returns (sum[H,W,3], stats dict).

        With ``d_out`` (a device pointer to W*H*3 doubles) the framebuffer stays in HBM and None is returned for it;
        ``noalloc`` selects rt_render_device_noalloc (static workspace, no allocation inside the call).
        """
        if world._dev is None:
            raise RtError("world.upload() first")
        s_count = self.samples_per_pixel if s_count is None else s_count
        st = F.RtStats()
        algo = world.ALGOS[world.algorithm]
        if d_out is not None:
            fn = F.hip_lib().rt_render_device_noalloc if noalloc else F.hip_lib().rt_render_device
            _hip_check(fn(world._dev, C.byref(self.c), algo, s_begin, s_count, self.max_depth, seed,
                          C.c_void_p(d_out), C.c_void_p(stream or 0), C.byref(st)))
            return None, st.as_dict()
        out = np.empty((self.image_height, self.image_width, 3), np.float64)
        _hip_check(F.hip_lib().rt_render(world._dev, C.byref(self.c), algo, s_begin, s_count, self.max_depth, seed,
                                         out.ctypes.data, C.byref(st)))
        return out, st.as_dict()

    def reserve(self, world, s_count=None):
        """rt_scene_reserve: allocate the render workspace for this camera's image now instead of inside the first render."""
        if world._dev is None:
            raise RtError("world.upload() first")
        _hip_check(F.hip_lib().rt_scene_reserve(world._dev, self.image_width, self.image_height,
                                                int(self.samples_per_pixel if s_count is None else s_count), self.max_depth))

    def resolve(self, summed, spp_total=None):
        """camera.rs:75 + color.rs:6-30: scale by 1/spp, gamma, clamp, quantise → uint8[H,W,3]."""
        summed = np.ascontiguousarray(summed, dtype=np.float64)
        rgb = np.empty(summed.shape, np.uint8)
        F.host_lib().rth_resolve(summed.ctypes.data, summed.size // 3, int(spp_total or self.samples_per_pixel), rgb.ctypes.data)
        return rgb


class Comm:
    """RCCL communicator of the product library (rt_comm_*, include/rt_hip.h): the spp-split's one exchange step,
    ncclReduce(sum, root) of the f64 framebuffers.  `unique_id()` on rank 0, hand the bytes to the other ranks, then every rank
    constructs Comm(id, n_ranks, rank) — a collective call."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(F.RT_COMM_ID_BYTES)
        _hip_check(F.hip_lib().rt_comm_unique_id(buf))
        return buf.raw

    def __init__(self, uid, n_ranks, rank):
        if len(uid) != F.RT_COMM_ID_BYTES:
            raise RtError(f"a communicator id has {F.RT_COMM_ID_BYTES} bytes")
        h = C.c_void_p()
        _hip_check(F.hip_lib().rt_comm_create(uid, int(n_ranks), int(rank), C.byref(h)))
        self._h, self.n_ranks, self.rank = h, int(n_ranks), int(rank)

    def reduce_sum(self, d_ptr, count, root=0, stream=None):
        """In-place sum of `count` doubles at device pointer `d_ptr` onto rank `root`; returns when the reduce has finished."""
        _hip_check(F.hip_lib().rt_comm_reduce_sum(self._h, C.c_void_p(d_ptr), int(count), int(root), C.c_void_p(stream or 0)))

    def render_reduce(self, cam, world, seed=1, s_begin=0, s_count=None, root=0):
        """rt_render_reduce: this rank's samples, then the reduce; the root gets the summed image, the others None."""
        s_count = cam.samples_per_pixel if s_count is None else s_count
        out = np.empty((cam.image_height, cam.image_width, 3), np.float64) if self.rank == root else None
        st = F.RtStats()
        _hip_check(F.hip_lib().rt_render_reduce(world._dev, C.byref(cam.c), world.ALGOS[world.algorithm], s_begin, s_count, cam.max_depth, seed,
                                                self._h, int(root), out.ctypes.data if out is not None else None, C.byref(st)))
        return out, st.as_dict()

    def close(self):
        if getattr(self, "_h", None):
            F.hip_lib().rt_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run(argv):
    """run(Cli) (src/main.rs:31-140): argv as the reference binary takes it, e.g.
    ``run(["scenes/weekend-final.json", "--algorithm", "bvh"])``.  Returns the exit code."""
    args = [b"raytracer"] + [a.encode() for a in argv]
    arr = (C.c_char_p * len(args))(*args)
    rc = F.host_lib().rth_run(len(args), arr)
    if rc != 0:
        raise RtError(F.host_lib().rth_last_error().decode())
    return rc
