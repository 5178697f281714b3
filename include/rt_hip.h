/*
 * rt_hip.h — C ABI of the MI355X (gfx950) wavefront path tracer, librt_hip.{so,a}.
 *
 * Drop-in boundary for ONE path of Davvos11/raytracer: what `run()` does between
 * `world.init()` (src/main.rs:128) and the end of `cam.render(&world, &mut file, &mut data)`
 * (src/main.rs:132, src/camera.rs:53-82).  The reference has no FFI; each entry point below
 * names the reference interface it replaces.  All structs are POD with fixed-width fields,
 * all pointers are plain host pointers unless a name starts with `d_`, all calls return
 * 0 on success or a negative RT_E_* code (no exceptions or panics cross the ABI) and
 * rt_last_error() describes the last failure on the calling thread.
 *
 * Threading: one host thread per process, one GPU per process (the device current when
 * rt_init() ran).  Calls are blocking unless stated.  The library owns every device
 * allocation it makes; host arrays are borrowed for the duration of the call.
 */
#ifndef RT_HIP_H
#define RT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { RT_OK = 0, RT_E_ARG = -1, RT_E_HIP = -2, RT_E_STATE = -3, RT_E_LIMIT = -4 };

/* trait Hittable implementors (src/hittable/sphere.rs:13-17, src/hittable/triangle.rs:12-18) */
enum { RT_SPHERE = 0, RT_TRIANGLE = 1 };
/* trait Material implementors (src/value/material.rs:23-25, 54-57, 82-84) */
enum { RT_LAMBERTIAN = 0, RT_METAL = 1, RT_DIELECTRIC = 2 };
/* IntersectionAlgorithm (src/utils/rtweekend.rs:57-62) */
enum { RT_ALGO_NAIVE = 0, RT_ALGO_BVH = 1, RT_ALGO_GRID = 2 };

/* One Hittable.  sphere: v[0..2] = center, v[3] = radius.  triangle: v[0..2]=v0, v[3..5]=v1, v[6..8]=v2.
 * `material` indexes the rt_material array. */
typedef struct { uint32_t type; uint32_t material; double v[9]; } rt_prim;

/* One Material.  Lambertian: albedo.  Metal: albedo, param = fuzz.  Dielectric: param = refraction_index. */
typedef struct { uint32_t type; uint32_t pad; double albedo[3]; double param; } rt_material;

/* BvhNode, field for field (src/acceleration/bvh.rs:35-42); `left`/`right` index the node array,
 * `first`/`count` index the permuted object order handed to rt_scene_set_bvh(). */
typedef struct { double min[3]; double max[3]; uint32_t is_leaf, left, right, first, count, pad; } rt_bvh_node;

/* Grid (src/acceleration/grid.rs:11-18) minus the object and box vectors, which travel as flat tables. */
typedef struct {
    double box_size[3]; double origin[3]; double end[3]; double total_size[3];
    uint64_t n_cells;
} rt_grid_desc;

/* Outputs of Camera::initialise (src/camera.rs:88-128) that get_ray (src/camera.rs:132-148) reads;
 * the device does no trigonometry. */
typedef struct {
    uint32_t image_width, image_height;
    double center[3], pixel00[3], du[3], dv[3], ddu[3], ddv[3];
    double defocus_angle;
} rt_camera;

/* Data counters (src/value/data.rs:10-15), the split of intersection_checks by call site
 * (box: bvh.rs:183, grid.rs:172 — primitive: hittable_list.rs:72, bvh.rs:197, grid.rs:287), and
 * per-kernel device time measured with HIP events when rt_set_profiling(1) is on. */
typedef struct {
    uint64_t primary_rays, scatter_rays, intersection_checks, traversal_steps,
             overlapping_aabb, gridbox_intersection_checks;
    uint64_t box_tests, prim_tests;
    uint64_t ray_segments;        /* rays pushed through extend = primary + continued scatter rays */
    uint64_t extend_launches;
    double   extend_ms;           /* sum of event-timed extend launches (0 when profiling is off) */
    double   shade_ms, generate_ms, resolve_ms;
    double   render_ms;           /* host wall time of the call, device synchronised on both sides */
} rt_stats;

typedef struct rt_scene rt_scene;

/* Bind this process to `device` (hipSetDevice) and create the library's streams.  No reference
 * counterpart (the reference is CPU only).  Must be called once before anything else. */
int rt_init(int device);

/* Per-launch HIP-event timing of the kernels on/off (default off). */
int rt_set_profiling(int on);

/* Replaces building `HittableList.objects` (src/hittable/hittable_list.rs:15-16): uploads the
 * primitives (file order) and materials.  Returns NULL on failure. */
rt_scene* rt_scene_create(const rt_prim* prims, uint32_t n_prims, const rt_material* mats, uint32_t n_mats);

/* Replaces storing the result of Bvh::new in HittableList.bvh (hittable_list.rs:37): takes the
 * host-built tree in the reference's own node layout and object permutation (`prim_index[k]` is
 * the file-order index of the k-th object of Bvh.objects) and flattens it for the device. */
int rt_scene_set_bvh(rt_scene*, const rt_bvh_node* nodes, uint32_t n_nodes, const uint32_t* prim_index, uint32_t n_prims);

/* Replaces HittableList.grid = Grid::new(...) (hittable_list.rs:43): per-cell AABB (min xyz, max xyz),
 * CSR offsets (n_cells + 1) and the concatenated GridBox.objects lists (file-order indices). */
int rt_scene_set_grid(rt_scene*, const rt_grid_desc*, const double* cell_aabb, const uint64_t* cell_offset, const uint32_t* prim_list);

/* Replaces HittableList.options (hittable_list.rs:20, Options rtweekend.rs:88-93) for the one option that acts on the
 * hot path: RT_OPT_DRAW_BOXES = `-o draw-boxes` (rtweekend.rs:92,128,146).  When set, every successful AABB test whose
 * entry point lies on a box edge marks the ray (aabb.rs:57-60,90-97) and ray_color returns red for it
 * (camera.rs:162-164,178-180); rt_trace reports the mark as bit 1 of out_hit.  The BVH build options are consumed by
 * the host builder and never reach the device. */
enum { RT_OPT_DRAW_BOXES = 1 };
int rt_scene_set_options(rt_scene*, uint32_t flags);

/* Replaces Bvh::new (src/acceleration/bvh.rs:19-180) for the default BvhSahPosition mode, on the GPU: the same nodes
 * (indices 0..=node_pointer, bvh.rs:14-15), numbering and object permutation (bvh.rs:13) as the reference's top-down SAH
 * build.  prims/centroids are host arrays (centroid = Hittable::centroid(), n*3 doubles); nodes_out must hold 2*n_prims
 * nodes, order_out n_prims indices. */
int rt_bvh_build(const rt_prim* prims, const double* centroids, uint32_t n_prims,
                 rt_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* order_out, uint32_t* depth_out);
/* The same for any AlgorithmOptions build mode (rtweekend.rs:137-147; get_split bvh.rs:86-142): RT_BVH_SAH_POSITION (the default,
 * every split position of every axis, bvh.rs:124-133), RT_BVH_SAH_PLANE (`-o bvh-sah-plane`: the median of each axis, bvh.rs:114-123),
 * RT_BVH_NAIVE (`-o bvh-naive`: median of the x order for nodes of >= 3 objects, bvh.rs:87-99). */
enum { RT_BVH_SAH_POSITION = 0, RT_BVH_SAH_PLANE = 1, RT_BVH_NAIVE = 2 };
int rt_bvh_build_mode(const rt_prim* prims, const double* centroids, uint32_t n_prims, int mode,
                      rt_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* order_out, uint32_t* depth_out);

/* Replaces HittableList.grid = Grid::new(...) (hittable_list.rs:43, grid.rs:23-61) entirely on the GPU: bins the scene's
 * objects into the fixed [-100,100]^3 grid of cell size `grid_size` with the reference's table layout (get_index slots,
 * last-writer cells, origin-cell clones, ascending object order per cell) and installs it as the scene's grid.
 * rt_scene_get_grid copies the installed tables back (any pointer may be NULL; sizes come from desc->n_cells and *list_len);
 * the reference has no counterpart for it (its Grid fields are private) — it exists for the parity tests. */
int rt_scene_build_grid(rt_scene*, double grid_size);
int rt_scene_get_grid(const rt_scene*, rt_grid_desc* desc, uint64_t* list_len, double* cell_aabb, uint64_t* cell_offset, uint32_t* prim_list);

/* Bytes the BVH extend kernel reads per sibling pair (two box tests) and per primitive test with the layout chosen for
 * this scene by rt_scene_set_bvh: 112 / 80 in general, 112 / 48 for all-sphere scenes, 64 / 48 when every coordinate lies on
 * an f32 lattice (PLY meshes: parser.rs:67-73).  No reference counterpart; bench.py prices the kernel's traffic with it. */
int rt_scene_bvh_layout(const rt_scene*, uint32_t* pair_bytes, uint32_t* prim_bytes);

/* Parity entry: one `HittableList::hit(r, Interval(tmin,tmax), &mut HitRecord::default(), data)`
 * (hittable_list.rs:65-102) per ray.  orig/dir are n*3 doubles (xyz interleaved).  Any output may be NULL.
 * out_prim: file-order index of the last accepted primitive (-1: none); out_t: rec.t after the call;
 * out_tp: the t at which rec.p was last written (rec.p == orig + out_tp*dir); out_hit: bit 0 = the return value,
 * bit 1 = rec.hits_aabb_edge (hittable.rs:20; only ever set under RT_OPT_DRAW_BOXES). */
int rt_trace(rt_scene*, int algorithm, const double* orig, const double* dir, uint64_t n, double tmin, double tmax,
             int32_t* out_prim, double* out_t, double* out_tp, uint8_t* out_hit, rt_stats* stats);

/* Parity entry for Camera::get_ray (camera.rs:132-148): primary ray of pixel (pix_i[k], pix_j[k]) for
 * global sample index `sample`, with the library's counter RNG. */
int rt_generate(const rt_camera*, uint64_t seed, uint32_t sample, const uint32_t* pix_i, const uint32_t* pix_j,
                uint64_t n, double* orig, double* dir);

/* Replaces the loops of Camera::render (camera.rs:64-73) for samples [s_begin, s_begin+s_count) of every
 * pixel: out_sum[(j*W+i)*3+c] receives the sum of ray_color (camera.rs:153-186) over those samples, added
 * in sample order.  `seed` keys the counter RNG; results do not depend on how samples are split over calls
 * or GPUs (up to f64 summation order of the partial sums).  out_sum is a host buffer of W*H*3 doubles. */
int rt_render(rt_scene*, const rt_camera*, int algorithm, uint32_t s_begin, uint32_t s_count, uint32_t max_depth,
              uint64_t seed, double* out_sum, rt_stats* stats);

/* Same, but the framebuffer stays in HBM: d_out_sum is a device buffer of W*H*3 doubles that is
 * overwritten.  The call waits (on the host) for prior work on `stream` and is complete on return.  `stream` is a
 * hipStream_t: the kernels run on it; NULL means the legacy default stream, in which case the kernels run on the library's
 * own non-blocking stream after the default stream has drained.  This is what an spp-split multi-GPU host reduces with RCCL.
 * max_depth == 0 gives a black frame and no counters but primary_rays (camera.rs:155-157); max_depth > 2^20 is RT_E_LIMIT. */
int rt_render_device(rt_scene*, const rt_camera*, int algorithm, uint32_t s_begin, uint32_t s_count, uint32_t max_depth,
                     uint64_t seed, double* d_out_sum, void* stream, rt_stats* stats);

/* No counterpart in the reference.  rt_render_device for callers that cannot allocate (code under HIP graph capture set-up,
 * tight serving loops, harnesses that forbid hipMalloc): same arguments, same result bit for bit, but the ray queues, hit
 * queue, per-path radiance records, queue lengths and counters live in a static device array of 50 MB that each GPU holds
 * from library load to process end, and the call makes no hipMalloc / hipFree / hipHostMalloc / event creation.  Limits
 * (RT_E_LIMIT beyond them): image_width * image_height <= 262144 pixels (samples are batched to fit 262144 paths in flight),
 * max_depth <= 1022.  Queue lengths are not read back: all max_depth bounces are launched, the ones that find an empty queue
 * do nothing.  Per-kernel times in rt_stats stay 0.  Every call on a GPU uses that GPU's one array: calls that overlap on one
 * GPU (two host threads, two processes do not share it) race; the scene must have been set up (rt_scene_set_bvh / _grid)
 * before, since those calls do allocate. */
int rt_render_device_noalloc(rt_scene*, const rt_camera*, int algorithm, uint32_t s_begin, uint32_t s_count, uint32_t max_depth,
                             uint64_t seed, double* d_out_sum, void* stream, rt_stats* stats);

/* No counterpart in the reference (its render owns no buffers that outlive a call).  Sizes and allocates, now, the per-scene
 * workspace that rt_render / rt_render_device of an image_width x image_height image with s_count samples per call would
 * allocate on first use (ray queues, hit queue, per-path radiance: 196 B per path in flight, up to 75 % of the free HBM), so
 * that the first timed render of a serving or benchmark loop does not pay for hipMalloc.  Optional; renders of other
 * sizes still resize the workspace themselves. */
int rt_scene_reserve(rt_scene*, uint32_t image_width, uint32_t image_height, uint32_t s_count, uint32_t max_depth);

/* ---- multi-GPU: the one exchange step of the path.  No counterpart in the reference (single process, CPU).
 * Samples are split over ranks, one process per GPU, every rank renders its global sample indices of every pixel with a
 * replicated scene; the W*H*3 f64 framebuffers are then summed to one rank with ONE RCCL ncclReduce(sum) over xGMI.
 * librccl is opened with dlopen on first use (no link-time dependency): without it these calls return RT_E_STATE.
 *   rt_comm_unique_id   ncclGetUniqueId: rank 0 calls it and hands the RT_COMM_ID_BYTES to the other ranks by any means
 *                       (the CLI uses a shared mapping, a launcher its own store)
 *   rt_comm_create      ncclCommInitRank on the device of rt_init(); collective over the n_ranks processes
 *   rt_comm_reduce_sum  in-place ncclReduce(ncclDouble, ncclSum) of `count` doubles at d_buf to `root`, enqueued on `stream`
 *                       (NULL: the library's stream) and waited for; on the other ranks d_buf is left unspecified
 *   rt_render_reduce    rt_render_device into a library-owned device framebuffer + rt_comm_reduce_sum + (root only) the
 *                       copy to the host array out_sum (W*H*3; may be NULL on the other ranks).  Counters are per rank.
 * One GPU can belong to a communicator only once: ranks that share a GPU (rehearsals) cannot use these calls. */
typedef struct rt_comm rt_comm;
#define RT_COMM_ID_BYTES 128
int rt_comm_unique_id(void* id_out);
int rt_comm_create(const void* id, int n_ranks, int rank, rt_comm** out);
int rt_comm_reduce_sum(rt_comm*, double* d_buf, uint64_t count, int root, void* stream);
int rt_comm_destroy(rt_comm*);
int rt_render_reduce(rt_scene*, const rt_camera*, int algorithm, uint32_t s_begin, uint32_t s_count, uint32_t max_depth,
                     uint64_t seed, rt_comm*, int root, double* out_sum, rt_stats* stats);

/* Replaces dropping the HittableList. */
void rt_scene_destroy(rt_scene*);

const char* rt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
