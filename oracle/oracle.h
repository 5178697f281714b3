/*
 * oracle.h — C ABI of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a CPU restatement of the reference's
 * (Davvos11/raytracer, Rust) `ray_color` path loop.  Only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may load it.  The product (raytracer_amd/, include/)
 * never links, loads or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference's own tests (src/test.rs:1-310) hold no assertion, golden
 * vector or fixture for this path, and the reference cannot be built here (no cargo/rustc).
 * The oracle is therefore pinned only by hand-derived known-answer vectors (tests/golden/)
 * and by audit against the cited reference lines.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* primitive types / material types (flat scene format shared with include/rt_hip.h) */
enum { ORC_SPHERE = 0, ORC_TRIANGLE = 1 };
enum { ORC_LAMBERTIAN = 0, ORC_METAL = 1, ORC_DIELECTRIC = 2 };
enum { ORC_ALGO_NAIVE = 0, ORC_ALGO_BVH = 1, ORC_ALGO_GRID = 2 };
enum { ORC_BVH_SAH_POSITION = 0, ORC_BVH_SAH_PLANE = 1, ORC_BVH_NAIVE = 2 };

/* sphere: v[0..2] centre, v[3] radius.  triangle: v[0..2] v0, v[3..5] v1, v[6..8] v2 */
typedef struct { uint32_t type; uint32_t material; double v[9]; } orc_prim;
/* lambertian: albedo.  metal: albedo + param=fuzz.  dielectric: param=refraction index */
typedef struct { uint32_t type; uint32_t pad; double albedo[3]; double param; } orc_material;
/* canonical BVH node, field-for-field src/acceleration/bvh.rs:35-42 */
typedef struct { double min[3]; double max[3]; uint32_t is_leaf, left, right, first, count, pad; } orc_bvh_node;

/* camera inputs (src/camera.rs:14-24) and derived values (src/camera.rs:25-35, initialise():88-128) */
typedef struct {
    double aspect_ratio; uint32_t image_width; uint32_t pad;
    double vfov; double look_from[3]; double look_at[3]; double v_up[3];
    double defocus_angle; double focus_dist;
} orc_camera_in;
typedef struct {
    uint32_t image_width, image_height;
    double center[3], pixel00[3], du[3], dv[3], ddu[3], ddv[3];
    double defocus_angle;
} orc_camera;

/* src/value/data.rs:6-15 counters, plus the split of intersection_checks by call site */
typedef struct {
    uint64_t primary_rays, scatter_rays, intersection_checks, traversal_steps,
             overlapping_aabb, gridbox_intersection_checks;
    uint64_t box_tests;   /* intersection_checks bumped at bvh.rs:183 / grid.rs:172 */
    uint64_t prim_tests;  /* intersection_checks bumped at hittable_list.rs:72, bvh.rs:197, grid.rs:287 */
} orc_counters;

typedef struct orc_scene orc_scene;

orc_scene* orc_scene_create(const orc_prim* prims, const double* centroids /* n*3 */, uint32_t n_prims,
                            const orc_material* mats, uint32_t n_mats);
void orc_scene_destroy(orc_scene*);

/* world.init(): hittable_list.rs:33-53.  bvh_mode per ORC_BVH_*; faithful O(N^2) build (bvh.rs:19-180) */
int orc_build_bvh(orc_scene*, int bvh_mode);
/* the same BvhSahPosition tree (byte-identical nodes and object order) from a second, independent statement of get_split that
 * sweeps prefix boxes instead of re-folding both sides per candidate: finishes on meshes of 10^5..10^6 triangles */
int orc_build_bvh_sweep(orc_scene*);
/* import an externally built BVH (same canonical layout) for scenes too big for the O(N^2) build */
int orc_set_bvh(orc_scene*, const orc_bvh_node* nodes, uint32_t n_nodes, const uint32_t* order, uint32_t n);
uint32_t orc_bvh_node_count(const orc_scene*);                 /* node_pointer + 1 */
int orc_get_bvh(const orc_scene*, orc_bvh_node* nodes_out, uint32_t* order_out);

/* Grid::new grid.rs:23-61 with the fixed bounds of hittable_list.rs:43 */
int orc_build_grid(orc_scene*, double grid_size);
uint64_t orc_grid_cell_count(const orc_scene*);
uint64_t orc_grid_list_len(const orc_scene*);
int orc_get_grid(const orc_scene*, double* cell_aabb /* n*6 */, uint64_t* cell_offset /* n+1 */, uint32_t* prim_list);

/* Options.draw_boxes (`-o draw-boxes`, rtweekend.rs:92,128,146): every successful AABB test whose entry point lies on a box
 * edge sets the sticky rec.hits_aabb_edge (aabb.rs:57-60,90-97) and ray_color returns red (camera.rs:162-164,178-180) */
void orc_set_draw_boxes(orc_scene*, int on);

/* one HittableList::hit (hittable_list.rs:65-102) per ray on a fresh HitRecord.
 * out_prim = original object index of the last accepted primitive (-1 if none accepted),
 * out_t = rec.t bits after the call, out_tp = t at which rec.p was last written,
 * out_hit = return value (bit 0) | rec.hits_aabb_edge << 1 (only ever set with draw_boxes on). */
int orc_trace(const orc_scene*, int algorithm, const double* orig /* n*3 */, const double* dir /* n*3 */,
              uint64_t n, double tmin, double tmax,
              int32_t* out_prim, double* out_t, double* out_tp, uint8_t* out_hit, orc_counters* counters);

/* Camera::initialise camera.rs:88-128 */
void orc_camera_init(const orc_camera_in*, orc_camera* out);
/* primary rays for path (pixel i,j ; global sample s): camera.rs:132-148 with the counter RNG */
void orc_generate(const orc_camera*, uint64_t seed, uint32_t s, const uint32_t* pix_i, const uint32_t* pix_j,
                  uint64_t n, double* orig, double* dir);

/* Camera::render camera.rs:64-73 restricted to samples [s_begin, s_begin+s_count) and rows [row_begin,row_end):
 * out_sum[(j*W+i)*3+c] = sum over samples of ray_color (camera.rs:153-186).  nthreads>1 splits rows. */
int orc_render(const orc_scene*, const orc_camera*, int algorithm, uint32_t s_begin, uint32_t s_count,
               uint32_t max_depth, uint64_t seed, uint32_t row_begin, uint32_t row_end,
               double* out_sum, orc_counters* counters, int nthreads);

/* camera.rs:75 + color.rs:6-30: scale, gamma, clamp, quantise -> W*H*3 bytes */
void orc_resolve(const double* sum, uint64_t n_pixels, uint32_t spp_total, uint8_t* rgb_out);

/* the shared counter-based RNG (documented in DESIGN.md): draw `k` of stream (path_lo,path_hi,stream) */
/* raw Philox4x32-10 block (checked against the Random123 known-answer vectors in tests) */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_rng_draw(uint64_t seed, uint32_t path_lo, uint32_t path_hi, uint32_t stream, uint32_t k);

#ifdef __cplusplus
}
#endif
#endif
