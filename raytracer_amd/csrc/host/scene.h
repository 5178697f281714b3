// scene.h — host-side scene model, loaders and acceleration-structure builders.
//
// This is the C++ restatement of the reference's host surface for the ray_color path (the reference is
// Rust and no Rust toolchain exists in the build image): scene JSON / PLY loading, the SAH BVH and uniform
// grid builders whose *layout* the device consumes, camera presets, options, output naming and stats.csv.
// It never calls the CPU oracle and contains no CPU renderer: rendering goes through include/rt_hip.h only.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/rt_hip.h"

namespace rth {

struct Scene {
    std::vector<rt_prim> prims;          // file order == HittableList.objects (hittable_list.rs:16)
    std::vector<double> centroids;       // Hittable::centroid() per object, xyz interleaved
    std::vector<rt_material> mats;
    std::string filename;                // as given on the command line (camera presets key on its prefix)
    double vertex_lattice = 0.0;         // PLY: every coordinate is (f64)f32 * 100 (parser.rs:67-73) -> 100; 0 = arbitrary f64
};

// serde_json::from_reader into HittableList (main.rs:36-41; typetag tags hittable.rs:32, material.rs:15)
bool load_json(const std::string& path, Scene& out, std::string& err);
// parse_ply (utils/parser.rs:15-81): f32 vertices widened and scaled by 100, 3-gon faces, one shared material
bool load_ply(const std::string& path, const rt_material& mat, Scene& out, std::string& err);
// Synthetic refinement (NOT a reference feature; used only to stand in for the absent full-resolution
// dragon): every triangle is split into 4 by edge midpoints, `levels` times.  Spheres are kept.  For a PLY mesh the
// midpoints are rounded onto the file format's own lattice ((f64)f32 * 100), so the result is what a finer PLY file
// would have produced — a mesh whose coordinates a PLY could not hold would not stand in for one.
void subdivide_triangles(Scene& sc, int levels);

enum BvhMode { BVH_SAH_POSITION = 0, BVH_SAH_PLANE = 1, BVH_NAIVE = 2 };   // rtweekend.rs:137-147

struct Bvh {
    std::vector<rt_bvh_node> nodes;      // indices 0..=node_pointer (bvh.rs:14-15)
    std::vector<uint32_t> order;         // Bvh.objects as file-order indices (bvh.rs:13)
    uint32_t depth = 0;
};
// Bvh::new (bvh.rs:19-26) with an exact-equivalent O(n log n)-per-node SAH sweep (prefix/suffix AABBs)
// in place of the reference's O(n^2) folds (bvh.rs:125-133,150-151).
bool build_bvh(const Scene& sc, BvhMode mode, Bvh& out, std::string& err);

struct Grid {
    rt_grid_desc desc;
    std::vector<double> cell_aabb;       // n_cells * 6
    std::vector<uint64_t> cell_offset;   // n_cells + 1
    std::vector<uint32_t> prim_list;
};
// Grid::new (grid.rs:23-61) with the fixed bounds of hittable_list.rs:43; objects are binned by
// rasterising their AABB instead of testing every cell against every object (same lists, same order).
bool build_grid(const Scene& sc, double grid_size, Grid& out, std::string& err);

// Camera (camera.rs:14-24) + the presets run() hard-codes (main.rs:72-104)
struct CameraIn {
    double aspect_ratio = 1.0; uint32_t image_width = 100; uint32_t samples_per_pixel = 10; uint32_t max_depth = 10;
    double vfov = 90.0; double look_from[3] = {0, 0, 0}; double look_at[3] = {0, 0, -1}; double v_up[3] = {0, 1, 0};
    double defocus_angle = 0.0; double focus_dist = 10.0;
};
bool camera_preset(const std::string& filename, int camera_index /* -1 = none */, CameraIn& cam, std::string& err);
void camera_initialise(const CameraIn& in, rt_camera& out);     // camera.rs:88-128

// camera.rs:75 + color.rs:6-30 → bytes; and the P3 text the reference writes (camera.rs:61,75-76)
void resolve_pixels(const double* sum, uint64_t n_pixels, uint32_t spp_total, uint8_t* rgb);
bool write_ppm_p3(const std::string& path, uint32_t w, uint32_t h, const uint8_t* rgb, std::string& err);
bool write_ppm_p6(const std::string& path, uint32_t w, uint32_t h, const uint8_t* rgb, std::string& err);   // addition: binary

} // namespace rth
