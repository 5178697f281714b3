// host_api.cpp — extern "C" surface of librt_host.so, for language bindings (ctypes in raytracer_amd/).
// Plain pointers and sizes only; handles are freed by the matching *_free.
#include "driver.h"

#include <cstring>

using namespace rth;

namespace { thread_local std::string g_err; }

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }

// ---- scenes
void* rth_scene_load(const char* path, int format) {
    Scene* s = new Scene();
    bool ok;
    if (format == FORMAT_PLY) {
        rt_material m; memset(&m, 0, sizeof m); m.type = RT_LAMBERTIAN; m.albedo[0] = 0.8; m.albedo[1] = 0.2; m.albedo[2] = 0.1;   // main.rs:43
        ok = load_ply(path, m, *s, g_err);
    } else ok = load_json(path, *s, g_err);
    if (!ok) { delete s; return nullptr; }
    return s;
}
void* rth_scene_from_arrays(const rt_prim* prims, const double* centroids, uint32_t n, const rt_material* mats, uint32_t m) {
    Scene* s = new Scene();
    s->prims.assign(prims, prims + n); s->centroids.assign(centroids, centroids + 3 * (size_t)n); s->mats.assign(mats, mats + m);
    return s;
}
void rth_scene_free(void* h) { delete (Scene*)h; }
void rth_scene_subdivide(void* h, int levels) { subdivide_triangles(*(Scene*)h, levels); }
// append every object (and its material) of `other` to `h` (used for the mixed sphere/triangle config)
void rth_scene_append(void* h, const void* other) {
    Scene& a = *(Scene*)h; const Scene& b = *(const Scene*)other;
    uint32_t base = (uint32_t)a.mats.size();
    a.mats.insert(a.mats.end(), b.mats.begin(), b.mats.end());
    for (rt_prim p : b.prims) { p.material += base; a.prims.push_back(p); }
    a.centroids.insert(a.centroids.end(), b.centroids.begin(), b.centroids.end());
    if (a.vertex_lattice != b.vertex_lattice) a.vertex_lattice = 0.0;
}
uint32_t rth_scene_n_prims(const void* h) { return (uint32_t)((const Scene*)h)->prims.size(); }
uint32_t rth_scene_n_mats(const void* h) { return (uint32_t)((const Scene*)h)->mats.size(); }
const rt_prim* rth_scene_prims(const void* h) { return ((const Scene*)h)->prims.data(); }
const double* rth_scene_centroids(const void* h) { return ((const Scene*)h)->centroids.data(); }
const rt_material* rth_scene_mats(const void* h) { return ((const Scene*)h)->mats.data(); }

// ---- BVH
void* rth_bvh_build(const void* scene, int mode) {
    Bvh* b = new Bvh();
    if (!build_bvh(*(const Scene*)scene, (BvhMode)mode, *b, g_err)) { delete b; return nullptr; }
    return b;
}
// Bvh::new on the GPU (rt_bvh_build_mode): same Bvh object as the host builder returns
void* rth_bvh_build_gpu(const void* scene, int mode) {
    const Scene& sc = *(const Scene*)scene;
    Bvh* b = new Bvh();
    const uint32_t n = (uint32_t)sc.prims.size();
    b->nodes.resize(2 * (size_t)n + 1); b->order.resize(n);
    uint32_t n_nodes = 0, depth = 0;
    if (rt_bvh_build_mode(sc.prims.data(), sc.centroids.data(), n, mode, b->nodes.data(), &n_nodes, b->order.data(), &depth) != RT_OK) {
        g_err = rt_last_error(); delete b; return nullptr;
    }
    b->nodes.resize(n_nodes); b->depth = depth;
    return b;
}
void rth_bvh_free(void* h) { delete (Bvh*)h; }
uint32_t rth_bvh_n_nodes(const void* h) { return (uint32_t)((const Bvh*)h)->nodes.size(); }
uint32_t rth_bvh_depth(const void* h) { return ((const Bvh*)h)->depth; }
const rt_bvh_node* rth_bvh_nodes(const void* h) { return ((const Bvh*)h)->nodes.data(); }
const uint32_t* rth_bvh_order(const void* h) { return ((const Bvh*)h)->order.data(); }

// ---- grid
void* rth_grid_build(const void* scene, double grid_size) {
    Grid* g = new Grid();
    if (!build_grid(*(const Scene*)scene, grid_size, *g, g_err)) { delete g; return nullptr; }
    return g;
}
void rth_grid_free(void* h) { delete (Grid*)h; }
const rt_grid_desc* rth_grid_desc(const void* h) { return &((const Grid*)h)->desc; }
const double* rth_grid_cell_aabb(const void* h) { return ((const Grid*)h)->cell_aabb.data(); }
const uint64_t* rth_grid_cell_offset(const void* h) { return ((const Grid*)h)->cell_offset.data(); }
const uint32_t* rth_grid_prim_list(const void* h) { return ((const Grid*)h)->prim_list.data(); }
uint64_t rth_grid_list_len(const void* h) { return ((const Grid*)h)->prim_list.size(); }

// ---- camera / output
int rth_camera_preset(const char* filename, int camera_index, int width, rt_camera* out, uint32_t* spp, uint32_t* max_depth) {
    CameraIn c;
    if (!camera_preset(filename, camera_index, c, g_err)) return -1;
    if (width > 0) c.image_width = (uint32_t)width;
    camera_initialise(c, *out);
    if (spp) *spp = c.samples_per_pixel;
    if (max_depth) *max_depth = c.max_depth;
    return 0;
}
void rth_resolve(const double* sum, uint64_t n_pixels, uint32_t spp_total, uint8_t* rgb) { resolve_pixels(sum, n_pixels, spp_total, rgb); }
int rth_write_ppm(const char* path, uint32_t w, uint32_t h, const uint8_t* rgb) { return write_ppm_p3(path, w, h, rgb, g_err) ? 0 : -1; }

// ---- the whole run(), argv as the reference's binary takes it (argv[0] ignored)
int rth_run(int argc, const char* const* argv) {
    Cli cli;
    if (!parse_cli(argc, argv, cli, g_err)) return 2;
    // --gpus N forks its workers, and a forked copy of a process that has initialised HIP (any embedding host is likely to
    // have) cannot use the GPU: the flag belongs to the `raytracer` binary, which forks before its first HIP call
    if (cli.gpus > 1 || cli.reduce == 1) { g_err = "--gpus / --reduce rccl need the raytracer binary (the workers are forked before any HIP call); in-process hosts split samples with rt_render_device"; return 2; }
    return run(cli, nullptr, g_err);
}
// pieces of the CLI surface, exposed for tests
int rth_output_filename(int argc, const char* const* argv, char* out, uint32_t cap) {
    Cli cli;
    if (!parse_cli(argc, argv, cli, g_err)) return -1;
    std::string v = check_valid_options(cli.options);
    if (!v.empty()) { g_err = v; return -2; }
    std::string s = get_output_filename(cli);
    if (s.size() + 1 > cap) return -3;
    memcpy(out, s.c_str(), s.size() + 1);
    return 0;
}

} // extern "C"
