// driver.cpp — command line, options, output naming, stats.csv and run() (reference: utils/rtweekend.rs,
// value/data.rs, main.rs).  The render itself is rt_render() from include/rt_hip.h; there is no CPU path.
#include "driver.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/stat.h>

namespace rth {

static const char* kHelp =
    "Usage: raytracer [OPTIONS] [FILENAME]\n\n"
    "Arguments:\n  [FILENAME]  The world / scene file\n\n"
    "Options:\n"
    "      --format <FORMAT>        The input file format [default: native] [possible values: native, ply]\n"
    "      --algorithm <ALGORITHM>  The intersection algorithm [default: bvh] [possible values: naive, bvh, grid]\n"
    "  -o, --options <OPTIONS>      Options for the algorithm [possible values: bvh-naive, bvh-sah-plane, bvh-sah-position, draw-boxes]\n"
    "  -g, --grid-size <GRID_SIZE>  Grid size (if algorithm is grid) [default: 25]\n"
    "      --stats                  Print scene statistics (as LaTeX table row) and exit\n"
    "      --camera <CAMERA>        Camera position (only for dragon scene)\n"
    "      --width <W>              Image width [default: 900]   (addition; main.rs:74 hard-codes it)\n"
    "      --spp <N>                Samples per pixel [default: 50] (addition; main.rs:75)\n"
    "      --max-depth <N>          Bounce limit [default: 50]   (addition; main.rs:76)\n"
    "      --seed <N>               Seed of the counter RNG [default: 1] (addition)\n"
    "      --gpu-build              Build the (default, SAH position) BVH on the GPU instead of on the host (addition)\n"
    "      --binary-ppm             Write the image as binary PPM (P6) instead of P3 text (addition)\n"
    "      --device <N>             GPU index [default: 0] (addition)\n"
    "      --subdivide <L>          Split every PLY triangle into 4, L times (addition; synthetic meshes)\n"
    "      --output-dir <DIR>       Where the .ppm and stats.csv go [default: output]\n"
    "  -h, --help                   Print help\n";

bool parse_cli(int argc, const char* const* argv, Cli& out, std::string& err) {
    out = Cli();
    auto need = [&](int& i, const char* flag, std::string& v) {
        std::string a = argv[i];
        size_t eq = a.find('=');
        if (a.rfind("--", 0) == 0 && eq != std::string::npos) { v = a.substr(eq + 1); return true; }
        if (i + 1 >= argc) { err = std::string("error: a value is required for '") + flag + "' but none was supplied"; return false; }
        v = argv[++i]; return true;
    };
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i], key = a.substr(0, a.find('=')), v;
        if (a == "-h" || a == "--help") { err = kHelp; return false; }
        else if (key == "--format") { if (!need(i, "--format", v)) return false; if (v == "native") out.format = FORMAT_NATIVE; else if (v == "ply") out.format = FORMAT_PLY; else { err = "error: invalid value '" + v + "' for '--format <FORMAT>' [possible values: native, ply]"; return false; } }
        else if (key == "--algorithm") { if (!need(i, "--algorithm", v)) return false; if (v == "naive") out.algorithm = RT_ALGO_NAIVE; else if (v == "bvh") out.algorithm = RT_ALGO_BVH; else if (v == "grid") out.algorithm = RT_ALGO_GRID; else { err = "error: invalid value '" + v + "' for '--algorithm <ALGORITHM>' [possible values: naive, bvh, grid]"; return false; } }
        else if (key == "-o" || key == "--options") {
            if (!need(i, "--options", v)) return false;
            if (v == "bvh-naive") out.options.push_back(OPT_BVH_NAIVE); else if (v == "bvh-sah-plane") out.options.push_back(OPT_BVH_SAH_PLANE);
            else if (v == "bvh-sah-position") out.options.push_back(OPT_BVH_SAH_POSITION); else if (v == "draw-boxes") out.options.push_back(OPT_DRAW_BOXES);
            else { err = "error: invalid value '" + v + "' for '--options <OPTIONS>' [possible values: bvh-naive, bvh-sah-plane, bvh-sah-position, draw-boxes]"; return false; }
        }
        else if (key == "-g" || key == "--grid-size") { if (!need(i, "--grid-size", v)) return false; char* e; out.grid_size = strtod(v.c_str(), &e); if (e == v.c_str() || *e) { err = "error: invalid value '" + v + "' for '--grid-size <GRID_SIZE>'"; return false; } }
        else if (a == "--stats") out.stats = true;
        else if (key == "--camera") { if (!need(i, "--camera", v)) return false; char* e; long c = strtol(v.c_str(), &e, 10); if (e == v.c_str() || *e || c < 0) { err = "error: invalid value '" + v + "' for '--camera <CAMERA>'"; return false; } out.camera = (int)c; }
        else if (key == "--width") { if (!need(i, "--width", v)) return false; out.width = atoi(v.c_str()); if (out.width < 1) { err = "error: --width must be >= 1"; return false; } }
        else if (key == "--spp") { if (!need(i, "--spp", v)) return false; out.spp = atoi(v.c_str()); if (out.spp < 1) { err = "error: --spp must be >= 1"; return false; } }
        else if (key == "--max-depth") { if (!need(i, "--max-depth", v)) return false; out.max_depth = atoi(v.c_str()); if (out.max_depth < 0) { err = "error: --max-depth must be >= 0"; return false; } }
        else if (key == "--seed") { if (!need(i, "--seed", v)) return false; out.seed = strtoull(v.c_str(), nullptr, 10); }
        else if (key == "--device") { if (!need(i, "--device", v)) return false; out.device = atoi(v.c_str()); }
        else if (key == "--subdivide") { if (!need(i, "--subdivide", v)) return false; out.subdivide = atoi(v.c_str()); if (out.subdivide < 0 || out.subdivide > 4) { err = "error: --subdivide must be in 0..4"; return false; } }
        else if (key == "--binary-ppm") out.binary_ppm = true;
        else if (key == "--gpu-build") out.gpu_build = true;
        else if (key == "--output-dir") { if (!need(i, "--output-dir", v)) return false; out.output_dir = v; }
        else if (!a.empty() && a[0] == '-' && a != "-") { err = "error: unexpected argument '" + a + "' found"; return false; }
        else if (!out.has_filename) { out.filename = a; out.has_filename = true; }
        else { err = "error: unexpected argument '" + a + "' found"; return false; }
    }
    return true;
}

static const char* option_debug_name(AlgorithmOption o) {
    switch (o) { case OPT_BVH_NAIVE: return "BvhNaive"; case OPT_BVH_SAH_PLANE: return "BvhSahPlane"; case OPT_BVH_SAH_POSITION: return "BvhSahPosition"; default: return "DrawBoxes"; }
}

std::string check_valid_options(const std::vector<AlgorithmOption>& options) {   // rtweekend.rs:150-160
    std::string list; int n = 0;
    for (AlgorithmOption o : options) if (o != OPT_DRAW_BOXES) { if (n++) list += ", "; list += option_debug_name(o); }
    if (n > 1) return "Can't have the following options at the same time: [" + list + "]";
    return "";
}

BvhMode bvh_mode_of(const std::vector<AlgorithmOption>& options) {            // bvh.rs:87,114: contains() checks
    for (AlgorithmOption o : options) if (o == OPT_BVH_NAIVE) return BVH_NAIVE;
    for (AlgorithmOption o : options) if (o == OPT_BVH_SAH_PLANE) return BVH_SAH_PLANE;
    return BVH_SAH_POSITION;
}

static std::string display_f64(double v) {               // Rust `{}` on f64: shortest round-trip, no forced ".0"
    char buf[64];
    for (int p = 1; p <= 17; ++p) { snprintf(buf, sizeof buf, "%.*g", p, v); if (strtod(buf, nullptr) == v) break; }
    return buf;
}
std::string format_f64(double v) {                       // csv/serde (ryu): integral values keep ".0"
    std::string s = display_f64(v);
    if (s.find_first_of(".en") == std::string::npos) s += ".0";
    return s;
}

std::string options_to_string(const Cli& cli) {          // rtweekend.rs:97-113
    std::vector<std::string> parts;
    for (AlgorithmOption o : cli.options) parts.push_back(option_debug_name(o));
    if (cli.algorithm == RT_ALGO_GRID) parts.push_back("size=" + display_f64(cli.grid_size));
    if (cli.camera >= 0) parts.push_back("pos" + std::to_string(cli.camera));
    std::string j;
    for (size_t i = 0; i < parts.size(); ++i) { if (i) j += "_"; j += parts[i]; }
    return j;
}

static const char* algorithm_display(int a) { return a == RT_ALGO_NAIVE ? "naive" : (a == RT_ALGO_BVH ? "bvh" : "grid"); }   // rtweekend.rs:64-78
static const char* algorithm_variant(int a) { return a == RT_ALGO_NAIVE ? "Naive" : (a == RT_ALGO_BVH ? "BVH" : "Grid"); }   // derive(Serialize)

std::string get_output_filename(const Cli& cli) {        // rtweekend.rs:175-197
    std::string path = cli.filename;
    size_t slash = path.find_last_of('/');
    std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    size_t dot = name.find_last_of('.');
    std::string stem = (dot == std::string::npos || dot == 0) ? name : name.substr(0, dot);   // Path::file_stem
    std::string opt = options_to_string(cli);
    if (!opt.empty()) opt = "-" + opt;
    return cli.output_dir + "/" + stem + "-" + algorithm_display(cli.algorithm) + opt + ".ppm";
}

static std::string csv_field(const std::string& s) {
    if (s.find_first_of(",\"\n\r") == std::string::npos) return s;
    std::string q = "\"";
    for (char c : s) { if (c == '"') q += '"'; q += c; }
    return q + "\"";
}

bool write_stats_csv(const std::string& path, const RunStats& s, std::string& err) {   // data.rs:54-70
    struct stat sb; bool exists = stat(path.c_str(), &sb) == 0;
    FILE* f = fopen(path.c_str(), "ab");
    if (!f) { err = "Cannot open \"" + path + "\""; return false; }
    if (!exists)
        fputs("seconds,init_time,primary_rays,scatter_rays,intersection_checks,traversal_steps,overlapping_aabb,"
              "gridbox_intersection_checks,filename,primitives,algorithm,options,image_width,image_height,samples_per_pixel,max_depth\n", f);
    fprintf(f, "%s,%s,%llu,%llu,%llu,%llu,%llu,%llu,%s,%llu,%s,%s,%u,%u,%u,%u\n", format_f64(s.seconds).c_str(), format_f64(s.init_time).c_str(),
            (unsigned long long)s.primary_rays, (unsigned long long)s.scatter_rays, (unsigned long long)s.intersection_checks,
            (unsigned long long)s.traversal_steps, (unsigned long long)s.overlapping_aabb, (unsigned long long)s.gridbox_intersection_checks,
            csv_field(s.filename).c_str(), (unsigned long long)s.primitives, s.algorithm.c_str(), csv_field(s.options).c_str(),
            s.image_width, s.image_height, s.samples_per_pixel, s.max_depth);
    bool ok = fclose(f) == 0;
    if (!ok) err = "Failed to write CSV data";
    return ok;
}

int run(const Cli& cli, RunStats* stats_out, std::string& err) {
    std::string v = check_valid_options(cli.options);    // main.rs:25-27
    if (!v.empty()) { err = v; return 2; }
    if (!cli.has_filename) {
        // main.rs:48-67 renders a scene generated in code and writes it to scenes/; the generators (utils/scenes.rs)
        // are outside the accelerated path: every scene they produce ships as a JSON file.
        err = "no scene file given: pass one of scenes/*.json or a PLY file (the built-in generators are not part of this build)";
        return 2;
    }
    bool draw_boxes = false;                             // Options::new rtweekend.rs:128
    for (AlgorithmOption o : cli.options) if (o == OPT_DRAW_BOXES) draw_boxes = true;

    Scene scene;
    if (cli.format == FORMAT_NATIVE) { if (!load_json(cli.filename, scene, err)) return 1; }
    else {
        rt_material m; memset(&m, 0, sizeof m); m.type = RT_LAMBERTIAN; m.albedo[0] = 0.8; m.albedo[1] = 0.2; m.albedo[2] = 0.1;   // main.rs:43
        if (!load_ply(cli.filename, m, scene, err)) return 1;
        if (cli.subdivide > 0) subdivide_triangles(scene, cli.subdivide);
    }
    scene.filename = cli.filename;

    CameraIn cam;
    if (!camera_preset(cli.filename, cli.camera, cam, err)) return 2;
    if (cli.width > 0) cam.image_width = (uint32_t)cli.width;
    if (cli.spp > 0) cam.samples_per_pixel = (uint32_t)cli.spp;
    if (cli.max_depth >= 0) cam.max_depth = (uint32_t)cli.max_depth;

    if (cli.stats) {                                     // main.rs:107-116
        size_t nl = 0, nm = 0, nd = 0;
        for (const rt_prim& p : scene.prims) { uint32_t t = scene.mats[p.material].type; if (t == RT_LAMBERTIAN) ++nl; else if (t == RT_METAL) ++nm; else ++nd; }
        printf("Name & \\# Primitives & \\# Lambertian primitives & \\# Metal primitives & \\# Dieelectric primitives \\\\\n");
        printf("%s & %zu & %zu & %zu & %zu\\\\\n", cli.filename.c_str(), scene.prims.size(), nl, nm, nd);
        return 0;
    }

    rt_camera rc; camera_initialise(cam, rc);
    const std::string out_filename = get_output_filename(cli);
    mkdir(cli.output_dir.c_str(), 0777);

    if (rt_init(cli.device) != RT_OK) { err = rt_last_error(); return 3; }
    RunStats rs;
    rs.filename = cli.filename; rs.primitives = scene.prims.size(); rs.algorithm = algorithm_variant(cli.algorithm);
    rs.options = options_to_string(cli); rs.image_width = rc.image_width; rs.image_height = rc.image_height;
    rs.samples_per_pixel = cam.samples_per_pixel; rs.max_depth = cam.max_depth;

    auto start = std::chrono::steady_clock::now();
    // world.init() main.rs:128 → hittable_list.rs:33-53 (host build, then flatten + upload)
    rt_scene* dev = rt_scene_create(scene.prims.data(), (uint32_t)scene.prims.size(), scene.mats.data(), (uint32_t)scene.mats.size());
    if (!dev) { err = rt_last_error(); return 3; }
    if (rt_scene_set_options(dev, draw_boxes ? (uint32_t)RT_OPT_DRAW_BOXES : 0u) != RT_OK) { err = rt_last_error(); rt_scene_destroy(dev); return 3; }
    if (cli.algorithm == RT_ALGO_BVH) {
        auto t = std::chrono::steady_clock::now();
        Bvh bvh;
        if (cli.gpu_build && bvh_mode_of(cli.options) == BVH_SAH_POSITION) {
            const uint32_t n = (uint32_t)scene.prims.size();
            bvh.nodes.resize(2 * (size_t)n + 1); bvh.order.resize(n);
            uint32_t n_nodes = 0;
            if (rt_bvh_build(scene.prims.data(), scene.centroids.data(), n, bvh.nodes.data(), &n_nodes, bvh.order.data(), &bvh.depth) != RT_OK) { err = rt_last_error(); rt_scene_destroy(dev); return 3; }
            bvh.nodes.resize(n_nodes);
        } else if (!build_bvh(scene, bvh_mode_of(cli.options), bvh, err)) { rt_scene_destroy(dev); return 1; }
        if (rt_scene_set_bvh(dev, bvh.nodes.data(), (uint32_t)bvh.nodes.size(), bvh.order.data(), (uint32_t)bvh.order.size()) != RT_OK) { err = rt_last_error(); rt_scene_destroy(dev); return 3; }
        fprintf(stderr, "BVH constructed in %.2fs (%zu nodes, depth %u)\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(), bvh.nodes.size(), bvh.depth);
    } else if (cli.algorithm == RT_ALGO_GRID) {
        auto t = std::chrono::steady_clock::now();
        if (cli.gpu_build) {
            if (!(cli.grid_size >= 1.0)) { err = "grid size must be at least 1"; rt_scene_destroy(dev); return 1; }
            if (rt_scene_build_grid(dev, cli.grid_size) != RT_OK) { err = rt_last_error(); rt_scene_destroy(dev); return 3; }
        } else {
            Grid grid;
            if (!build_grid(scene, cli.grid_size, grid, err)) { rt_scene_destroy(dev); return 1; }
            if (rt_scene_set_grid(dev, &grid.desc, grid.cell_aabb.data(), grid.cell_offset.data(), grid.prim_list.data()) != RT_OK) { err = rt_last_error(); rt_scene_destroy(dev); return 3; }
        }
        fprintf(stderr, "Grid constructed in %.2fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count());
    }
    rs.init_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();

    // cam.render() main.rs:132
    const uint64_t n_pixels = (uint64_t)rc.image_width * rc.image_height;
    std::vector<double> sum(n_pixels * 3);
    rt_stats st;
    if (rt_render(dev, &rc, cli.algorithm, 0, cam.samples_per_pixel, cam.max_depth, cli.seed, sum.data(), &st) != RT_OK) { err = rt_last_error(); rt_scene_destroy(dev); return 3; }
    rt_scene_destroy(dev);
    std::vector<uint8_t> rgb(n_pixels * 3);
    resolve_pixels(sum.data(), n_pixels, cam.samples_per_pixel, rgb.data());
    if (!(cli.binary_ppm ? write_ppm_p6 : write_ppm_p3)(out_filename, rc.image_width, rc.image_height, rgb.data(), err)) return 1;
    rs.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    rs.primary_rays = st.primary_rays; rs.scatter_rays = st.scatter_rays; rs.intersection_checks = st.intersection_checks;
    rs.traversal_steps = st.traversal_steps; rs.overlapping_aabb = st.overlapping_aabb; rs.gridbox_intersection_checks = st.gridbox_intersection_checks;

    // data.print() data.rs:43-52
    printf("Total primary rays: %llu\n", (unsigned long long)rs.primary_rays);
    printf("Total scatter rays: %llu\n", (unsigned long long)rs.scatter_rays);
    printf("Overlapping AABBs: %llu\n", (unsigned long long)rs.overlapping_aabb);
    printf("Total intersection checks: %llu\n", (unsigned long long)rs.intersection_checks);
    printf("Total gridbox intersection checks: %llu\n", (unsigned long long)rs.gridbox_intersection_checks);
    printf("Total traversal steps: %llu\n", (unsigned long long)rs.traversal_steps);
    printf("Total init time: %s\n", display_f64(rs.init_time).c_str());
    printf("Total time: %s\n", display_f64(rs.seconds).c_str());
    if (!write_stats_csv(cli.output_dir + "/stats.csv", rs, err)) return 1;
    fprintf(stderr, "Wrote image to %s. Duration %.2fs (device render %.1f ms, %.1f Mrays/s)\n", out_filename.c_str(), rs.seconds, st.render_ms,
            st.render_ms > 0 ? (double)(st.primary_rays + st.scatter_rays) / st.render_ms / 1e3 : 0.0);
    if (stats_out) *stats_out = rs;
    return 0;
}

} // namespace rth
