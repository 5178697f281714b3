// driver.cpp — command line, options, output naming, stats.csv and run() (reference: utils/rtweekend.rs,
// value/data.rs, main.rs).  The render itself is rt_render() from include/rt_hip.h; there is no CPU path.
#include "driver.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

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
    "      --gpu-build              Build the BVH (any -o mode) or the grid on the GPU instead of on the host (addition)\n"
    "      --binary-ppm             Write the image as binary PPM (P6) instead of P3 text (addition)\n"
    "      --device <N>             GPU index [default: 0] (addition)\n"
    "      --gpus <N>               Split the samples over N GPUs, one forked worker process each, partial sums added in rank\n"
    "                               order (addition; with an explicit --device every worker uses that GPU: rehearsal) [default: 1]\n"
    "      --reduce <HOW>           How --gpus sums the workers' framebuffers: rccl = one ncclReduce(sum) to worker 0 over xGMI,\n"
    "                               host = the parent adds partial sums from shared memory; auto = rccl when every worker has its own\n"
    "                               GPU [default: auto] [possible values: auto, rccl, host]\n"
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
        else if (key == "--device") { if (!need(i, "--device", v)) return false; out.device = atoi(v.c_str()); out.device_given = true; }
        else if (key == "--gpus") { if (!need(i, "--gpus", v)) return false; out.gpus = atoi(v.c_str()); if (out.gpus < 1 || out.gpus > 64) { err = "error: --gpus must be in 1..64"; return false; } }
        else if (key == "--reduce") { if (!need(i, "--reduce", v)) return false; if (v == "auto") out.reduce = 0; else if (v == "rccl") out.reduce = 1; else if (v == "host") out.reduce = 2; else { err = "error: invalid value '" + v + "' for '--reduce <HOW>' [possible values: auto, rccl, host]"; return false; } }
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

// dist.py's spp_range: global sample indices [begin, begin+count) of `rank`; remainders go to the lowest ranks
void spp_range(uint32_t rank, uint32_t world, uint32_t spp, uint32_t& begin, uint32_t& count) {
    const uint32_t base = spp / world, rem = spp % world;
    begin = rank * base + (rank < rem ? rank : rem);
    count = base + (rank < rem ? 1u : 0u);
}

namespace {
struct Accel { bool have_bvh = false, have_grid = false; Bvh bvh; Grid grid; };

// The device part of world.init() + cam.render() for samples [s_begin, s_begin + s_count) on one GPU: upload, flatten (or
// build on the GPU with --gpu-build), rt_render.  `sum` receives W*H*3 partial sums.
// With `comm_id` (rt_comm_unique_id bytes, RCCL path): the rank joins an n_ranks communicator and rt_render_reduce sums the
// framebuffers onto rank 0 with one ncclReduce; only rank 0's `sum` is written.
int render_slice(const Cli& cli, const Scene& scene, const Accel& acc, const rt_camera& rc, uint32_t max_depth, bool draw_boxes, int device,
                 uint32_t s_begin, uint32_t s_count, double* sum, rt_stats* st, double* init_seconds, std::string& err,
                 const void* comm_id = nullptr, int n_ranks = 1, int rank = 0) {
    if (rt_init(device) != RT_OK) { err = rt_last_error(); return 3; }
    // init_time covers world.init() (main.rs:126-129): upload + flatten / GPU build, not HIP start-up (context, code object)
    auto start = std::chrono::steady_clock::now();
    rt_scene* dev = rt_scene_create(scene.prims.data(), (uint32_t)scene.prims.size(), scene.mats.data(), (uint32_t)scene.mats.size());
    if (!dev) { err = rt_last_error(); return 3; }
    auto fail_hip = [&]() { err = rt_last_error(); rt_scene_destroy(dev); return 3; };
    if (rt_scene_set_options(dev, draw_boxes ? (uint32_t)RT_OPT_DRAW_BOXES : 0u) != RT_OK) return fail_hip();
    if (cli.algorithm == RT_ALGO_BVH) {
        if (acc.have_bvh) {
            if (rt_scene_set_bvh(dev, acc.bvh.nodes.data(), (uint32_t)acc.bvh.nodes.size(), acc.bvh.order.data(), (uint32_t)acc.bvh.order.size()) != RT_OK) return fail_hip();
        } else {                                         // --gpu-build (any build mode)
            auto t = std::chrono::steady_clock::now();
            Bvh bvh; const uint32_t n = (uint32_t)scene.prims.size();
            bvh.nodes.resize(2 * (size_t)n + 1); bvh.order.resize(n);
            uint32_t n_nodes = 0;
            if (rt_bvh_build_mode(scene.prims.data(), scene.centroids.data(), n, (int)bvh_mode_of(cli.options), bvh.nodes.data(), &n_nodes, bvh.order.data(), &bvh.depth) != RT_OK) return fail_hip();
            bvh.nodes.resize(n_nodes);
            if (rt_scene_set_bvh(dev, bvh.nodes.data(), n_nodes, bvh.order.data(), n) != RT_OK) return fail_hip();
            fprintf(stderr, "BVH constructed in %.2fs (%zu nodes, depth %u)\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(), bvh.nodes.size(), bvh.depth);
        }
    } else if (cli.algorithm == RT_ALGO_GRID) {
        if (acc.have_grid) {
            if (rt_scene_set_grid(dev, &acc.grid.desc, acc.grid.cell_aabb.data(), acc.grid.cell_offset.data(), acc.grid.prim_list.data()) != RT_OK) return fail_hip();
        } else {
            auto t = std::chrono::steady_clock::now();
            if (!(cli.grid_size >= 1.0)) { err = "grid size must be at least 1"; rt_scene_destroy(dev); return 1; }
            if (rt_scene_build_grid(dev, cli.grid_size) != RT_OK) return fail_hip();
            fprintf(stderr, "Grid constructed in %.2fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count());
        }
    }
    *init_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    if (comm_id) {
        rt_comm* comm = nullptr;
        if (rt_comm_create(comm_id, n_ranks, rank, &comm) != RT_OK) return fail_hip();
        if (rt_render_reduce(dev, &rc, cli.algorithm, s_begin, s_count, max_depth, cli.seed, comm, 0, rank == 0 ? sum : nullptr, st) != RT_OK) { err = rt_last_error(); rt_comm_destroy(comm); rt_scene_destroy(dev); return 3; }
        rt_comm_destroy(comm);
    } else if (rt_render(dev, &rc, cli.algorithm, s_begin, s_count, max_depth, cli.seed, sum, st) != RT_OK) return fail_hip();
    rt_scene_destroy(dev);
    return 0;
}
} // namespace

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

    RunStats rs;
    rs.filename = cli.filename; rs.primitives = scene.prims.size(); rs.algorithm = algorithm_variant(cli.algorithm);
    rs.options = options_to_string(cli); rs.image_width = rc.image_width; rs.image_height = rc.image_height;
    rs.samples_per_pixel = cam.samples_per_pixel; rs.max_depth = cam.max_depth;

    auto start = std::chrono::steady_clock::now();
    // world.init() main.rs:128 → hittable_list.rs:33-53, host part: the builders run once, before any worker exists
    Accel acc;
    if (cli.algorithm == RT_ALGO_BVH && !cli.gpu_build) {
        auto t = std::chrono::steady_clock::now();
        if (!build_bvh(scene, bvh_mode_of(cli.options), acc.bvh, err)) return 1;
        acc.have_bvh = true;
        fprintf(stderr, "BVH constructed in %.2fs (%zu nodes, depth %u)\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(), acc.bvh.nodes.size(), acc.bvh.depth);
    } else if (cli.algorithm == RT_ALGO_GRID && !cli.gpu_build) {
        auto t = std::chrono::steady_clock::now();
        if (!build_grid(scene, cli.grid_size, acc.grid, err)) return 1;
        acc.have_grid = true;
        fprintf(stderr, "Grid constructed in %.2fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count());
    }
    const double host_init = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();

    // cam.render() main.rs:132
    const uint64_t n_pixels = (uint64_t)rc.image_width * rc.image_height;
    std::vector<double> sum(n_pixels * 3);
    rt_stats st; memset(&st, 0, sizeof st);
    // --reduce: rccl needs one GPU per worker; an explicit --device puts every worker on that GPU (rehearsal) -> host sums
    const bool distinct_gpus = !(cli.device_given && cli.gpus > 1);
    if (cli.reduce == 1 && !distinct_gpus) { err = "--reduce rccl needs one GPU per worker: a communicator cannot hold the same GPU twice (drop --device, or use --reduce host)"; return 2; }
    const bool use_rccl = cli.reduce == 1 || (cli.reduce == 0 && cli.gpus > 1 && distinct_gpus);
    if (cli.gpus <= 1 && !use_rccl) {
        double dev_init = 0;
        const int rcode = render_slice(cli, scene, acc, rc, cam.max_depth, draw_boxes, cli.device, 0, cam.samples_per_pixel, sum.data(), &st, &dev_init, err);
        if (rcode) return rcode;
        rs.init_time = host_init + dev_init;
    } else {
        // SURVEY 8(e) without a launcher: one worker process per GPU, forked BEFORE anything touches HIP (a forked copy of a
        // process that has initialised the GPU is not usable), each rendering its own global sample indices of every pixel.
        //   rccl: worker 0 draws the communicator id and publishes it in the shared mapping; every worker joins, renders into its
        //         device framebuffer and takes part in ONE ncclReduce(sum, root 0) (rt_render_reduce); worker 0 writes the sum.
        //   host: every worker writes its partial sums into the mapping; the parent adds them in rank order.
        const uint32_t N = (uint32_t)cli.gpus;
        struct Slot { int rc; double dev_init; rt_stats st; char err[480]; };
        struct Shared { volatile int id_ready; char id[RT_COMM_ID_BYTES]; };
        const size_t fb_bytes = (size_t)n_pixels * 3 * sizeof(double);
        const size_t n_fb = use_rccl ? 1 : N;
        const size_t total = n_fb * fb_bytes + (size_t)N * sizeof(Slot) + sizeof(Shared);
        void* shm = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
        if (shm == MAP_FAILED) { err = "mmap of the shared framebuffers failed"; return 1; }
        Slot* slots = (Slot*)((char*)shm + n_fb * fb_bytes);
        Shared* shared = (Shared*)((char*)slots + (size_t)N * sizeof(Slot));
        fflush(stdout); fflush(stderr);
        std::vector<pid_t> pids(N, -1);
        for (uint32_t g = 0; g < N; ++g) {
            uint32_t sb, scount; spp_range(g, N, cam.samples_per_pixel, sb, scount);
            pid_t pid = fork();
            if (pid < 0) { err = "fork failed"; for (uint32_t k = 0; k < g; ++k) { kill(pids[k], SIGKILL); waitpid(pids[k], nullptr, 0); } munmap(shm, total); return 1; }
            if (pid == 0) {
                std::string werr; Slot& sl = slots[g];
                const int dev = cli.device_given ? cli.device : (int)g;
                if (use_rccl) {
                    sl.rc = 0;
                    if (g == 0) {                              // rank 0 draws the id after ITS rt_init (nothing touched HIP before the fork)
                        if (rt_init(dev) != RT_OK || rt_comm_unique_id(shared->id) != RT_OK) { werr = rt_last_error(); sl.rc = 3; }
                        __sync_synchronize(); shared->id_ready = sl.rc ? -1 : 1;
                    } else {
                        while (shared->id_ready == 0) usleep(1000);
                        __sync_synchronize();
                        if (shared->id_ready < 0) { werr = "rank 0 could not create the communicator id"; sl.rc = 3; }
                    }
                    if (!sl.rc) sl.rc = render_slice(cli, scene, acc, rc, cam.max_depth, draw_boxes, dev, sb, scount, (double*)shm, &sl.st, &sl.dev_init, werr, shared->id, (int)N, (int)g);
                } else {
                    sl.rc = render_slice(cli, scene, acc, rc, cam.max_depth, draw_boxes, dev, sb, scount,
                                         (double*)((char*)shm + (size_t)g * fb_bytes), &sl.st, &sl.dev_init, werr);
                }
                snprintf(sl.err, sizeof sl.err, "%s", werr.c_str());
                fflush(stderr);
                _exit(sl.rc ? 1 : 0);                 // no static destructors of the parent's copy
            }
            pids[g] = pid;
        }
        // wait for all workers; the first failure ends the others (a rank that died would leave its peers waiting in the
        // communicator's rendezvous or in the reduce for ever)
        int bad = 0; uint32_t left = N;
        std::vector<bool> done(N, false);
        while (left) {
            bool progressed = false;
            for (uint32_t g = 0; g < N; ++g) {
                if (done[g]) continue;
                int status = 0;
                const pid_t w = waitpid(pids[g], &status, WNOHANG);
                if (w == 0) continue;
                done[g] = true; --left; progressed = true;
                if (w < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) {
                    if (!bad) err = "worker " + std::to_string(g) + ": " + ((w > 0 && WIFEXITED(status) && slots[g].err[0]) ? std::string(slots[g].err) : std::string("ended abnormally"));
                    bad = 3;
                    for (uint32_t k = 0; k < N; ++k) if (!done[k]) kill(pids[k], SIGKILL);
                }
            }
            if (!progressed) usleep(2000);
        }
        if (bad) { munmap(shm, total); return bad; }
        double dev_init = 0;
        for (uint32_t g = 0; g < N; ++g) {
            if (use_rccl) { if (g == 0) memcpy(sum.data(), shm, fb_bytes); }        // already summed on the GPUs, rank 0 wrote it
            else {
                const double* part = (const double*)((const char*)shm + (size_t)g * fb_bytes);
                if (g == 0) memcpy(sum.data(), part, fb_bytes);
                else for (size_t i = 0; i < sum.size(); ++i) sum[i] = sum[i] + part[i];
            }
            const rt_stats& w = slots[g].st;
            st.primary_rays += w.primary_rays; st.scatter_rays += w.scatter_rays; st.intersection_checks += w.intersection_checks;
            st.traversal_steps += w.traversal_steps; st.overlapping_aabb += w.overlapping_aabb; st.gridbox_intersection_checks += w.gridbox_intersection_checks;
            if (w.render_ms > st.render_ms) st.render_ms = w.render_ms;
            if (slots[g].dev_init > dev_init) dev_init = slots[g].dev_init;
        }
        munmap(shm, total);
        rs.init_time = host_init + dev_init;
        fprintf(stderr, "%u worker%s, framebuffers summed %s\n", N, N == 1 ? "" : "s", use_rccl ? "with one RCCL ncclReduce (sum, root 0)" : "by the parent from shared memory");
    }
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
