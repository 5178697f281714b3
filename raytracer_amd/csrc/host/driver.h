// driver.h — the reference's command-line surface (utils/rtweekend.rs:8-197) and run() (main.rs:31-140),
// with cam.render()/world.init() replaced by calls into include/rt_hip.h.
#pragma once
#include <string>
#include <vector>

#include "scene.h"

namespace rth {

enum FileFormat { FORMAT_NATIVE = 0, FORMAT_PLY = 1 };                      // rtweekend.rs:81-86
enum AlgorithmOption { OPT_BVH_NAIVE = 0, OPT_BVH_SAH_PLANE, OPT_BVH_SAH_POSITION, OPT_DRAW_BOXES };   // rtweekend.rs:137-147

struct Cli {                                                                // rtweekend.rs:8-30
    std::string filename; bool has_filename = false;
    FileFormat format = FORMAT_NATIVE;
    int algorithm = RT_ALGO_BVH;
    std::vector<AlgorithmOption> options;
    double grid_size = 25.0;
    bool stats = false;
    int camera = -1;                    // Option<usize>: -1 = None
    // additions (the reference hard-codes these in main.rs:72-78; BASELINE configs need them as flags)
    int width = -1, spp = -1, max_depth = -1;
    uint64_t seed = 1;
    int device = 0; bool device_given = false;
    int gpus = 1;                       // --gpus N: spp-split over N forked workers, one GPU each; CLI binary only
    int reduce = 0;                     // --reduce auto|rccl|host (0/1/2): how the workers' framebuffers are summed.  rccl = one ncclReduce(sum, root 0)
                                        // over xGMI (rt_render_reduce); host = partial sums added by the parent through shared memory.  auto = rccl when
                                        // N > 1 and every worker has a GPU of its own, host otherwise (workers sharing one GPU cannot form a communicator)
    int subdivide = 0;                  // synthetic refinement levels for PLY meshes (not a reference feature)
    std::string output_dir = "output";
    bool gpu_build = false;             // --gpu-build: Bvh::new (any build mode) / Grid::new on the GPU
    bool binary_ppm = false;            // --binary-ppm: write P6 instead of the reference's P3 text (addition)
};

// clap-equivalent parsing; returns false and sets err on a bad command line (help text in err for --help)
bool parse_cli(int argc, const char* const* argv, Cli& out, std::string& err);
// check_valid_options rtweekend.rs:150-160: empty string when valid
std::string check_valid_options(const std::vector<AlgorithmOption>& options);
// Display for Options rtweekend.rs:97-113
std::string options_to_string(const Cli& cli);
// get_output_filename rtweekend.rs:175-197
std::string get_output_filename(const Cli& cli);
BvhMode bvh_mode_of(const std::vector<AlgorithmOption>& options);
// raytracer_amd/dist.py spp_range: the global sample indices of one rank
void spp_range(uint32_t rank, uint32_t world, uint32_t spp, uint32_t& begin, uint32_t& count);

struct RunStats {                       // value/data.rs:6-24, in CSV column order
    double seconds = 0, init_time = 0;
    uint64_t primary_rays = 0, scatter_rays = 0, intersection_checks = 0, traversal_steps = 0, overlapping_aabb = 0,
             gridbox_intersection_checks = 0;
    std::string filename; uint64_t primitives = 0; std::string algorithm, options;
    uint32_t image_width = 0, image_height = 0, samples_per_pixel = 0, max_depth = 0;
};
bool write_stats_csv(const std::string& path, const RunStats& s, std::string& err);   // data.rs:54-70
std::string format_f64(double v);       // shortest round-trip, Rust-style ("12.0", "0.001")

// run() main.rs:31-140.  Returns 0 on success; on failure returns non-zero with the message in err
// (where the reference would panic!/expect()).
int run(const Cli& cli, RunStats* stats_out, std::string& err);

} // namespace rth
