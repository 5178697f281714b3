// host_sanitize.cpp — the host library's loaders and builders (scene.cpp, accel.cpp: threaded PLY parse, threaded SAH build,
// grid binning, subdivision, camera, resolve, PPM writers) run under the host sanitizers.  Not part of the product; built and run by
// tests/test_host_sanitize.py as
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all ... host_sanitize.cpp scene.cpp accel.cpp
//   g++ -O1 -g -fsanitize=thread ...                                      (the helper threads of the builders)
// Prints one line per case; any sanitizer report makes the process exit non-zero.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../raytracer_amd/csrc/host/scene.h"

using namespace rth;

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); ++fails; } } while (0)

static void build_all(const Scene& sc, const char* label, bool fine_grid) {
    std::string err;
    for (int m = 0; m < 3; ++m) {
        Bvh b;
        CHECK(build_bvh(sc, (BvhMode)m, b, err), "%s: build_bvh mode %d: %s", label, m, err.c_str());
        CHECK(b.order.size() == sc.prims.size() && !b.nodes.empty() && b.nodes.size() <= 2 * sc.prims.size(), "%s: node / order sizes", label);
        std::vector<char> seen(sc.prims.size(), 0);
        for (uint32_t o : b.order) { CHECK(o < sc.prims.size() && !seen[o], "%s: order is not a permutation", label); if (o < sc.prims.size()) seen[o] = 1; }
        size_t leaves = 0, covered = 0;
        for (const rt_bvh_node& n : b.nodes) {
            if (n.is_leaf) { ++leaves; covered += n.count; CHECK((size_t)n.first + n.count <= sc.prims.size(), "%s: leaf range", label); }
            else CHECK(n.left < b.nodes.size() && n.right < b.nodes.size(), "%s: child index", label);
        }
        CHECK(covered == sc.prims.size(), "%s: leaves cover %zu of %zu objects", label, covered, sc.prims.size());
        printf("%s: bvh mode %d: %zu nodes, depth %u, %zu leaves\n", label, m, b.nodes.size(), b.depth, leaves);
    }
    for (double s : {25.0, 5.0, 1.0}) {
        if (s == 1.0 && !fine_grid) continue;
        Grid g;
        CHECK(build_grid(sc, s, g, err), "%s: build_grid %g: %s", label, s, err.c_str());
        CHECK(g.cell_offset.size() == g.desc.n_cells + 1 && g.cell_aabb.size() == 6 * g.desc.n_cells && g.cell_offset.back() == g.prim_list.size(), "%s: grid tables", label);
        for (uint32_t p : g.prim_list) if (p >= sc.prims.size()) { CHECK(false, "%s: grid list entry out of range", label); break; }
        printf("%s: grid %g: %llu cells, %zu list entries\n", label, s, (unsigned long long)g.desc.n_cells, g.prim_list.size());
    }
}

int main(int argc, char** argv) {
    const std::string root = argc > 1 ? argv[1] : ".";
    std::string err;
    const char* jsons[] = {"simple_fuzzy_metal", "simple_shiny_metal", "simple_hollow_glass", "triangle_test", "triangle_materials",
                           "weekend-1-50-25", "weekend-2-90-5", "weekend-5-80-15", "weekend-final"};
    for (const char* j : jsons) {
        Scene sc;
        const std::string p = root + "/scenes/" + j + ".json";
        CHECK(load_json(p, sc, err), "%s: %s", p.c_str(), err.c_str());
        if (sc.prims.empty()) continue;
        build_all(sc, j, strcmp(j, "weekend-final") == 0);
        CameraIn ci; rt_camera rc;
        CHECK(camera_preset(p, -1, ci, err), "%s: camera preset: %s", j, err.c_str());
        camera_initialise(ci, rc);
    }
    rt_material m; memset(&m, 0, sizeof m); m.type = RT_LAMBERTIAN; m.albedo[0] = 0.8; m.albedo[1] = 0.2; m.albedo[2] = 0.1;
    {
        Scene sc;
        CHECK(load_ply(root + "/scenes/dragon_recon/dragon_vrip_res4.ply", m, sc, err), "res4: %s", err.c_str());
        build_all(sc, "dragon_res4", true);
        subdivide_triangles(sc, 1);
        build_all(sc, "dragon_res4 x4", false);
    }
    {   // big enough for the threaded PLY parse (>= 65 536 rows) and the helper threads of the BVH build (>= 32 768 objects)
        Scene sc;
        CHECK(load_ply(root + "/scenes/dragon_recon/dragon_vrip_res3.ply", m, sc, err), "res3: %s", err.c_str());
        subdivide_triangles(sc, 1);
        Bvh b;
        CHECK(build_bvh(sc, BVH_SAH_POSITION, b, err), "res3 x4: %s", err.c_str());
        printf("dragon_res3 x4: %zu objects, %zu nodes, depth %u\n", sc.prims.size(), b.nodes.size(), b.depth);
        for (int c = 0; c < 4; ++c) { CameraIn ci; CHECK(camera_preset(sc.filename.empty() ? "scenes/dragon_recon/dragon_vrip_res3.ply" : sc.filename, c, ci, err), "camera %d", c); }
    }
    {   // error paths: missing file, truncated JSON, bad PLY header
        Scene sc;
        CHECK(!load_json(root + "/scenes/does-not-exist.json", sc, err), "missing file accepted");
        const std::string tmp = "/tmp/host_sanitize_bad";
        FILE* f = fopen((tmp + ".json").c_str(), "w"); fputs("{\"objects\": [ {\"type\": \"Sphere\", \"center\": [0, 0", f); fclose(f);
        CHECK(!load_json(tmp + ".json", sc, err), "truncated JSON accepted");
        f = fopen((tmp + ".ply").c_str(), "w"); fputs("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nend_header\n0 0\n", f); fclose(f);
        Scene s2; CHECK(!load_ply(tmp + ".ply", m, s2, err), "bad PLY accepted");
    }
    {   // resolve + writers
        std::vector<double> sum(3 * 64 * 36); for (size_t i = 0; i < sum.size(); ++i) sum[i] = (double)(i % 97) / 24.0;
        std::vector<uint8_t> rgb(sum.size());
        resolve_pixels(sum.data(), 64 * 36, 4, rgb.data());
        CHECK(write_ppm_p3("/tmp/host_sanitize.ppm", 64, 36, rgb.data(), err), "p3: %s", err.c_str());
        CHECK(write_ppm_p6("/tmp/host_sanitize6.ppm", 64, 36, rgb.data(), err), "p6: %s", err.c_str());
    }
    printf("%s (%d failures)\n", fails ? "FAILED" : "ok", fails);
    return fails ? 1 : 0;
}
