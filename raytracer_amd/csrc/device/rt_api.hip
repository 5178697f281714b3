// rt_api.hip — extern "C" entry points of librt_hip (include/rt_hip.h): scene upload, BVH / grid
// flattening, the wavefront bounce loop, and the parity entries.  Single translation unit with the kernels.
#include "rt_kernels.hip"
#include "../../../include/rt_hip.h"

#include <dlfcn.h>
#include <rccl/rccl.h>       // types and enums only: the functions are looked up with dlsym (rt_comm_*)

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace rtb {   // rt_build.hip
int build(const rt_prim* prims, const double* centroids, uint32_t n, int mode, rt_bvh_node* nodes_out, uint32_t* n_nodes_out,
          uint32_t* order_out, uint32_t* depth_out, hipStream_t st, std::string& err);
struct GridResult { rt_grid_desc desc; double* cell_aabb; unsigned long long* cell_offset; uint32_t* prim_list; unsigned long long list_len; };
int build_grid(const rt_prim* prims, uint32_t n, double s, GridResult& out, hipStream_t st, std::string& err);
}

// Workspace of rt_render_device_noalloc (include/rt_hip.h): ray queues, hit queue, per-path radiance, queue lengths and
// counters for up to RT_NOALLOC_MAX_PATHS paths in flight, in a static device array instead of hipMalloc memory.  Each GPU the
// library runs on holds its own copy (50 MB) until the process ends.  Host code never uses the name `scratch` as an address:
// rt_init looks the device address up with hipGetSymbolAddress, once per device, and the kernels receive pointers derived
// from that, exactly as they receive the hipMalloc'd workspace in rt_render_device.
#define RT_NOALLOC_MAX_PATHS (1u << 18)
#define RT_NOALLOC_MAX_DEPTH 1022u
#define RT_NOALLOC_BYTES ((size_t)RT_NOALLOC_MAX_PATHS * 196u + (size_t)(RT_NOALLOC_MAX_DEPTH + 2u) * 8u + 4096u)
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_BYTES];

namespace {

thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(RT_E_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

bool g_inited = false;
int g_device = -1;
bool g_profiling = false;
hipStream_t g_stream = nullptr;       // the current device's stream (g_dev[g_device].stream)
// what rt_init keeps per device: the library's stream, two events for the pipelined queue-length read-back, and the device
// address of the static workspace of rt_render_device_noalloc (every GPU holds its own copy of the array at its own address)
#define RT_MAX_DEVICES 64
struct DevState { hipStream_t stream = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; unsigned char* scratch = nullptr; bool ready = false; };
DevState g_dev[RT_MAX_DEVICES];
uint32_t g_extend_blocks = 2048;      // fixed grids: 2048 x 128 threads = 16 waves on each of 256 CUs
uint32_t g_stream_blocks = 4096;
// k_accumulate: each lane streams one pixel's run of samples, so a wave keeps 64 cache lines live; four waves per CU keep
// those inside the 32 KB L1 (4096 x 512 threads: 7.0 ms for the 4K frame's 128 samples, 1024 x 64: 5.7 ms)
uint32_t g_acc_blocks = 1024, g_acc_threads = 64;           // RT_ACC_BLOCKS, RT_ACC_THREADS
uint64_t g_batch_paths = 0;           // paths in flight per batch; 0 = size from free HBM (RT_BATCH_PATHS overrides)
uint32_t g_tune = (8u << 16) | (1u << 8) | 1u;   // refill threshold | cost(internal) | cost(leaf) of the phase vote
unsigned long long* g_dbg = nullptr;  // diagnostic builds (-DRT_STAMPS) only
int g_extend_variant = 2;             // 1 = one ray per thread per pass, 2 = persistent waves with dynamic refill
int g_lg_samples = 7;                 // log2 of the samples of one pixel per generate wave; 7 = choose per batch (RT_LG_SAMPLES overrides)
bool g_tri_pre = true;                // RT_TRI_PRE=0: naive / grid walks recompute the triangle normal per test (A/B)
bool g_bounce_log = false;            // RT_BOUNCE_LOG=1: per-bounce kernel times on stderr while profiling is on
bool g_cell_tri = true;               // RT_CELL_TRI=0: the grid walk reads list entry -> plane record instead of the per-position records (A/B)
bool g_compact = true;                // use the f32-lattice node/triangle forms when a scene allows it (RT_COMPACT=0 disables)
uint32_t g_grid_coop_share_num = 1, g_grid_coop_share_den = 2;   // ... and only if such cells hold at least this share of all list entries (RT_GRID_COOP_SHARE, percent)
uint32_t g_grid_coop_min = 128;       // RT_GRID_COOP_MIN: cells whose list has at least this many triangles are tested by the whole wave (k_extend_grid_coop); 0 = never
bool g_cell_tri_always = false;       // RT_CELL_TRI=2: per-position plane records also for sparse grids
bool g_grid_codes = true;             // RT_GRID_CODES=0: the grid walk always reads a cell's box and list offsets from the tables (A/B)
int g_shade_mode = 3;                 // RT_SHADE_MODE bit 0: prefetch the next tile's queue entry; bit 1: 48-byte spheres in shade (k_shade<MODE>).  Both on:
                                      // weekend-final shade 58.3 -> 53.4 ms (profiles/round3_shade_variants.md); RT_SHADE_MODE=0 is the round-2 kernel
int g_lds_tree = 1;                   // RT_LDS_TREE=0: never stage a small scene's pair / primitive tables in LDS (k_extend_bvh2<..., LDS>), 2: also for short queues, 3: fail if the scene does not qualify (tests)
uint32_t g_lds_min_rays = 1u << 16;   // RT_LDS_MIN: queues shorter than this keep the 128-thread kernel (a short queue wants its rays spread over the CUs)
uint32_t g_lds_bytes = 0, g_cus = 256;   // per-block LDS limit and CU count of the current device
bool g_pair_lines = true;             // RT_PAIR_LINES=0: the lattice pair table keeps the reference's numbering (parent pair p -> left child's pair p + 1) instead of
                                      // the line-paired order of pair_line_order() (A/B: extend 569 -> 550 ms on the headline workload)
uint32_t g_pair_lines_min = 32768;    // RT_PAIR_LINES_MIN: ... for trees of at least this many pairs (2 MB of records)
bool g_shade_tri32 = true;            // RT_SHADE_TRI32=0
bool g_pairn = false;                 // RT_PAIRN=1: lattice scenes use the 32-byte new-planes record (DPairN) instead of the 64-byte DPair32.  Opt-in:
                                      // measured 8.5 % slower on the headline workload (profiles/round3_extend_variants.md), kept for the A/B

struct EventPool {
    std::vector<hipEvent_t> ev; size_t used = 0;
    hipEvent_t get() {
        if (used == ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; ev.push_back(e); }
        return ev[used++];
    }
    void reset() { used = 0; }
};
EventPool g_events;
struct TimedSpan { hipEvent_t a, b; int kind; };
enum { K_GENERATE = 0, K_EXTEND, K_SHADE, K_RESOLVE };

struct Workspace {
    uint64_t cap = 0;            // path capacity
    uint32_t depth_cap = 0;      // entries in `counts`
    double* qbuf[2] = {nullptr, nullptr};   // 9 doubles per path each
    uint32_t* qid[2] = {nullptr, nullptr};
    int32_t* hit_prim = nullptr; double* hit_tp = nullptr; double* hit_tn = nullptr;
    double* rad = nullptr;
    uint32_t* counts = nullptr;      // queue lengths per bounce, then (at +depth_cap) the extend kernels' fetch counters
    unsigned long long* stats = nullptr;
    uint32_t* h_count = nullptr; // pinned, 16 words: ring of the last queue lengths read back
    unsigned long long* h_stats = nullptr;
    bool is_static = false;      // a view of the static device array (rt_render_device_noalloc): nothing to free, no pinned host words
    void release() {
        if (is_static) return;
        for (int i = 0; i < 2; ++i) { if (qbuf[i]) (void)hipFree(qbuf[i]); if (qid[i]) (void)hipFree(qid[i]); qbuf[i] = nullptr; qid[i] = nullptr; }
        if (hit_prim) (void)hipFree(hit_prim); if (hit_tp) (void)hipFree(hit_tp); if (hit_tn) (void)hipFree(hit_tn); if (rad) (void)hipFree(rad);
        if (counts) (void)hipFree(counts); if (stats) (void)hipFree(stats);
        if (h_count) (void)hipHostFree(h_count); if (h_stats) (void)hipHostFree(h_stats);
        hit_prim = nullptr; hit_tp = nullptr; hit_tn = nullptr; rad = nullptr; counts = nullptr; stats = nullptr; h_count = nullptr; h_stats = nullptr;
        cap = 0; depth_cap = 0;
    }
    RayQ q(int i) const {
        RayQ r; double* b = qbuf[i];
        r.ox = b; r.oy = b + cap; r.oz = b + 2 * cap; r.dx = b + 3 * cap; r.dy = b + 4 * cap; r.dz = b + 5 * cap;
        r.tr = b + 6 * cap; r.tg = b + 7 * cap; r.tb = b + 8 * cap; r.id = qid[i];
        return r;
    }
};

V3 tov(const double* p) { V3 r; r.x = p[0]; r.y = p[1]; r.z = p[2]; return r; }

} // namespace

struct rt_scene {
    uint32_t n_prims = 0, n_mats = 0;
    bool mixed = false;                // spheres and triangles together: shade needs the accept-time t as well
    bool draw_boxes = false;           // Options.draw_boxes (rtweekend.rs:92)
    std::vector<rt_prim> h_prims;
    DPrim* d_prims = nullptr;          // file order (naive, grid)
    DTriPre* d_tri_pre = nullptr;      // all-triangle scenes: normal and plane offset per triangle for the naive / grid walks
    DMat* d_mats = nullptr;
    // BVH
    bool have_bvh = false; uint32_t n_nodes = 0, bvh_depth = 0;
    bool bvh_empty_leaf = false;       // a leaf without primitives (only the root of an empty scene): plain kernel only
    DNode* d_nodes = nullptr; DPrim* d_prims_bvh = nullptr; StackEntry* d_stack = nullptr;
    uint32_t stack_blocks = 0;         // extend blocks d_stack was sized for (RT_EXTEND_BLOCKS may change between rt_init calls)
    DPair32* d_pairs32 = nullptr; DPrim32* d_prims32 = nullptr; double lattice_scale = 0.0;   // compact forms; scale 0 = not available
    DPair64* d_pairs64 = nullptr; DSphere* d_spheres = nullptr;                               // generic pairs; spheres only for all-sphere scenes
    uint32_t pair64_stride = (uint32_t)sizeof(DPair64);                                       // 112, or 128 for big trees (one record per L1 line)
    uint32_t n_pairs64 = 0, n_bvh_prims = 0;                                                  // entries of d_pairs64 / d_prims_bvh (the LDS variant stages both)
    DPairN* d_pairsn = nullptr; uint32_t root_own[6] = {0, 0, 0, 0, 0, 0};                    // new-planes-only pairs (two accesses per visit), when the tree allows
    double root_mn[3] = {0, 0, 0}, root_mx[3] = {0, 0, 0};
    // grid
    bool have_grid = false; DGrid grid{}; rt_grid_desc grid_desc{}; uint64_t grid_list_len = 0;
    double* d_cell_aabb = nullptr; uint64_t* d_cell_offset = nullptr; uint32_t* d_prim_list = nullptr;
    DCellTri* d_cell_tri = nullptr;    // all-triangle scenes: plane record + index per list position (k_cell_tri)
    uint32_t* d_cell_code = nullptr;   // two bits per cell: list not empty / box not the implicit one (DGrid::cell_code)
    uint32_t* d_cell_chunk = nullptr; DCellChunk* d_chunks = nullptr; uint32_t n_chunks = 0;   // cooperative cell test: chunked lists of the well-filled cells
    Workspace ws;
};

namespace {

// Order of the 64-byte sibling-pair records in the lattice table (DPair32 / DPairN).  The traversal reads a pair right after its
// parent's pair, and the vector L1's line is 128 bytes = two records: a pair that starts a line (even slot) is followed in the SAME line
// by the pair of its bigger child (by box area: the child more rays enter, and enter first), so that descent finds its record in
// the line the previous visit fetched; a pair in the second half of a line sends both children to lines of their own.  Subtrees
// stay contiguous (depth-first, big child first).  Slots left empty by the alignment are filled with pairs that have no internal
// child (they need no line-mate; measured: no effect on speed, the table is 22 % smaller), else stay zero.  Device-side numbering only: the tree, the visit order and
// every counter are what they were.  perm[old pair] = slot; returns the number of slots (lines == false: the identity).  Pair 0 (dummy + root) and pair 1 (the
// root's children, where every ray starts: k_extend_bvh2 derives it from the root node) keep their places.
uint32_t pair_line_order(const std::vector<DNode>& dn, uint32_t n_nodes, size_t n_pairs, bool lines, std::vector<uint32_t>& perm) {
    perm.assign(n_pairs, 0xffffffffu);
    perm[0] = 0;
    if (!lines || n_pairs < 2 || n_nodes < 3 || (dn[1].b & RT_LEAF)) { for (size_t p = 1; p < n_pairs; ++p) perm[p] = (uint32_t)p; return (uint32_t)n_pairs; }
    auto area = [&](const DNode& X) { const double ex = X.mx[0] - X.mn[0], ey = X.mx[1] - X.mn[1], ez = X.mx[2] - X.mn[2]; return ex * ey + ey * ez + ez * ex; };
    auto internal = [&](uint32_t di) { return !(dn[di].b & RT_LEAF); };
    struct It { uint32_t p; bool fresh; };                 // old pair index (children of device node 2p, 2p+1 are dn[2p].a ...); must it start a line?
    std::vector<It> todo; todo.push_back({1u, false});
    uint32_t next = 1;
    std::vector<uint32_t> holes;
    while (!todo.empty()) {
        const It it = todo.back(); todo.pop_back();
        const uint32_t l = 2 * it.p, r = l + 1;            // the two device nodes this pair describes
        const bool li = internal(l), ri = r <= n_nodes && internal(r);
        uint32_t slot;
        if (it.fresh && (next & 1u) && (li || ri)) holes.push_back(next++);                 // start a line; the skipped slot is filled later
        if (it.fresh && !(li || ri) && !holes.empty()) { slot = holes.back(); holes.pop_back(); }   // no internal child: needs no line-mate, any free half line does
        else slot = next++;
        perm[it.p] = slot;
        const bool mate = (slot & 1u) == 0;
        if (li && ri) {
            const bool left_big = area(dn[l]) >= area(dn[r]);        // measured: bigger 540.4 ms, left 545.0, right 543.9, smaller 547.7
            const uint32_t big = (left_big ? dn[l].a : dn[r].a) >> 1, small = (left_big ? dn[r].a : dn[l].a) >> 1;
            todo.push_back({small, true}); todo.push_back({big, !mate});
        } else if (li) todo.push_back({dn[l].a >> 1, !mate});
        else if (ri) todo.push_back({dn[r].a >> 1, !mate});
    }
    return next;
}

// c == (f64)f32 * S bit for bit?  (then the device can rebuild c from the f32 with the same two operations)
bool on_lattice(double c, double S, float& f) {
    if (!(c - c == 0.0)) return false;                 // inf / NaN
    f = (float)(c / S);
    const double back = (double)f * S;
    return memcmp(&back, &c, sizeof c) == 0;
}

DPrim to_dprim(const rt_prim& p, uint32_t orig) {
    DPrim d; for (int i = 0; i < 9; ++i) d.v[i] = p.v[i];
    d.tm = (p.type == RT_TRIANGLE ? 0x80000000u : 0u) | (p.material & 0x7fffffffu);
    d.orig = orig;
    return d;
}

int ensure_workspace(rt_scene* sc, uint64_t cap, uint32_t depth_cap) {
    Workspace& w = sc->ws;
    if (w.cap >= cap && w.depth_cap >= depth_cap) return RT_OK;
    if (cap < w.cap) cap = w.cap;
    if (depth_cap < w.depth_cap) depth_cap = w.depth_cap;
    w.release();
    for (int i = 0; i < 2; ++i) {
        HIPCHECK(hipMalloc(&w.qbuf[i], cap * 9 * sizeof(double)));
        HIPCHECK(hipMalloc(&w.qid[i], cap * sizeof(uint32_t)));
    }
    HIPCHECK(hipMalloc(&w.hit_prim, cap * sizeof(int32_t)));
    HIPCHECK(hipMalloc(&w.hit_tp, cap * sizeof(double)));
    HIPCHECK(hipMalloc(&w.hit_tn, cap * sizeof(double)));
    HIPCHECK(hipMalloc(&w.rad, cap * 3 * sizeof(double)));
    HIPCHECK(hipMalloc(&w.counts, (size_t)depth_cap * 2 * sizeof(uint32_t)));
    HIPCHECK(hipMalloc(&w.stats, ST_N * sizeof(unsigned long long)));
    HIPCHECK(hipHostMalloc(&w.h_count, 64));
    HIPCHECK(hipHostMalloc(&w.h_stats, ST_N * sizeof(unsigned long long)));
    w.cap = cap; w.depth_cap = depth_cap;
    return RT_OK;
}

int launch_extend(rt_scene* sc, int algorithm, const RayQ& in, const uint32_t* count, uint64_t n_fixed, double tmin, double tmax,
                  const ExtendOut& out, uint32_t* fetch_ctr, unsigned long long* d_stats, hipStream_t st, uint64_t n_hint = 0) {
    // n_hint = an upper bound of the queue length when the host knows one (0: unknown): a short queue gets a grid of one lane
    // per ray instead of 4096 waves that each pay an atomic on the fetch counter to learn that nothing is left (88 per
    // microsecond on one word).  Without a hint the persistent kernel's surplus blocks leave on their own (they read *count).
    uint32_t nblk = g_extend_blocks;
    // the BVH kernels index the continuation stack by (level, thread): never launch more blocks than it was sized for
    if (algorithm == RT_ALGO_BVH && sc->stack_blocks && nblk > sc->stack_blocks) nblk = sc->stack_blocks;
    if (n_hint) { const uint64_t want = ((n_hint + RT_BLOCK - 1) / RT_BLOCK + 7) & ~7ull; if (want < nblk) nblk = (uint32_t)want; }
    dim3 grid(nblk), block(RT_BLOCK);
    switch (algorithm) {
    case RT_ALGO_BVH:
        if (!sc->have_bvh) return fail(RT_E_STATE, "rt_scene_set_bvh() has not been called (reference: \"Please run HittableList.init() first\")");
        if (sc->draw_boxes)            // debug overlay: the simple one-ray-per-thread kernel carries the edge flag
            hipLaunchKernelGGL(k_extend_bvh<true>, grid, block, 0, st, sc->d_nodes, sc->d_prims_bvh, in, count, n_fixed, tmin, tmax, out, sc->d_stack, d_stats);
        else if (g_extend_variant == 1 || sc->bvh_empty_leaf)
            hipLaunchKernelGGL(k_extend_bvh<false>, grid, block, 0, st, sc->d_nodes, sc->d_prims_bvh, in, count, n_fixed, tmin, tmax, out, sc->d_stack, d_stats);
        else {
            BvhView v; v.nodes = sc->d_nodes; v.prims = sc->d_prims_bvh; v.pairs64 = sc->d_pairs64; v.spheres = sc->d_spheres;
            v.pairs32 = sc->d_pairs32; v.prims32 = sc->d_prims32; v.scale = sc->lattice_scale; v.pair64_stride = sc->pair64_stride;
            v.pairsn = sc->d_pairsn; for (int f = 0; f < 6; ++f) v.root_own[f] = sc->root_own[f];
            const bool c32 = g_compact && sc->lattice_scale > 0.0, sph = g_compact && sc->d_spheres != nullptr;
            const bool cn = c32 && g_pairn && sc->d_pairsn != nullptr;
            // small scenes with generic pairs (the JSON scenes): tree and primitives fit in LDS beside a 4-entry stack window
            v.lds_pairs16 = sc->n_pairs64 * (uint32_t)(sizeof(DPair64) / 16); v.lds_prims16 = sc->n_bvh_prims * (uint32_t)((sph ? sizeof(DSphere) : sizeof(DPrim)) / 16);
            const size_t lds_need = ((size_t)RT_LSTK * RT_LBLOCK + v.lds_pairs16 + v.lds_prims16) * 16;
            const uint32_t lblocks = (uint32_t)std::min<uint64_t>(g_cus, (uint64_t)sc->stack_blocks * RT_BLOCK / RT_LBLOCK);
            const uint64_t n_known = count ? n_hint : n_fixed;        // 0 = unknown (a device-side count without a hint)
            if (!c32 && g_lds_tree && sc->n_pairs64 && lds_need <= g_lds_bytes && lblocks > 0 && (g_lds_tree >= 2 || n_known == 0 || n_known >= g_lds_min_rays)) {
#define RT_LAUNCH_BVH2L(P, T) do { static bool attr_set[RT_MAX_DEVICES]; \
                if (!attr_set[g_device]) { HIPCHECK(hipFuncSetAttribute((const void*)k_extend_bvh2<0, P, T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_lds_bytes)); attr_set[g_device] = true; } \
                hipLaunchKernelGGL((k_extend_bvh2<0, P, T, true>), dim3(lblocks), dim3(RT_LBLOCK), lds_need, st, v, in, count, n_fixed, tmin, tmax, out, sc->d_stack, fetch_ctr, g_tune, d_stats, g_dbg); } while (0)
                if (sph) { if (out.trace) RT_LAUNCH_BVH2L(2, true); else RT_LAUNCH_BVH2L(2, false); }
                else     { if (out.trace) RT_LAUNCH_BVH2L(0, true); else RT_LAUNCH_BVH2L(0, false); }
#undef RT_LAUNCH_BVH2L
                break;
            }
            if (g_lds_tree == 3) return fail(RT_E_STATE, "RT_LDS_TREE=3 (tests): this scene cannot take the LDS-resident kernel (lattice scene, or %zu bytes of LDS needed of %u)", lds_need, g_lds_bytes);
#define RT_LAUNCH_BVH2(N, P, T) hipLaunchKernelGGL((k_extend_bvh2<N, P, T>), grid, block, 0, st, v, in, count, n_fixed, tmin, tmax, out, sc->d_stack, fetch_ctr, g_tune, d_stats, g_dbg)
            if (cn)       { if (out.trace) RT_LAUNCH_BVH2(2, 1, true); else RT_LAUNCH_BVH2(2, 1, false); }
            else if (c32) { if (out.trace) RT_LAUNCH_BVH2(1, 1, true); else RT_LAUNCH_BVH2(1, 1, false); }
            else if (sph) { if (out.trace) RT_LAUNCH_BVH2(0, 2, true); else RT_LAUNCH_BVH2(0, 2, false); }
            else          { if (out.trace) RT_LAUNCH_BVH2(0, 0, true); else RT_LAUNCH_BVH2(0, 0, false); }
#undef RT_LAUNCH_BVH2
        }
        break;
    case RT_ALGO_NAIVE:
        hipLaunchKernelGGL(k_extend_naive, grid, block, 0, st, sc->d_prims, sc->d_tri_pre, sc->n_prims, in, count, n_fixed, tmin, tmax, out, d_stats);
        break;
    case RT_ALGO_GRID: {
        if (!sc->have_grid) return fail(RT_E_STATE, "rt_scene_set_grid() has not been called (reference: \"Please run Grid::new() first\")");
        // the hoisted triangle constants pay where cells hold many triangles (size 25: -16 %, size 5: -6 % on the res4 dragon);
        // with nearly empty cells (size 1) the second array only costs cache (+5 %)
        const DTriPre* gpre = (sc->grid_list_len * 10ull >= sc->grid.n_cells) ? sc->d_tri_pre : nullptr;
        if (sc->draw_boxes)
            hipLaunchKernelGGL(k_extend_grid<true>, grid, block, 0, st, sc->grid, sc->d_prims, gpre, in, count, n_fixed, tmin, tmax, out, fetch_ctr, d_stats);
        else if (sc->n_chunks && sc->d_tri_pre)          // some cells are well filled: the whole wave tests their lists (all-triangle scenes)
            // always the full grid: a short queue is spread thinly over all waves (see the kernel), not packed into a few
            hipLaunchKernelGGL(k_extend_grid_coop, dim3(g_extend_blocks), block, 0, st, sc->grid, sc->d_prims, sc->d_tri_pre, in, count, n_fixed, tmin, tmax, out, fetch_ctr, d_stats);
        else
            hipLaunchKernelGGL(k_extend_grid<false>, grid, block, 0, st, sc->grid, sc->d_prims, gpre, in, count, n_fixed, tmin, tmax, out, fetch_ctr, d_stats);
        break;
    }
    default:
        return fail(RT_E_ARG, "unknown algorithm %d", algorithm);
    }
    HIPCHECK(hipGetLastError());
    return RT_OK;
}

const DPrim* shade_prims(const rt_scene* sc, int algorithm) { return algorithm == RT_ALGO_BVH ? sc->d_prims_bvh : sc->d_prims; }

void fill_stats(rt_stats* s, const unsigned long long* h) {
    s->scatter_rays = h[ST_SCATTER];
    s->box_tests = h[ST_BOX]; s->prim_tests = h[ST_PRIM];
    s->intersection_checks = h[ST_BOX] + h[ST_PRIM];
    s->traversal_steps = h[ST_STEPS]; s->overlapping_aabb = h[ST_OVERLAP];
    s->gridbox_intersection_checks = h[ST_GRIDBOX];
    s->ray_segments = h[ST_SEGMENTS];
}

// Batch = as many samples per pixel in flight as HBM comfortably holds (196 B of queue state per path): long paths
// leave a tail of small, latency-bound launches per batch, so fewer, larger batches are faster (288 GB per GPU).
// Sizes (and, when needed, allocates) the scene's workspace for renders of s_count samples of a WH-pixel image; *k_out = samples per batch.
int size_workspace(rt_scene* sc, uint64_t WH, uint32_t s_count, uint32_t max_depth, uint32_t* k_out) {
    uint64_t batch_paths = g_batch_paths;
    if (batch_paths == 0) {
        size_t free_b = 0, total_b = 0;
        HIPCHECK(hipMemGetInfo(&free_b, &total_b));
        uint64_t have = free_b + sc->ws.cap * 196ull;                     // what this scene's workspace already holds counts as free
        batch_paths = (uint64_t)((double)have * 0.75 / 196.0);            // every batch pays a fixed tail of small launches: as few as fit
        if (batch_paths < (1ull << 22)) batch_paths = 1ull << 22;
    }
    uint64_t k64 = batch_paths / WH; if (k64 < 1) k64 = 1; if (s_count && k64 > s_count) k64 = s_count;
    while (k64 > 1 && WH * k64 > 0xf0000000ull) --k64;                      // path ids and the extend fetch counter are 32-bit (headroom for one chunk per wave)
    if (s_count) { const uint64_t nb = (s_count + k64 - 1) / k64; k64 = (s_count + nb - 1) / nb; }   // equal batches
    uint32_t k = (uint32_t)k64;
    if (WH * k > 0xffffffffull) return fail(RT_E_LIMIT, "batch too large");
    int rc = ensure_workspace(sc, WH * k, max_depth + 2);
    while (rc == RT_E_HIP && k > 1 && g_batch_paths == 0) {               // the allocation did not fit after all: smaller batches
        (void)hipGetLastError();
        sc->ws.release();
        k = (k + 1) / 2;
        rc = ensure_workspace(sc, WH * k, max_depth + 2);
    }
    if (rc) return rc;
    *k_out = k;
    return RT_OK;
}

// The bounce loop.  `w` is the workspace the call runs in: the scene's own (sized from free HBM by size_workspace) or a view of
// the static device array (rt_render_device_noalloc); k = samples per pixel per batch, already fitted to w.cap.
// Queue lengths: every kernel strides over *count in device memory, so the host needs them only to stop early and to give
// short queues a small grid.  The read-back is pipelined: bounce b+1 is launched before the host looks at the length bounce b
// produced (an empty queue costs two launches that find nothing to do), so the GPU never waits for the host between bounces;
// a queue never grows (a path scatters into at most one ray), so the last length read is an upper bound for the launch shape.
// Without pinned host words (static workspace) nothing is read back and all max_depth bounces are launched.
int render_impl(rt_scene* sc, Workspace& w, uint32_t k, const rt_camera* cam, int algorithm, uint32_t s_begin, uint32_t s_count, uint32_t max_depth,
                uint64_t seed, double* d_fb, hipStream_t st, rt_stats* stats) {
    const uint64_t WH = (uint64_t)cam->image_width * cam->image_height;
    int rc = RT_OK;
    const bool profiling = g_profiling && !w.is_static;                  // event pools grow on demand: not in the no-allocation call
    const bool readback = w.h_count != nullptr;
    DevState& dev = g_dev[g_device];

    DCamera dc; dc.W = cam->image_width; dc.H = cam->image_height;
    dc.center = tov(cam->center); dc.pixel00 = tov(cam->pixel00); dc.du = tov(cam->du); dc.dv = tov(cam->dv);
    dc.ddu = tov(cam->ddu); dc.ddv = tov(cam->ddv); dc.defocus_angle = cam->defocus_angle;

    HIPCHECK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    HIPCHECK(hipMemsetAsync(w.stats, 0, ST_N * sizeof(unsigned long long), st));
    std::vector<TimedSpan> spans;
    g_events.reset();
    double ms[4] = {0, 0, 0, 0}; uint64_t extend_launches = 0;
    auto begin_span = [&](int kind) { if (profiling) { TimedSpan s; s.a = g_events.get(); s.b = g_events.get(); s.kind = kind; (void)hipEventRecord(s.a, st); spans.push_back(s); } };
    auto end_span = [&]() { if (profiling) (void)hipEventRecord(spans.back().b, st); };
    auto drain_spans = [&]() {
        for (const TimedSpan& s : spans) { float f = 0; if (hipEventElapsedTime(&f, s.a, s.b) == hipSuccess) ms[s.kind] += f; }
        spans.clear(); g_events.reset();
    };

    const dim3 sgrid(g_stream_blocks), block(RT_SBLOCK);
    DRootBox rb; memset(&rb, 0, sizeof rb);
    // with draw-boxes the root test must also report its edge flag, so extend does it and generate culls nothing
    if (algorithm == RT_ALGO_BVH && !sc->draw_boxes) { rb.enabled = 1; for (int c = 0; c < 3; ++c) { rb.mn[c] = sc->root_mn[c]; rb.mx[c] = sc->root_mx[c]; } }
    const DPrim* sprims = shade_prims(sc, algorithm);
    bool first = true;
    // camera.rs:155-157: with max_depth == 0 ray_color returns black before world.hit is called: no queue, no counters
    if (s_count == 0 || max_depth == 0) HIPCHECK(hipMemsetAsync(d_fb, 0, WH * 3 * sizeof(double), st));
    for (uint32_t s0 = 0; max_depth != 0 && s0 < s_count; s0 += k) {
        const uint32_t kb = (s_count - s0 < k) ? (s_count - s0) : k;
        HIPCHECK(hipMemsetAsync(w.counts, 0, (size_t)w.depth_cap * 2 * sizeof(uint32_t), st));
        begin_span(K_GENERATE);
        // samples of a pixel per wave: the largest power of two (<= 64) that leaves at most 1/8 of the last group's lanes empty
        uint32_t lg_s = g_lg_samples <= 6 ? (uint32_t)g_lg_samples : 6u;
        if (g_lg_samples > 6) while (lg_s > 0 && (((kb + (1u << lg_s) - 1u) >> lg_s) << lg_s) - kb > kb / 8u) --lg_s;
        hipLaunchKernelGGL(k_generate, sgrid, block, 0, st, dc, rb, s_begin + s0, kb, lg_s, seed, w.q(0), w.counts, w.rad, w.stats);
        end_span();
        uint64_t n_known = readback ? WH * kb : 0;                       // upper bound of the current queue's length (0: none)
        const size_t span0 = spans.size();
        uint32_t launched = 0;
        for (uint32_t b = 0; b < max_depth; ++b) {
            ExtendOut eo; eo.prim = w.hit_prim; eo.t = nullptr; eo.tp = w.hit_tp; eo.tn = sc->mixed ? w.hit_tn : nullptr; eo.hit = nullptr; eo.trace = 0;
            begin_span(K_EXTEND);
            rc = launch_extend(sc, algorithm, w.q(b & 1), w.counts + b, 0, 0.001, __builtin_huge_val(), eo, w.counts + w.depth_cap + b, w.stats, st, n_known);
            if (rc) return rc;
            end_span();
            begin_span(K_SHADE);
            {
                const DSphere* sph = (algorithm == RT_ALGO_BVH && (g_shade_mode & 2)) ? sc->d_spheres : nullptr;     // BVH order, like sprims
                // all-triangle lattice scenes: the BVH-ordered 64-byte-stride lattice triangles (RT_SHADE_TRI32=0: the 80-byte DPrim, A/B)
                const DPrim32* t32 = (algorithm == RT_ALGO_BVH && (g_shade_mode & 2) && g_compact && sc->lattice_scale > 0.0 && g_shade_tri32) ? sc->d_prims32 : nullptr;
#define RT_LAUNCH_SHADE(M) hipLaunchKernelGGL(k_shade<M>, sgrid, block, 0, st, sprims, sph, t32, sc->lattice_scale, sc->d_mats, sc->n_mats, w.q(b & 1), w.counts + b, w.hit_prim, w.hit_tp, sc->mixed ? w.hit_tn : nullptr, \
                               w.q((b + 1) & 1), w.counts + b + 1, w.rad, (uint32_t)WH, s_begin + s0, kb, b, max_depth, seed, w.stats)
                switch (g_shade_mode & 3) { case 1: RT_LAUNCH_SHADE(1); break; case 2: RT_LAUNCH_SHADE(2); break; case 3: RT_LAUNCH_SHADE(3); break; default: RT_LAUNCH_SHADE(0); }
#undef RT_LAUNCH_SHADE
            }
            end_span();
            ++launched;
            if (!readback) continue;
            HIPCHECK(hipMemcpyAsync(w.h_count + (b & 1), w.counts + b + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIPCHECK(hipEventRecord(dev.ev[b & 1], st));
            if (b == 0) continue;
            HIPCHECK(hipEventSynchronize(dev.ev[(b - 1) & 1]));         // bounce b is in flight while the host looks at what bounce b-1 left
            const uint32_t left = w.h_count[(b - 1) & 1];                // = length of the queue bounce b was launched on
            if (left == 0) { --launched; break; }
            n_known = left;
        }
        extend_launches += launched;
        begin_span(K_RESOLVE);
        hipLaunchKernelGGL(k_accumulate, dim3(g_acc_blocks), dim3(g_acc_threads), 0, st, w.rad, (uint32_t)WH, kb, d_fb, first ? 1 : 0);
        end_span();
        first = false;
        if (profiling) {
            HIPCHECK(hipStreamSynchronize(st));
            if (g_bounce_log)                                            // diagnostic: RT_BOUNCE_LOG=1 with profiling on
                for (size_t i = span0, b = 0; i + 1 < spans.size() && spans[i].kind == K_EXTEND; i += 2, ++b) {
                    float fe = 0, fs = 0;
                    (void)hipEventElapsedTime(&fe, spans[i].a, spans[i].b); (void)hipEventElapsedTime(&fs, spans[i + 1].a, spans[i + 1].b);
                    fprintf(stderr, "bounce %2zu: extend %9.3f ms, shade %8.3f ms\n", b, fe, fs);
                }
            drain_spans();
        }
    }
    unsigned long long h_stats_local[ST_N];                              // static workspace: no pinned words, plain blocking copy
    unsigned long long* hs = w.h_stats ? w.h_stats : h_stats_local;
    if (w.h_stats) HIPCHECK(hipMemcpyAsync(hs, w.stats, ST_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    if (!w.h_stats) HIPCHECK(hipMemcpy(hs, w.stats, ST_N * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHECK(hipGetLastError());
    if (profiling) drain_spans();
    auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        memset(stats, 0, sizeof *stats);
        fill_stats(stats, hs);
        stats->primary_rays = WH * s_count;                           /* camera.rs:70, exact */
        stats->extend_launches = extend_launches;
        stats->generate_ms = ms[K_GENERATE]; stats->extend_ms = ms[K_EXTEND]; stats->shade_ms = ms[K_SHADE]; stats->resolve_ms = ms[K_RESOLVE];
        stats->render_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    }
    return RT_OK;
}

// argument checks shared by the render entry points
#define RT_MAX_DEPTH_LIMIT 0x100000u       // counts[] holds two words per bounce; the reference's usual depth is 50 (main.rs:76)
int check_render_args(rt_scene* sc, const rt_camera* cam, int algorithm, uint32_t max_depth, const double* d_fb) {
    if (!g_inited) return fail(RT_E_STATE, "rt_init() has not been called");
    if (!sc || !cam || !d_fb) return fail(RT_E_ARG, "null argument");
    const uint64_t WH = (uint64_t)cam->image_width * cam->image_height;
    if (WH == 0 || WH > 0x7fffffffull) return fail(RT_E_LIMIT, "image of %llu pixels is outside (0, 2^31)", (unsigned long long)WH);
    if (algorithm < 0 || algorithm > 2) return fail(RT_E_ARG, "unknown algorithm %d", algorithm);
    if (algorithm == RT_ALGO_BVH && !sc->have_bvh) return fail(RT_E_STATE, "rt_scene_set_bvh() has not been called");
    if (algorithm == RT_ALGO_GRID && !sc->have_grid) return fail(RT_E_STATE, "rt_scene_set_grid() has not been called");
    if (max_depth > RT_MAX_DEPTH_LIMIT) return fail(RT_E_LIMIT, "max_depth %u is above the limit of %u", max_depth, RT_MAX_DEPTH_LIMIT);
    return RT_OK;
}

int render_dynamic(rt_scene* sc, const rt_camera* cam, int algorithm, uint32_t s_begin, uint32_t s_count, uint32_t max_depth,
                   uint64_t seed, double* d_fb, hipStream_t st, rt_stats* stats) {
    { const int rca = check_render_args(sc, cam, algorithm, max_depth, d_fb); if (rca) return rca; }
    const uint64_t WH = (uint64_t)cam->image_width * cam->image_height;
    uint32_t k = 0;
    // an empty render (no samples, or camera.rs:155-157's depth 0) only clears the framebuffer: one sample's worth of workspace,
    // not "as many samples as HBM holds" (s_count 0 means that to size_workspace)
    { const int rcw = size_workspace(sc, WH, (s_count && max_depth) ? s_count : 1u, max_depth, &k); if (rcw) return rcw; }
    return render_impl(sc, sc->ws, k, cam, algorithm, s_begin, s_count, max_depth, seed, d_fb, st, stats);
}

} // namespace

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }

int rt_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(RT_E_HIP, "no HIP device available (%s); this library has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(RT_E_ARG, "device %d out of range [0,%d)", device, n);
    if (device >= RT_MAX_DEVICES) return fail(RT_E_LIMIT, "device %d: this library keeps state for %d devices", device, RT_MAX_DEVICES);
    HIPCHECK(hipSetDevice(device));
    DevState& dv = g_dev[device];
    if (!dv.ready) {                                  // first rt_init on this device; a later one finds its state again
        // built into a local and committed only when every step succeeded: a failed rt_init leaves the device un-initialised
        // (and what it did create released), so a retry starts over instead of finding half a state
        DevState nv;
        auto undo = [&]() { if (nv.ev[0]) (void)hipEventDestroy(nv.ev[0]); if (nv.ev[1]) (void)hipEventDestroy(nv.ev[1]); if (nv.stream) (void)hipStreamDestroy(nv.stream); };
        hipError_t e2 = hipStreamCreateWithFlags(&nv.stream, hipStreamNonBlocking);
        if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&nv.ev[0], hipEventDisableTiming);
        if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&nv.ev[1], hipEventDisableTiming);
        // rt_render_device_noalloc's workspace: the array's name is a host-side stand-in, its device address is looked up once
        // per device, with that device current
        void* p = nullptr;
        if (e2 == hipSuccess) e2 = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e2 != hipSuccess || !p) { undo(); return fail(RT_E_HIP, "rt_init(%d): device set-up failed: %s", device, hipGetErrorString(e2)); }
        nv.scratch = (unsigned char*)p; nv.ready = true;
        dv = nv;
    }
    g_stream = dv.stream; g_device = device;
    g_batch_paths = 0;                                // every rt_init re-reads the environment: an unset variable means the default again
    if (const char* s = getenv("RT_BATCH_PATHS")) { unsigned long long v = strtoull(s, nullptr, 10); if (v >= 4096) g_batch_paths = v; }
    if (const char* s = getenv("RT_EXTEND_VARIANT")) { int v = atoi(s); if (v == 1 || v == 2) g_extend_variant = v; }
    if (const char* s = getenv("RT_COMPACT")) g_compact = atoi(s) != 0;
    g_pairn = false; if (const char* s = getenv("RT_PAIRN")) g_pairn = atoi(s) != 0;
    g_shade_tri32 = true; if (const char* s = getenv("RT_SHADE_TRI32")) g_shade_tri32 = atoi(s) != 0;
    g_pair_lines = true; if (const char* s = getenv("RT_PAIR_LINES")) g_pair_lines = atoi(s) != 0;
    g_pair_lines_min = 32768; if (const char* s = getenv("RT_PAIR_LINES_MIN")) g_pair_lines_min = (uint32_t)strtoul(s, nullptr, 10);
    g_lds_tree = 1; if (const char* s = getenv("RT_LDS_TREE")) g_lds_tree = atoi(s);
    g_lds_min_rays = 1u << 16; if (const char* s = getenv("RT_LDS_MIN")) g_lds_min_rays = (uint32_t)strtoul(s, nullptr, 10);
    { int v = 0; HIPCHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device)); g_lds_bytes = v > 0 ? (uint32_t)v : 0;
      HIPCHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device)); g_cus = v > 0 ? (uint32_t)v : 256; }
    g_grid_codes = true; if (const char* s = getenv("RT_GRID_CODES")) g_grid_codes = atoi(s) != 0;
    g_shade_mode = 3; if (const char* s = getenv("RT_SHADE_MODE")) g_shade_mode = atoi(s) & 3;
    g_grid_coop_min = 128; if (const char* s = getenv("RT_GRID_COOP_MIN")) g_grid_coop_min = (uint32_t)strtoul(s, nullptr, 10);
    g_grid_coop_share_num = 1; g_grid_coop_share_den = 2;
    if (const char* s = getenv("RT_GRID_COOP_SHARE")) { g_grid_coop_share_num = (uint32_t)strtoul(s, nullptr, 10); g_grid_coop_share_den = 100; }
    if (const char* s = getenv("RT_LG_SAMPLES")) { int v = atoi(s); if (v >= 0 && v <= 7) g_lg_samples = v; }
    if (const char* s = getenv("RT_TRI_PRE")) g_tri_pre = atoi(s) != 0;
    if (const char* s = getenv("RT_BOUNCE_LOG")) g_bounce_log = atoi(s) != 0;
    g_cell_tri_always = false;
    if (const char* s = getenv("RT_CELL_TRI")) { g_cell_tri = atoi(s) != 0; g_cell_tri_always = atoi(s) == 2; }
    if (const char* s = getenv("RT_TUNE")) g_tune = (uint32_t)strtoul(s, nullptr, 0);
    if (const char* s = getenv("RT_STREAM_BLOCKS")) { unsigned long v = strtoul(s, nullptr, 10); if (v >= 8 && v <= 65536) g_stream_blocks = (uint32_t)v & ~7u; }
    if (const char* s = getenv("RT_ACC_BLOCKS")) { unsigned long v = strtoul(s, nullptr, 10); if (v >= 1 && v <= 1048576) g_acc_blocks = (uint32_t)v; }
    if (const char* s = getenv("RT_ACC_THREADS")) { unsigned long v = strtoul(s, nullptr, 10); if (v >= 64 && v <= RT_SBLOCK && v % 64 == 0) g_acc_threads = (uint32_t)v; }
    if (const char* s = getenv("RT_EXTEND_BLOCKS")) { unsigned long v = strtoul(s, nullptr, 10); if (v >= 8 && v <= 65536) g_extend_blocks = (uint32_t)v & ~7u; }
#ifdef RT_STAMPS
    if (!g_dbg) { HIPCHECK(hipMalloc(&g_dbg, 24 * sizeof(unsigned long long))); HIPCHECK(hipMemset(g_dbg, 0, 24 * sizeof(unsigned long long))); }
#endif
    g_inited = true;
    return RT_OK;
}

int rt_set_profiling(int on) { g_profiling = on != 0; return RT_OK; }

rt_scene* rt_scene_create(const rt_prim* prims, uint32_t n_prims, const rt_material* mats, uint32_t n_mats) {
    if (!g_inited) { fail(RT_E_STATE, "rt_init() has not been called"); return nullptr; }
    if ((n_prims && !prims) || (n_mats && !mats)) { fail(RT_E_ARG, "null array"); return nullptr; }
    if (n_prims > 0x1fffffffu) { fail(RT_E_LIMIT, "more than 2^29 primitives"); return nullptr; }
    for (uint32_t i = 0; i < n_prims; ++i) {
        if (prims[i].type > RT_TRIANGLE) { fail(RT_E_ARG, "primitive %u has unknown type %u", i, prims[i].type); return nullptr; }
        if (prims[i].material >= n_mats) { fail(RT_E_ARG, "primitive %u names material %u of %u", i, prims[i].material, n_mats); return nullptr; }
    }
    for (uint32_t i = 0; i < n_mats; ++i)
        if (mats[i].type > RT_DIELECTRIC) { fail(RT_E_ARG, "material %u has unknown type %u", i, mats[i].type); return nullptr; }
    rt_scene* sc = new rt_scene();
    sc->n_prims = n_prims; sc->n_mats = n_mats;
    sc->h_prims.assign(prims, prims + n_prims);
    { bool sph = false, tri = false; for (uint32_t i = 0; i < n_prims; ++i) (prims[i].type == RT_TRIANGLE ? tri : sph) = true; sc->mixed = sph && tri; }
    std::vector<DPrim> dp(n_prims ? n_prims : 1);
    for (uint32_t i = 0; i < n_prims; ++i) dp[i] = to_dprim(prims[i], i);
    std::vector<DMat> dm(n_mats ? n_mats : 1);
    for (uint32_t i = 0; i < n_mats; ++i) {
        dm[i].albedo[0] = mats[i].albedo[0]; dm[i].albedo[1] = mats[i].albedo[1]; dm[i].albedo[2] = mats[i].albedo[2];
        dm[i].param = mats[i].param; dm[i].type = mats[i].type; dm[i].pad = 0;
    }
    bool ok = hipMalloc(&sc->d_prims, dp.size() * sizeof(DPrim)) == hipSuccess
           && hipMalloc(&sc->d_mats, dm.size() * sizeof(DMat)) == hipSuccess
           && hipMemcpy(sc->d_prims, dp.data(), dp.size() * sizeof(DPrim), hipMemcpyHostToDevice) == hipSuccess
           && hipMemcpy(sc->d_mats, dm.data(), dm.size() * sizeof(DMat), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { fail(RT_E_HIP, "scene upload failed: %s", hipGetErrorString(hipGetLastError())); rt_scene_destroy(sc); return nullptr; }
    bool all_tri = n_prims > 0; for (uint32_t i = 0; i < n_prims; ++i) all_tri = all_tri && prims[i].type == RT_TRIANGLE;
    if (all_tri && g_tri_pre) {
        if (hipMalloc(&sc->d_tri_pre, (size_t)n_prims * sizeof(DTriPre)) != hipSuccess) { (void)hipGetLastError(); sc->d_tri_pre = nullptr; }   // optional: the walks fall back to the full test
        else { hipLaunchKernelGGL(k_tri_pre, dim3((n_prims + 255) / 256), dim3(256), 0, g_stream, sc->d_prims, n_prims, sc->d_tri_pre); (void)hipStreamSynchronize(g_stream); }
    }
    return sc;
}

void rt_scene_destroy(rt_scene* sc) {
    if (!sc) return;
    (void)hipDeviceSynchronize();
#ifdef RT_STAMPS
    if (g_dbg) {
        unsigned long long h[24];
        if (hipMemcpy(h, g_dbg, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[12]) {
            fprintf(stderr, "[stamps] waves %llu | cycles/wave: loop %.3g refill %.3g internal %.3g leaf %.3g | iterations/wave %.0f (refill %.0f internal %.0f leaf %.0f)\n",
                    h[12], (double)h[0] / h[12], (double)h[1] / h[12], (double)h[2] / h[12], (double)h[3] / h[12], (double)h[4] / h[12], (double)h[5] / h[12], (double)h[6] / h[12], (double)h[7] / h[12]);
            fprintf(stderr, "[stamps] cycles per phase call: refill %.0f internal %.0f leaf %.0f | lanes per call: refill %.1f internal %.1f leaf %.1f | live lanes/iteration %.1f\n",
                    h[5] ? (double)h[1] / h[5] : 0.0, h[6] ? (double)h[2] / h[6] : 0.0, h[7] ? (double)h[3] / h[7] : 0.0,
                    h[5] ? (double)h[10] / h[5] : 0.0, h[6] ? (double)h[8] / h[6] : 0.0, h[7] ? (double)h[9] / h[7] : 0.0, h[4] ? (double)h[11] / h[4] : 0.0);
            fprintf(stderr, "[stamps] internal phases on one pair (wave-uniform): %.4f of the phases, %.4f of the lane visits; >= 3/4 of the lanes on the first lane's pair: %.4f of the phases\n",
                    h[6] ? (double)h[13] / h[6] : 0.0, h[8] ? (double)h[14] / h[8] : 0.0, h[6] ? (double)h[15] / h[6] : 0.0);
            fprintf(stderr, "[stamps] continuations that entered an internal node (DPairN box re-fetches): %llu = %.4f per internal visit\n", h[16], h[17] ? (double)h[16] / h[17] : 0.0);
            fprintf(stderr, "[stamps] continuation stack: %.3f pushes per ray, %.4f of them leave the %d-entry LDS window for HBM (%.4f per ray)\n",
                    h[20] ? (double)h[19] / h[20] : 0.0, h[19] ? (double)h[18] / h[19] : 0.0, RT_STK, h[20] ? (double)h[18] / h[20] : 0.0);
            (void)hipMemset(g_dbg, 0, sizeof h);
        }
    }
#endif
    if (sc->d_prims) (void)hipFree(sc->d_prims); if (sc->d_mats) (void)hipFree(sc->d_mats); if (sc->d_tri_pre) (void)hipFree(sc->d_tri_pre);
    if (sc->d_nodes) (void)hipFree(sc->d_nodes); if (sc->d_prims_bvh) (void)hipFree(sc->d_prims_bvh); if (sc->d_stack) (void)hipFree(sc->d_stack);
    if (sc->d_pairs32) (void)hipFree(sc->d_pairs32); if (sc->d_prims32) (void)hipFree(sc->d_prims32);
    if (sc->d_pairs64) (void)hipFree(sc->d_pairs64); if (sc->d_spheres) (void)hipFree(sc->d_spheres); if (sc->d_pairsn) (void)hipFree(sc->d_pairsn);
    if (sc->d_cell_aabb) (void)hipFree(sc->d_cell_aabb); if (sc->d_cell_offset) (void)hipFree(sc->d_cell_offset); if (sc->d_prim_list) (void)hipFree(sc->d_prim_list);
    if (sc->d_cell_tri) (void)hipFree(sc->d_cell_tri);
    if (sc->d_cell_chunk) (void)hipFree(sc->d_cell_chunk); if (sc->d_chunks) (void)hipFree(sc->d_chunks); if (sc->d_cell_code) (void)hipFree(sc->d_cell_code);
    sc->ws.release();
    delete sc;
}

int rt_scene_set_options(rt_scene* sc, uint32_t flags) {
    if (!sc) return fail(RT_E_ARG, "null argument");
    if (flags & ~(uint32_t)RT_OPT_DRAW_BOXES) return fail(RT_E_ARG, "unknown option bits 0x%x", flags & ~(uint32_t)RT_OPT_DRAW_BOXES);
    sc->draw_boxes = (flags & RT_OPT_DRAW_BOXES) != 0;
    return RT_OK;
}

int rt_bvh_build(const rt_prim* prims, const double* centroids, uint32_t n_prims,
                 rt_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* order_out, uint32_t* depth_out) {
    return rt_bvh_build_mode(prims, centroids, n_prims, RT_BVH_SAH_POSITION, nodes_out, n_nodes_out, order_out, depth_out);
}

int rt_bvh_build_mode(const rt_prim* prims, const double* centroids, uint32_t n_prims, int mode,
                      rt_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* order_out, uint32_t* depth_out) {
    if (!g_inited) return fail(RT_E_STATE, "rt_init() has not been called");
    if (!prims || !centroids || !nodes_out || !n_nodes_out || !order_out || !depth_out) return fail(RT_E_ARG, "null argument");
    for (uint32_t i = 0; i < n_prims; ++i) if (prims[i].type > RT_TRIANGLE) return fail(RT_E_ARG, "primitive %u has unknown type %u", i, prims[i].type);
    std::string err;
    if (mode < RT_BVH_SAH_POSITION || mode > RT_BVH_NAIVE) return fail(RT_E_ARG, "unknown BVH build mode %d", mode);
    const int rc = rtb::build(prims, centroids, n_prims, mode, nodes_out, n_nodes_out, order_out, depth_out, g_stream, err);
    if (rc != RT_OK) return fail(rc, "%s", err.c_str());
    return RT_OK;
}

int rt_scene_bvh_layout(const rt_scene* sc, uint32_t* pair_bytes, uint32_t* prim_bytes) {
    if (!sc || !sc->have_bvh) return fail(RT_E_STATE, "rt_scene_set_bvh() has not been called");
    const bool persistent = !sc->draw_boxes && g_extend_variant == 2 && !sc->bvh_empty_leaf;
    const bool c32 = persistent && g_compact && sc->lattice_scale > 0.0, sph = persistent && g_compact && sc->d_spheres != nullptr;
    const bool cn = c32 && g_pairn && sc->d_pairsn != nullptr;      // 32 of the record's 64 bytes per visit (the other 32 only when a continuation enters an internal node)
    if (pair_bytes) *pair_bytes = cn ? 32u : c32 ? (uint32_t)sizeof(DPair32) : (persistent ? (uint32_t)sizeof(DPair64) : 2u * (uint32_t)sizeof(DNode));
    if (prim_bytes) *prim_bytes = c32 ? RT_PRIM32_READ_BYTES /* bytes read per test; the table's stride is 64 */ : (sph ? (uint32_t)sizeof(DSphere) : (uint32_t)sizeof(DPrim));
    return RT_OK;
}

int rt_scene_set_bvh(rt_scene* sc, const rt_bvh_node* nodes, uint32_t n_nodes, const uint32_t* prim_index, uint32_t n) {
    if (!sc || !nodes || (n && !prim_index)) return fail(RT_E_ARG, "null argument");
    if (n != sc->n_prims) return fail(RT_E_ARG, "prim_index has %u entries, scene has %u primitives", n, sc->n_prims);
    if (n_nodes == 0) return fail(RT_E_ARG, "empty node array");
    // validate shape on the host: the kernel trusts indices, sibling adjacency and the depth bound
    std::vector<uint8_t> seen(n_nodes, 0);
    std::vector<std::pair<uint32_t, uint32_t>> todo; todo.push_back({0u, 1u});
    uint32_t depth = 0; uint64_t covered = 0; bool empty_leaf = false;
    while (!todo.empty()) {
        auto [i, d] = todo.back(); todo.pop_back();
        if (i >= n_nodes) return fail(RT_E_ARG, "node index %u out of range", i);
        if (seen[i]) return fail(RT_E_ARG, "node %u reachable twice: not a tree", i);
        seen[i] = 1; if (d > depth) depth = d;
        const rt_bvh_node& nd = nodes[i];
        if (nd.is_leaf) {
            if ((uint64_t)nd.first + nd.count > n) return fail(RT_E_ARG, "leaf %u covers [%u,%u) of %u objects", i, nd.first, nd.first + nd.count, n);
            covered += nd.count; if (nd.count == 0) empty_leaf = true;
        } else {
            if (nd.right != nd.left + 1) return fail(RT_E_ARG, "node %u: children %u,%u are not adjacent (bvh.rs:167-168 makes them so)", i, nd.left, nd.right);
            // bvh.rs:167-169 hands out child indices in pairs starting at 1, so a left child's index is odd; the pair tables
            // of the persistent kernel are addressed by (left + 1) / 2 and rely on it
            if ((nd.left & 1u) == 0u) return fail(RT_E_ARG, "node %u: left child %u has an even index (bvh.rs:167-169 numbers children 2k+1, 2k+2)", i, nd.left);
            todo.push_back({nd.right, d + 1}); todo.push_back({nd.left, d + 1});
        }
    }
    for (uint32_t k = 0; k < n; ++k) if (prim_index[k] >= n) return fail(RT_E_ARG, "prim_index[%u] = %u out of range", k, prim_index[k]);
    // device node i+1 = host node i: children of a node are (2k+1, 2k+2) on the host (bvh.rs:167-168), so shifting by one
    // puts every sibling pair on a 128-byte boundary (hipMalloc memory is at least 256-byte aligned)
    std::vector<DNode> dn((size_t)n_nodes + 1);
    memset(&dn[0], 0, sizeof(DNode));
    for (uint32_t i = 0; i < n_nodes; ++i) {
        const rt_bvh_node& s = nodes[i]; DNode& d = dn[(size_t)i + 1];
        for (int c = 0; c < 3; ++c) { d.mn[c] = s.min[c]; d.mx[c] = s.max[c]; }
        if (s.is_leaf) { d.a = s.first; d.b = s.count | RT_LEAF; } else { d.a = s.left + 1; d.b = 0; }
        d.pad0 = d.pad1 = 0;
    }
    std::vector<DPrim> dp(n ? n : 1);
    for (uint32_t k = 0; k < n; ++k) dp[k] = to_dprim(sc->h_prims[prim_index[k]], prim_index[k]);
    if (sc->d_nodes) (void)hipFree(sc->d_nodes); if (sc->d_prims_bvh) (void)hipFree(sc->d_prims_bvh); if (sc->d_stack) (void)hipFree(sc->d_stack);
    if (sc->d_pairs32) (void)hipFree(sc->d_pairs32); if (sc->d_prims32) (void)hipFree(sc->d_prims32);
    if (sc->d_pairs64) (void)hipFree(sc->d_pairs64); if (sc->d_spheres) (void)hipFree(sc->d_spheres); if (sc->d_pairsn) (void)hipFree(sc->d_pairsn);
    sc->d_pairs64 = nullptr; sc->d_spheres = nullptr; sc->d_pairsn = nullptr;
    sc->d_nodes = nullptr; sc->d_prims_bvh = nullptr; sc->d_stack = nullptr; sc->d_pairs32 = nullptr; sc->d_prims32 = nullptr; sc->lattice_scale = 0.0; sc->have_bvh = false;
    HIPCHECK(hipMalloc(&sc->d_nodes, dn.size() * sizeof(DNode)));
    HIPCHECK(hipMalloc(&sc->d_prims_bvh, dp.size() * sizeof(DPrim)));
    HIPCHECK(hipMalloc(&sc->d_stack, (size_t)depth * g_extend_blocks * RT_BLOCK * sizeof(StackEntry)));
    sc->stack_blocks = g_extend_blocks;
    HIPCHECK(hipMemcpy(sc->d_nodes, dn.data(), dn.size() * sizeof(DNode), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(sc->d_prims_bvh, dp.data(), dp.size() * sizeof(DPrim), hipMemcpyHostToDevice));
    // generic sibling pairs (112 B) and, for all-sphere scenes, 48-B spheres
    {
        std::vector<DPair64> pairs(((size_t)n_nodes + 2) / 2);
        for (auto& pr : pairs) memset(&pr, 0, sizeof pr);
        for (size_t p = 0; p < pairs.size(); ++p)
            for (int side = 0; side < 2; ++side) {
                const size_t di = 2 * p + side;
                if (di == 0 || di > n_nodes) continue;
                const DNode& nd = dn[di];
                double* box = pairs[p].box + 6 * side;
                for (int c = 0; c < 3; ++c) { box[c] = nd.mn[c]; box[3 + c] = nd.mx[c]; }
                const uint32_t a = (nd.b & RT_LEAF) ? nd.a : nd.a >> 1;
                if (side == 0) { pairs[p].la = a; pairs[p].lb = nd.b; } else { pairs[p].ra = a; pairs[p].rb = nd.b; }
            }
        // Big trees: a 128-byte stride puts every record in ONE line of the vector L1 (packed, seven of eight straddle two): the dragon
        // mesh with RT_COMPACT=0 914 -> 859 ms extend.  Small trees stay packed: an L1/L2-resident tree only grows (weekend-final from HBM: 108 -> 122 ms).
        sc->pair64_stride = (g_pair_lines && pairs.size() >= g_pair_lines_min) ? 128u : (uint32_t)sizeof(DPair64);
        HIPCHECK(hipMalloc((void**)&sc->d_pairs64, pairs.size() * (size_t)sc->pair64_stride));
        if (sc->pair64_stride == sizeof(DPair64)) HIPCHECK(hipMemcpy(sc->d_pairs64, pairs.data(), pairs.size() * sizeof(DPair64), hipMemcpyHostToDevice));
        else {
            std::vector<unsigned char> wide(pairs.size() * (size_t)sc->pair64_stride, 0);
            for (size_t p = 0; p < pairs.size(); ++p) memcpy(&wide[p * sc->pair64_stride], &pairs[p], sizeof(DPair64));
            HIPCHECK(hipMemcpy(sc->d_pairs64, wide.data(), wide.size(), hipMemcpyHostToDevice));
        }
        sc->n_pairs64 = (uint32_t)pairs.size(); sc->n_bvh_prims = n;
        bool all_spheres = n > 0;
        for (uint32_t k = 0; all_spheres && k < n; ++k) all_spheres = sc->h_prims[prim_index[k]].type == RT_SPHERE;
        if (all_spheres) {
            std::vector<DSphere> sp(n);
            for (uint32_t k = 0; k < n; ++k) {
                const rt_prim& t = sc->h_prims[prim_index[k]];
                sp[k].c[0] = t.v[0]; sp[k].c[1] = t.v[1]; sp[k].c[2] = t.v[2]; sp[k].r = t.v[3];
                sp[k].tm = dp[k].tm; sp[k].orig = dp[k].orig; sp[k].pad0 = sp[k].pad1 = 0;
            }
            HIPCHECK(hipMalloc(&sc->d_spheres, sp.size() * sizeof(DSphere)));
            HIPCHECK(hipMemcpy(sc->d_spheres, sp.data(), sp.size() * sizeof(DSphere), hipMemcpyHostToDevice));
        }
    }
    // compact forms: all triangles, and every box plane and vertex coordinate on one f32 lattice (scale 100 = PLY, or 1)
    for (double S : {100.0, 1.0}) {
        bool ok = n > 0 && !empty_leaf;
        const size_t n_old_pairs = ((size_t)n_nodes + 2) / 2;
        std::vector<uint32_t> perm;                                          // old pair index (device node >> 1) -> slot in the table
        // (tables that fit an XCD's 4 MB L2 several times over gain nothing from it and would only grow by the alignment holes: res4, 710 KB: 35.4 vs 35.1 ms)
        const uint32_t n_slots = pair_line_order(dn, n_nodes, n_old_pairs, g_pair_lines && n_old_pairs >= g_pair_lines_min, perm);
        std::vector<DPair32> pairs(n_slots);
        std::vector<DPrim32> tris(n);
        for (auto& pr : pairs) memset(&pr, 0, sizeof pr);
        for (size_t p = 0; ok && p < n_old_pairs; ++p) {
            if (perm[p] == 0xffffffffu) continue;                            // not reachable from the root (cannot happen for a tree)
            DPair32& rec = pairs[perm[p]];
            for (int side = 0; ok && side < 2; ++side) {
                const size_t di = 2 * p + side;
                if (di == 0 || di > n_nodes) continue;
                const DNode& nd = dn[di];
                float* box = rec.box + 6 * side;
                for (int c = 0; ok && c < 3; ++c) ok = on_lattice(nd.mn[c], S, box[c]) && on_lattice(nd.mx[c], S, box[3 + c]);
                const uint32_t a = (nd.b & RT_LEAF) ? nd.a : perm[nd.a >> 1];   // internal child: slot of ITS children's pair
                if (side == 0) { rec.la = a; rec.lb = nd.b; } else { rec.ra = a; rec.rb = nd.b; }
            }
        }
        for (uint32_t k = 0; ok && k < n; ++k) {
            const rt_prim& t = sc->h_prims[prim_index[k]];
            ok = t.type == RT_TRIANGLE;
            for (int c = 0; ok && c < 9; ++c) ok = on_lattice(t.v[c], S, tris[k].v[c]);
            tris[k].tm = dp[k].tm; tris[k].orig = dp[k].orig; tris[k].pad = 0;
        }
        if (!ok) continue;
        HIPCHECK(hipMalloc(&sc->d_pairs32, pairs.size() * sizeof(DPair32)));
        HIPCHECK(hipMalloc(&sc->d_prims32, tris.size() * sizeof(DPrim32)));
        HIPCHECK(hipMemcpy(sc->d_pairs32, pairs.data(), pairs.size() * sizeof(DPair32), hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(sc->d_prims32, tris.data(), tris.size() * sizeof(DPrim32), hipMemcpyHostToDevice));
        sc->lattice_scale = S;
        // new-planes-only records (DPairN, rt_device.h): every face of a node's box must be one child's plane bit for bit, and the
        // tree must fit the packed child words; otherwise the scene keeps DPair32
        {
            std::vector<DPairN> pn(pairs.size());
            for (auto& r : pn) memset(&r, 0, sizeof r);
            bool okn = pairs.size() < (1u << 24) && n < (1u << 24);
            auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
            auto word = [&](uint32_t a, uint32_t b, bool& good) -> uint32_t {
                if (b & RT_LEAF) { const uint32_t cnt = b & 0x3fffffffu; if (cnt > 15u || a >= (1u << 24)) good = false; return RT_LEAF | (a << 4) | (cnt & 15u); }
                if (a >= (1u << 24)) good = false;
                return a << 4;
            };
            for (size_t di = 1; okn && di <= n_nodes; ++di) {
                const DNode& X = dn[di];
                if (X.b & RT_LEAF) continue;
                const size_t p = perm[X.a >> 1];                        // X's children are device nodes X.a, X.a + 1: their pair's slot
                const DPair32& q = pairs[p]; DPairN& r = pn[p];
                float own[6]; bool lat = true;
                for (int c = 0; c < 3; ++c) lat = lat && on_lattice(X.mn[c], S, own[c]) && on_lattice(X.mx[c], S, own[3 + c]);
                if (!lat) { okn = false; break; }
                uint32_t sel = 0;
                for (int f = 0; f < 6; ++f) {
                    const uint32_t P = bits(own[f]), L = bits(q.box[f]), R = bits(q.box[6 + f]);
                    r.own[f] = P;
                    if (L == P) r.nw[f] = R;                                 // the right child owns the new plane
                    else if (R == P) { r.nw[f] = L; sel |= 1u << f; }        // the left child does
                    else okn = false;
                }
                r.wl = word(q.la, q.lb, okn) | ((sel & 7u) << 28);
                r.wr = word(q.ra, q.rb, okn) | (((sel >> 3) & 7u) << 28);
            }
            if (okn) {
                float ro[6];
                for (int c = 0; c < 3; ++c) okn = okn && on_lattice(dn[1].mn[c], S, ro[c]) && on_lattice(dn[1].mx[c], S, ro[3 + c]);
                for (int f = 0; f < 6; ++f) sc->root_own[f] = bits(ro[f]);
            }
            if (okn) {
                HIPCHECK(hipMalloc(&sc->d_pairsn, pn.size() * sizeof(DPairN)));
                HIPCHECK(hipMemcpy(sc->d_pairsn, pn.data(), pn.size() * sizeof(DPairN), hipMemcpyHostToDevice));
            }
        }
        break;
    }
    for (int c = 0; c < 3; ++c) { sc->root_mn[c] = nodes[0].min[c]; sc->root_mx[c] = nodes[0].max[c]; }
    sc->n_nodes = n_nodes; sc->bvh_depth = depth; sc->bvh_empty_leaf = empty_leaf; sc->have_bvh = true;
    (void)covered;
    return RT_OK;
}

static void install_grid(rt_scene* sc, const rt_grid_desc& gd, uint64_t list_len) {      // device tables are already in sc->d_cell_* / d_prim_list
    DGrid& g = sc->grid;
    g.box_size = tov(gd.box_size); g.origin = tov(gd.origin); g.end = tov(gd.end); g.total_size = tov(gd.total_size);
    g.n_cells = gd.n_cells;
    double tz = gd.total_size[2];                                   /* grid.rs:152: `self.total_size.z() as u32` */
    g.max_depth = !(tz > 0.0) ? 0u : (tz >= 4294967295.0 ? 0xffffffffu : (uint32_t)tz);
    g.cell_aabb = sc->d_cell_aabb; g.cell_offset = sc->d_cell_offset; g.prim_list = sc->d_prim_list;
    if (sc->d_cell_tri) { (void)hipFree(sc->d_cell_tri); sc->d_cell_tri = nullptr; }
    // optional (the walk falls back to the per-triangle plane records): 48 B per list position, at most 8 GiB; only where cells
    // hold many triangles (res4 dragon at 900x506x50: size 25 1 679 -> 1 469 ms, size 5 349 -> 322 ms, size 2 equal, size 1 +4 %)
    if (g_cell_tri && sc->d_tri_pre && list_len && (list_len * 10ull >= gd.n_cells || g_cell_tri_always) && list_len <= (8ull << 30) / sizeof(DCellTri)
        && hipMalloc(&sc->d_cell_tri, list_len * sizeof(DCellTri)) == hipSuccess) {
        hipLaunchKernelGGL(k_cell_tri, dim3((unsigned)((list_len + 255) / 256)), dim3(256), 0, g_stream, sc->d_prim_list, sc->d_tri_pre, list_len, sc->d_cell_tri);
        if (hipStreamSynchronize(g_stream) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(sc->d_cell_tri); sc->d_cell_tri = nullptr; }
    } else (void)hipGetLastError();
    g.cell_tri = sc->d_cell_tri;
    // cell codes (DGrid::cell_code): only for an integer cell size that tiles the world, where slot = x + nx * (y + ny * z) names loop
    // cell (x, y, z); k_grid_codes then checks every cell's stored box against the implicit one, so the walk may trust code 0
    if (sc->d_cell_code) { (void)hipFree(sc->d_cell_code); sc->d_cell_code = nullptr; }
    g.cell_code = nullptr; g.nx = g.ny = g.nz = 0;
    {
        bool ok = g_grid_codes && gd.n_cells < (1ull << 32);
        uint32_t nn[3] = {0, 0, 0};
        for (int c = 0; ok && c < 3; ++c) {
            const double sz = gd.box_size[c], cnt = gd.total_size[c] / sz;
            ok = sz >= 1.0 && sz == floor(sz) && cnt == floor(cnt) && cnt >= 1.0 && cnt < 65536.0 && gd.origin[c] == floor(gd.origin[c]) && fabs(gd.origin[c]) < 1e9;
            nn[c] = ok ? (uint32_t)cnt : 0u;
        }
        ok = ok && (uint64_t)nn[0] * nn[1] * nn[2] == gd.n_cells;
        if (ok && hipMalloc(&sc->d_cell_code, ((gd.n_cells + 15) / 16) * sizeof(uint32_t)) == hipSuccess) {
            g.nx = nn[0]; g.ny = nn[1]; g.nz = nn[2];
            const uint64_t words = (gd.n_cells + 15) / 16;
            hipLaunchKernelGGL(k_grid_codes, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, g_stream, g, sc->d_cell_code);
            if (hipStreamSynchronize(g_stream) == hipSuccess) g.cell_code = sc->d_cell_code;
            else { (void)hipGetLastError(); (void)hipFree(sc->d_cell_code); sc->d_cell_code = nullptr; g.nx = g.ny = g.nz = 0; }
        } else (void)hipGetLastError();
    }
    // cooperative cell test: chunk the lists of the cells that hold at least g_grid_coop_min triangles (optional: any failure
    // leaves the per-lane walk in charge)
    if (sc->d_cell_chunk) { (void)hipFree(sc->d_cell_chunk); sc->d_cell_chunk = nullptr; }
    if (sc->d_chunks) { (void)hipFree(sc->d_chunks); sc->d_chunks = nullptr; }
    sc->n_chunks = 0; g.cell_chunk = nullptr; g.chunks = nullptr;
    if (g_grid_coop_min && sc->d_tri_pre && list_len >= g_grid_coop_min && gd.n_cells < 0xffffffffull) {
        std::vector<uint64_t> off(gd.n_cells + 1);
        if (hipMemcpy(off.data(), sc->d_cell_offset, off.size() * sizeof(uint64_t), hipMemcpyDeviceToHost) == hipSuccess) {
            std::vector<uint32_t> cell_chunk(gd.n_cells, RT_NO_CHUNK), big_cells, big_first;
            uint64_t n_chunks = 0, big_entries = 0;
            for (uint64_t c = 0; c < gd.n_cells; ++c) {
                const uint64_t len = off[c + 1] - off[c];
                if (len >= g_grid_coop_min && len <= 0xffffffffull) { cell_chunk[c] = (uint32_t)n_chunks; big_cells.push_back((uint32_t)c); big_first.push_back((uint32_t)n_chunks); n_chunks += (len + 63) / 64; big_entries += len; }
            }
            // the cooperative kernel pays for its parking rounds with occupancy (3 waves per SIMD instead of 4); where most list entries
            // sit in short lists (fine grids: res4 at size 1 has 3 such cells among 1016 populated ones) the per-lane walk stays in charge
            if (big_entries * g_grid_coop_share_den < list_len * g_grid_coop_share_num) n_chunks = 0;
            uint32_t *d_big = nullptr, *d_first = nullptr;
            if (n_chunks && n_chunks * sizeof(DCellChunk) <= (8ull << 30)
                && hipMalloc(&sc->d_cell_chunk, gd.n_cells * sizeof(uint32_t)) == hipSuccess && hipMalloc(&sc->d_chunks, n_chunks * sizeof(DCellChunk)) == hipSuccess
                && hipMalloc(&d_big, big_cells.size() * sizeof(uint32_t)) == hipSuccess && hipMalloc(&d_first, big_first.size() * sizeof(uint32_t)) == hipSuccess
                && hipMemcpy(sc->d_cell_chunk, cell_chunk.data(), gd.n_cells * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess
                && hipMemcpy(d_big, big_cells.data(), big_cells.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess
                && hipMemcpy(d_first, big_first.data(), big_first.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess) {
                hipLaunchKernelGGL(k_cell_chunks, dim3((unsigned)n_chunks), dim3(64), 0, g_stream, sc->d_cell_offset, sc->d_prim_list, d_big, d_first, (uint32_t)big_cells.size(),
                                   sc->d_prims, sc->d_tri_pre, sc->d_chunks, (uint32_t)n_chunks);
                if (hipStreamSynchronize(g_stream) == hipSuccess) { sc->n_chunks = (uint32_t)n_chunks; g.cell_chunk = sc->d_cell_chunk; g.chunks = sc->d_chunks; }
            }
            if (d_big) (void)hipFree(d_big); if (d_first) (void)hipFree(d_first);
            if (!sc->n_chunks) {
                (void)hipGetLastError();
                if (sc->d_cell_chunk) { (void)hipFree(sc->d_cell_chunk); sc->d_cell_chunk = nullptr; }
                if (sc->d_chunks) { (void)hipFree(sc->d_chunks); sc->d_chunks = nullptr; }
            }
        } else (void)hipGetLastError();
    }
    sc->grid_desc = gd; sc->grid_list_len = list_len;
    sc->have_grid = true;
}

int rt_scene_build_grid(rt_scene* sc, double grid_size) {
    if (!g_inited) return fail(RT_E_STATE, "rt_init() has not been called");
    if (!sc) return fail(RT_E_ARG, "null argument");
    rtb::GridResult r; std::string err;
    const int rc = rtb::build_grid(sc->h_prims.data(), sc->n_prims, grid_size, r, g_stream, err);
    if (rc != RT_OK) return fail(rc, "%s", err.c_str());
    if (sc->d_cell_aabb) (void)hipFree(sc->d_cell_aabb); if (sc->d_cell_offset) (void)hipFree(sc->d_cell_offset); if (sc->d_prim_list) (void)hipFree(sc->d_prim_list);
    sc->d_cell_aabb = r.cell_aabb; sc->d_cell_offset = (uint64_t*)r.cell_offset; sc->d_prim_list = r.prim_list;
    install_grid(sc, r.desc, r.list_len);
    return RT_OK;
}

int rt_scene_get_grid(const rt_scene* sc, rt_grid_desc* desc, uint64_t* list_len, double* cell_aabb, uint64_t* cell_offset, uint32_t* prim_list) {
    if (!sc || !sc->have_grid) return fail(RT_E_STATE, "the scene has no grid");
    if (desc) *desc = sc->grid_desc;
    if (list_len) *list_len = sc->grid_list_len;
    const uint64_t nc = sc->grid_desc.n_cells;
    if (cell_aabb) HIPCHECK(hipMemcpy(cell_aabb, sc->d_cell_aabb, nc * 6 * sizeof(double), hipMemcpyDeviceToHost));
    if (cell_offset) HIPCHECK(hipMemcpy(cell_offset, sc->d_cell_offset, (nc + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (prim_list && sc->grid_list_len) HIPCHECK(hipMemcpy(prim_list, sc->d_prim_list, sc->grid_list_len * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_scene_set_grid(rt_scene* sc, const rt_grid_desc* gd, const double* cell_aabb, const uint64_t* cell_offset, const uint32_t* prim_list) {
    if (!sc || !gd || !cell_aabb || !cell_offset) return fail(RT_E_ARG, "null argument");
    const uint64_t nc = gd->n_cells;
    if (nc == 0) return fail(RT_E_ARG, "grid has no cells");
    if (cell_offset[0] != 0) return fail(RT_E_ARG, "cell_offset[0] must be 0");
    for (uint64_t i = 0; i < nc; ++i) if (cell_offset[i + 1] < cell_offset[i]) return fail(RT_E_ARG, "cell_offset not monotonic at %llu", (unsigned long long)i);
    const uint64_t len = cell_offset[nc];
    if (len && !prim_list) return fail(RT_E_ARG, "null prim_list");
    for (uint64_t k = 0; k < len; ++k) if (prim_list[k] >= sc->n_prims) return fail(RT_E_ARG, "prim_list[%llu] out of range", (unsigned long long)k);
    if (sc->d_cell_aabb) (void)hipFree(sc->d_cell_aabb); if (sc->d_cell_offset) (void)hipFree(sc->d_cell_offset); if (sc->d_prim_list) (void)hipFree(sc->d_prim_list);
    sc->d_cell_aabb = nullptr; sc->d_cell_offset = nullptr; sc->d_prim_list = nullptr; sc->have_grid = false;
    HIPCHECK(hipMalloc(&sc->d_cell_aabb, nc * 6 * sizeof(double)));
    HIPCHECK(hipMalloc(&sc->d_cell_offset, (nc + 1) * sizeof(uint64_t)));
    HIPCHECK(hipMalloc(&sc->d_prim_list, (len ? len : 1) * sizeof(uint32_t)));
    HIPCHECK(hipMemcpy(sc->d_cell_aabb, cell_aabb, nc * 6 * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(sc->d_cell_offset, cell_offset, (nc + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (len) HIPCHECK(hipMemcpy(sc->d_prim_list, prim_list, len * sizeof(uint32_t), hipMemcpyHostToDevice));
    install_grid(sc, *gd, len);
    return RT_OK;
}

int rt_trace(rt_scene* sc, int algorithm, const double* orig, const double* dir, uint64_t n, double tmin, double tmax,
             int32_t* out_prim, double* out_t, double* out_tp, uint8_t* out_hit, rt_stats* stats) {
    if (!g_inited) return fail(RT_E_STATE, "rt_init() has not been called");
    if (!sc || (n && (!orig || !dir))) return fail(RT_E_ARG, "null argument");
    if (algorithm == RT_ALGO_BVH && !sc->have_bvh) return fail(RT_E_STATE, "rt_scene_set_bvh() has not been called");
    if (algorithm == RT_ALGO_GRID && !sc->have_grid) return fail(RT_E_STATE, "rt_scene_set_grid() has not been called");
    if (algorithm < 0 || algorithm > 2) return fail(RT_E_ARG, "unknown algorithm %d", algorithm);
    const uint64_t chunk = 1ull << 22;
    int rc = ensure_workspace(sc, n < chunk ? (n ? n : 1) : chunk, 4);
    if (rc) return rc;
    Workspace& w = sc->ws;
    hipStream_t st = g_stream;
    // staging lives in the parts of the workspace a trace does not use: the second ray queue (interleaved origins and
    // directions of a chunk), the accept-time t array (rec.t) and the second id array (the hit byte)
    double *d_o = w.qbuf[1], *d_d = w.qbuf[1] + 3 * w.cap, *d_t = w.hit_tn; uint8_t* d_h = (uint8_t*)w.qid[1];
    HIPCHECK(hipMemsetAsync(w.stats, 0, ST_N * sizeof(unsigned long long), st));
    int ret = RT_OK;
    for (uint64_t off = 0; off < n && ret == RT_OK; off += chunk) {
        const uint64_t m = (n - off < chunk) ? (n - off) : chunk;
        hipError_t e = hipMemcpyAsync(d_o, orig + 3 * off, m * 3 * sizeof(double), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_d, dir + 3 * off, m * 3 * sizeof(double), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { ret = fail(RT_E_HIP, "ray upload failed: %s", hipGetErrorString(e)); break; }
        hipLaunchKernelGGL(k_load_rays, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d_o, d_d, m, w.q(0));
        ExtendOut eo; eo.prim = w.hit_prim; eo.t = d_t; eo.tp = w.hit_tp; eo.tn = nullptr; eo.hit = d_h; eo.trace = 1;
        if (hipMemsetAsync(w.counts, 0, (size_t)w.depth_cap * 2 * sizeof(uint32_t), st) != hipSuccess) { ret = fail(RT_E_HIP, "memset failed"); break; }
        ret = launch_extend(sc, algorithm, w.q(0), nullptr, m, tmin, tmax, eo, w.counts + w.depth_cap, w.stats, st, m);
        if (ret) break;
        if (out_prim) e = hipMemcpyAsync(out_prim + off, w.hit_prim, m * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && out_t) e = hipMemcpyAsync(out_t + off, d_t, m * sizeof(double), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && out_tp) e = hipMemcpyAsync(out_tp + off, w.hit_tp, m * sizeof(double), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && out_hit) e = hipMemcpyAsync(out_hit + off, d_h, m, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) ret = fail(RT_E_HIP, "trace failed: %s", hipGetErrorString(e));
    }
    if (ret == RT_OK) {
        hipError_t e = hipMemcpy(w.h_stats, w.stats, ST_N * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (e != hipSuccess) ret = fail(RT_E_HIP, "stats readback failed: %s", hipGetErrorString(e));
        else if (stats) { memset(stats, 0, sizeof *stats); fill_stats(stats, w.h_stats); }
    }
    return ret;
}

int rt_generate(const rt_camera* cam, uint64_t seed, uint32_t sample, const uint32_t* pix_i, const uint32_t* pix_j,
                uint64_t n, double* orig, double* dir) {
    if (!g_inited) return fail(RT_E_STATE, "rt_init() has not been called");
    if (!cam || (n && (!pix_i || !pix_j || !orig || !dir))) return fail(RT_E_ARG, "null argument");
    if (n == 0) return RT_OK;
    for (uint64_t k = 0; k < n; ++k)
        if (pix_i[k] >= cam->image_width || pix_j[k] >= cam->image_height) return fail(RT_E_ARG, "pixel %llu outside the image", (unsigned long long)k);
    DCamera dc; dc.W = cam->image_width; dc.H = cam->image_height;
    dc.center = tov(cam->center); dc.pixel00 = tov(cam->pixel00); dc.du = tov(cam->du); dc.dv = tov(cam->dv);
    dc.ddu = tov(cam->ddu); dc.ddv = tov(cam->ddv); dc.defocus_angle = cam->defocus_angle;
    struct Staging {                                         // freed on every exit path
        uint32_t *i = nullptr, *j = nullptr; double *o = nullptr, *d = nullptr;
        ~Staging() { if (i) (void)hipFree(i); if (j) (void)hipFree(j); if (o) (void)hipFree(o); if (d) (void)hipFree(d); }
    } sg;
    HIPCHECK(hipMalloc(&sg.i, n * 4)); HIPCHECK(hipMalloc(&sg.j, n * 4));
    HIPCHECK(hipMalloc(&sg.o, n * 24)); HIPCHECK(hipMalloc(&sg.d, n * 24));
    uint32_t *d_i = sg.i, *d_j = sg.j; double *d_o = sg.o, *d_d = sg.d;
    HIPCHECK(hipMemcpy(d_i, pix_i, n * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_j, pix_j, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_generate_list, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_stream, dc, sample, seed, d_i, d_j, n, d_o, d_d);
    HIPCHECK(hipStreamSynchronize(g_stream));
    HIPCHECK(hipMemcpy(orig, d_o, n * 24, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(dir, d_d, n * 24, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_scene_reserve(rt_scene* sc, uint32_t image_width, uint32_t image_height, uint32_t s_count, uint32_t max_depth) {
    if (!g_inited) return fail(RT_E_STATE, "rt_init() has not been called");
    if (!sc) return fail(RT_E_ARG, "null argument");
    const uint64_t WH = (uint64_t)image_width * image_height;
    if (WH == 0 || WH > 0x7fffffffull) return fail(RT_E_LIMIT, "image of %llu pixels is outside (0, 2^31)", (unsigned long long)WH);
    uint32_t k = 0;
    return size_workspace(sc, WH, s_count, max_depth, &k);
}

int rt_render_device(rt_scene* sc, const rt_camera* cam, int algorithm, uint32_t s_begin, uint32_t s_count, uint32_t max_depth,
                     uint64_t seed, double* d_out_sum, void* stream, rt_stats* stats) {
    if (!g_inited) return fail(RT_E_STATE, "rt_init() has not been called");
    // NULL is the legacy default stream: the library's own stream does not synchronise with it implicitly, so wait here
    if (!stream) HIPCHECK(hipStreamSynchronize(nullptr));
    return render_dynamic(sc, cam, algorithm, s_begin, s_count, max_depth, seed, d_out_sum, stream ? (hipStream_t)stream : g_stream, stats);
}

int rt_render_device_noalloc(rt_scene* sc, const rt_camera* cam, int algorithm, uint32_t s_begin, uint32_t s_count, uint32_t max_depth,
                             uint64_t seed, double* d_out_sum, void* stream, rt_stats* stats) {
    { const int rca = check_render_args(sc, cam, algorithm, max_depth, d_out_sum); if (rca) return rca; }
    const uint64_t WH = (uint64_t)cam->image_width * cam->image_height;
    if (WH > RT_NOALLOC_MAX_PATHS) return fail(RT_E_LIMIT, "rt_render_device_noalloc: image of %llu pixels is above the static workspace's %u paths", (unsigned long long)WH, (unsigned)RT_NOALLOC_MAX_PATHS);
    if (max_depth > RT_NOALLOC_MAX_DEPTH) return fail(RT_E_LIMIT, "rt_render_device_noalloc: max_depth %u is above the static workspace's %u", max_depth, (unsigned)RT_NOALLOC_MAX_DEPTH);
    unsigned char* base = g_dev[g_device].scratch;                  // device address from hipGetSymbolAddress (rt_init), never the array's name
    if (!base) return fail(RT_E_STATE, "rt_init() did not resolve the static workspace on device %d", g_device);
    Workspace w; w.is_static = true;
    const uint64_t cap = RT_NOALLOC_MAX_PATHS;
    w.cap = cap; w.depth_cap = RT_NOALLOC_MAX_DEPTH + 2;
    size_t off = 0;
    auto take = [&](size_t bytes) { unsigned char* p = base + off; off += (bytes + 255) & ~(size_t)255; return p; };
    w.qbuf[0] = (double*)take(cap * 9 * sizeof(double)); w.qbuf[1] = (double*)take(cap * 9 * sizeof(double));
    w.rad = (double*)take(cap * 3 * sizeof(double));
    w.hit_tp = (double*)take(cap * sizeof(double)); w.hit_tn = (double*)take(cap * sizeof(double));
    w.qid[0] = (uint32_t*)take(cap * sizeof(uint32_t)); w.qid[1] = (uint32_t*)take(cap * sizeof(uint32_t));
    w.hit_prim = (int32_t*)take(cap * sizeof(int32_t));
    w.counts = (uint32_t*)take((size_t)w.depth_cap * 2 * sizeof(uint32_t));
    w.stats = (unsigned long long*)take(ST_N * sizeof(unsigned long long));
    if (off > RT_NOALLOC_BYTES) return fail(RT_E_STATE, "static workspace layout (%zu B) exceeds the array (%zu B)", off, (size_t)RT_NOALLOC_BYTES);
    uint64_t k64 = cap / WH; if (s_count && k64 > s_count) k64 = s_count;
    if (s_count) { const uint64_t nb = (s_count + k64 - 1) / k64; k64 = (s_count + nb - 1) / nb; }   // equal batches
    if (!stream) HIPCHECK(hipStreamSynchronize(nullptr));
    // every queue, counter and record of the array that a batch reads is written by that batch first (counts and stats are
    // cleared by render_impl); nothing is kept between calls
    return render_impl(sc, w, (uint32_t)k64, cam, algorithm, s_begin, s_count, max_depth, seed, d_out_sum, stream ? (hipStream_t)stream : g_stream, stats);
}

// ---- RCCL: ncclReduce(sum, root) of the framebuffers over xGMI.  The library is opened on first use.
struct rt_comm { ncclComm_t comm = nullptr; int n_ranks = 0, rank = 0, device = -1; };
namespace {
struct Rccl {
    void* lib = nullptr; bool tried = false; std::string err;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr; decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclReduce) reduce = nullptr; decltype(&ncclCommDestroy) comm_destroy = nullptr; decltype(&ncclGetErrorString) error_string = nullptr;
} g_rccl;
int rccl_load() {
    if (g_rccl.tried) return g_rccl.lib ? RT_OK : fail(RT_E_STATE, "RCCL is not available: %s", g_rccl.err.c_str());
    g_rccl.tried = true;
    const char* names[] = {getenv("RT_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) { if (!nm || !*nm) continue; g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (g_rccl.lib) break; g_rccl.err = dlerror(); }
    if (!g_rccl.lib) return fail(RT_E_STATE, "RCCL is not available: %s", g_rccl.err.c_str());
    g_rccl.get_unique_id = (decltype(g_rccl.get_unique_id))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (decltype(g_rccl.comm_init_rank))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.reduce = (decltype(g_rccl.reduce))dlsym(g_rccl.lib, "ncclReduce");
    g_rccl.comm_destroy = (decltype(g_rccl.comm_destroy))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.error_string = (decltype(g_rccl.error_string))dlsym(g_rccl.lib, "ncclGetErrorString");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.reduce || !g_rccl.comm_destroy || !g_rccl.error_string) {
        g_rccl.err = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclReduce / ncclCommDestroy / ncclGetErrorString";
        dlclose(g_rccl.lib); g_rccl.lib = nullptr;
        return fail(RT_E_STATE, "RCCL is not available: %s", g_rccl.err.c_str());
    }
    return RT_OK;
}
#define RCCLCHECK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail(RT_E_HIP, "%s failed: %s", #x, g_rccl.error_string(r_)); } while (0)
} // namespace

int rt_comm_unique_id(void* id_out) {
    static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "RT_COMM_ID_BYTES must be the size of ncclUniqueId");
    if (!id_out) return fail(RT_E_ARG, "null argument");
    { const int rc = rccl_load(); if (rc) return rc; }
    ncclUniqueId id;
    RCCLCHECK(g_rccl.get_unique_id(&id));
    memcpy(id_out, &id, sizeof id);
    return RT_OK;
}

int rt_comm_create(const void* id, int n_ranks, int rank, rt_comm** out) {
    if (!g_inited) return fail(RT_E_STATE, "rt_init() has not been called");
    if (!id || !out) return fail(RT_E_ARG, "null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(RT_E_ARG, "rank %d of %d", rank, n_ranks);
    { const int rc = rccl_load(); if (rc) return rc; }
    HIPCHECK(hipSetDevice(g_device));
    ncclUniqueId uid; memcpy(&uid, id, sizeof uid);
    ncclComm_t c = nullptr;
    RCCLCHECK(g_rccl.comm_init_rank(&c, n_ranks, uid, rank));
    rt_comm* rc = new rt_comm(); rc->comm = c; rc->n_ranks = n_ranks; rc->rank = rank; rc->device = g_device;
    *out = rc;
    return RT_OK;
}

int rt_comm_reduce_sum(rt_comm* c, double* d_buf, uint64_t count, int root, void* stream) {
    if (!c || !c->comm || (count && !d_buf)) return fail(RT_E_ARG, "null argument");
    if (root < 0 || root >= c->n_ranks) return fail(RT_E_ARG, "root %d of %d ranks", root, c->n_ranks);
    if (c->device != g_device) return fail(RT_E_STATE, "the communicator was created on device %d, the current device is %d", c->device, g_device);
    hipStream_t st = stream ? (hipStream_t)stream : g_stream;
    if (count) RCCLCHECK(g_rccl.reduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum, root, c->comm, st));
    HIPCHECK(hipStreamSynchronize(st));
    return RT_OK;
}

int rt_comm_destroy(rt_comm* c) {
    if (!c) return RT_OK;
    int ret = RT_OK;
    if (c->comm && g_rccl.comm_destroy) { ncclResult_t r = g_rccl.comm_destroy(c->comm); if (r != ncclSuccess) ret = fail(RT_E_HIP, "ncclCommDestroy failed: %s", g_rccl.error_string(r)); }
    delete c;
    return ret;
}

int rt_render_reduce(rt_scene* sc, const rt_camera* cam, int algorithm, uint32_t s_begin, uint32_t s_count, uint32_t max_depth,
                     uint64_t seed, rt_comm* c, int root, double* out_sum, rt_stats* stats) {
    if (!cam || !c) return fail(RT_E_ARG, "null argument");
    if (root < 0 || root >= c->n_ranks) return fail(RT_E_ARG, "root %d of %d ranks", root, c->n_ranks);
    if (c->rank == root && !out_sum) return fail(RT_E_ARG, "the root rank needs out_sum");
    const uint64_t WH = (uint64_t)cam->image_width * cam->image_height;
    double* d_fb = nullptr;
    HIPCHECK(hipMalloc(&d_fb, (WH ? WH : 1) * 3 * sizeof(double)));
    int rc = render_dynamic(sc, cam, algorithm, s_begin, s_count, max_depth, seed, d_fb, g_stream, stats);
    // every rank takes part in the reduce even if its render failed?  No: a failed rank returns, and its peers' reduce would wait
    // for it — the caller must treat any rank's error as fatal for the job (the CLI kills its workers)
    if (rc == RT_OK) rc = rt_comm_reduce_sum(c, d_fb, WH * 3, root, g_stream);
    if (rc == RT_OK && c->rank == root) {
        hipError_t e = hipMemcpy(out_sum, d_fb, WH * 3 * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RT_E_HIP, "framebuffer readback failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_fb);
    return rc;
}

int rt_render(rt_scene* sc, const rt_camera* cam, int algorithm, uint32_t s_begin, uint32_t s_count, uint32_t max_depth,
              uint64_t seed, double* out_sum, rt_stats* stats) {
    if (!cam || !out_sum) return fail(RT_E_ARG, "null argument");
    const uint64_t WH = (uint64_t)cam->image_width * cam->image_height;
    double* d_fb = nullptr;
    HIPCHECK(hipMalloc(&d_fb, (WH ? WH : 1) * 3 * sizeof(double)));
    int rc = render_dynamic(sc, cam, algorithm, s_begin, s_count, max_depth, seed, d_fb, g_stream, stats);
    if (rc == RT_OK) {
        hipError_t e = hipMemcpy(out_sum, d_fb, WH * 3 * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RT_E_HIP, "framebuffer readback failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_fb);
    return rc;
}

} // extern "C"
