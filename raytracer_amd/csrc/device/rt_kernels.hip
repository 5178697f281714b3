// rt_kernels.hip — wavefront path-tracing kernels for gfx950 (MI355X): generate, extend (BVH / grid /
// naive), shade (+ wave64 ballot/prefix-sum compaction into the next queue) and resolve.
//
// Queues are SoA arrays of f64 in HBM; one lane owns one path.  All kernels are launched with a fixed
// grid and stride over `*count` (read from device memory), so the host never has to know a queue length
// to launch, and blocks that share an XCD (blockIdx % 8, observed round-robin placement) take contiguous
// chunks of the queue so that coherent rays meet in the same L2.
#include "rt_device.h"
#include "rt_kernels.h"

// ---------------------------------------------------------------------------------------------
// helpers
RT_DEV uint32_t xcd_chunk(uint32_t b, uint32_t nb) {
    // blocks b and b+8 share an XCD: give each XCD a contiguous range of chunks (bijective for nb % 8 == 0)
    return (nb & 7u) ? b : (b & 7u) * (nb >> 3) + (b >> 3);
}
RT_DEV uint64_t wave_sum(uint64_t v) {
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
RT_DEV void stat_add(unsigned long long* stats, int slot, uint64_t v) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&stats[slot], (unsigned long long)v);
}
// block-level compaction: one atomic per workgroup (a single counter word saturates near 88 atomics/us, so one
// atomic per wave makes a 10^8-path launch atomic-bound).  Every thread of the block must call it, the same number of times.
RT_DEV uint32_t block_compact_slot(bool alive, uint32_t* counter, uint32_t* s_tmp /* [blockDim.x/64] */) {
    const unsigned long long mask = __ballot(alive);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    if (lane == 0) s_tmp[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < n_waves; ++w) total += s_tmp[w];
        uint32_t base = total ? atomicAdd(counter, total) : 0u;
        for (uint32_t w = 0; w < n_waves; ++w) { const uint32_t c = s_tmp[w]; s_tmp[w] = base; base += c; }
    }
    __syncthreads();
    return s_tmp[wave] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// camera.rs:182-184
RT_DEV V3 sky_color(V3 d) {
    V3 unit_direction = unit(d);
    double a = 0.5 * (unit_direction.y + 1.0);
    return (1.0 - a) * v3(1.0, 1.0, 1.0) + a * v3(0.5, 0.7, 1.0);
}

// ---------------------------------------------------------------------------------------------
// generate: Camera::get_ray (camera.rs:132-148), sample_square (:189-191), defocus_disk_sample (:145-148).
// One lane per (pixel, local sample).  A wave holds 2^lg_s samples of each pixel of a 2^lg_tw x 2^lg_th pixel tile
// (lg_s + lg_tw + lg_th = 6; lane = sample * pixels + pixel): with 64 samples of ONE pixel per wave the lanes' primary
// rays walk the same nodes and hit the same few triangles, and their scattered rays start next to each other, so the
// extend kernel's lanes stay in step much longer than with an 8x8 tile of one sample.  Paths are what they were —
// path id = s_local * W*H + (j*W + i); RNG path = (j*W+i, s_base + s_local), stream 0 — only their queue order changes.
// With a BVH the root box test of the first world.hit (hittable_list.rs:84, bvh.rs:182-185) is done here: a primary ray
// that misses it is a sky pixel sample (camera.rs:182-184) and never enters the queue (its box test is counted here).
__global__ __launch_bounds__(RT_SBLOCK) void k_generate(DCamera cam, DRootBox root, uint32_t s_base, uint32_t s_count, uint32_t lg_s, uint64_t seed,
                                                        RayQ out, uint32_t* count_out, double* __restrict__ rad,
                                                        unsigned long long* stats) {
    __shared__ uint32_t s_tmp[RT_SBLOCK / 64];
    __shared__ CoopLds s_coop[RT_SBLOCK / 64];
    const uint32_t W = cam.W, H = cam.H;
    const uint32_t lg_p = 6u - lg_s, lg_tw = (lg_p + 1u) >> 1, lg_th = lg_p >> 1;          // 64 = samples x tile width x tile height
    const uint32_t tiles_x = (W + (1u << lg_tw) - 1u) >> lg_tw, tiles_y = (H + (1u << lg_th) - 1u) >> lg_th;
    const uint32_t n_groups = (s_count + (1u << lg_s) - 1u) >> lg_s;
    const uint64_t n_tiles = (uint64_t)tiles_x * tiles_y, n_waves = n_tiles * n_groups;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = RT_SBLOCK / 64;
    const uint32_t lp = lane & ((1u << lg_p) - 1u);                                        // pixel within the tile; lane >> lg_p = sample within the group
    uint64_t c_culled = 0;
    for (uint64_t wb = (uint64_t)blockIdx.x * wpb; wb < n_waves; wb += (uint64_t)gridDim.x * wpb) {
        const uint64_t w = wb + wave;
        uint32_t s_local = ((uint32_t)(w / n_tiles) << lg_s) + (lane >> lg_p);
        uint32_t tile = (uint32_t)(w % n_tiles);
        uint32_t i = ((tile % tiles_x) << lg_tw) + (lp & ((1u << lg_tw) - 1u)), j = ((tile / tiles_x) << lg_th) + (lp >> lg_tw);
        bool valid = w < n_waves && i < W && j < H && s_local < s_count;
        V3 orig = v3(0, 0, 0), dir = v3(0, 0, 0);
        const uint32_t pix = j * W + i;
        const uint32_t id = s_local * (W * H) + pix;
        // draws 0, 1 = sample_square; from draw 2 on: the tries of random_in_unit_disk, shared out over the wave
        const bool lens = cam.defocus_angle > 0.0;
        double ox = 0.0, oy = 0.0;
        if (valid) { rng_block((uint32_t)seed, (uint32_t)(seed >> 32), pix, s_base + s_local, 0u, 0u, ox, oy); ox = ox - 0.5; oy = oy - 0.5; }
        const V3 disk = coop_reject<2>(valid && lens, (uint32_t)seed, (uint32_t)(seed >> 32), pix, s_base + s_local, 0u, 1u, &s_coop[wave]);
        if (valid) {
            V3 pixel_sample = cam.pixel00 + (((double)i + ox) * cam.du) + (((double)j + oy) * cam.dv);
            if (!lens) {
                orig = cam.center;
            } else {
                orig = cam.center + (disk.x * cam.ddu) + (disk.y * cam.ddv);
            }
            dir = pixel_sample - orig;
            if (root.enabled) {
                V3 inv = v3(1.0 / dir.x, 1.0 / dir.y, 1.0 / dir.z);
                double e0 = 0.001, e1 = __builtin_huge_val();        // camera.rs:161
                if (!slab(root.mn, root.mx, orig, inv, e0, e1)) {
                    V3 c = sky_color(dir);                            // throughput is (1,1,1)
                    double* slot = rad + ((uint64_t)pix * s_count + s_local) * 3u;   // rad[pixel][sample][rgb]: a wave's samples of a pixel are contiguous
                    slot[0] = c.x; slot[1] = c.y; slot[2] = c.z;
                    valid = false; ++c_culled;
                }
            }
        }
        uint32_t q = block_compact_slot(valid, count_out, s_tmp);
        if (valid) {
            out.ox[q] = orig.x; out.oy[q] = orig.y; out.oz[q] = orig.z;
            out.dx[q] = dir.x; out.dy[q] = dir.y; out.dz[q] = dir.z;
            out.tr[q] = 1.0; out.tg[q] = 1.0; out.tb[q] = 1.0;
            out.id[q] = id;
        }
    }
    if (stats && root.enabled) { stat_add(stats, ST_BOX, c_culled); stat_add(stats, ST_SEGMENTS, c_culled); }
}

// parity entry for get_ray: explicit pixel lists, AoS output, no queue
__global__ void k_generate_list(DCamera cam, uint32_t sample, uint64_t seed, const uint32_t* pi, const uint32_t* pj,
                                uint64_t n, double* orig, double* dir) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t i = pi[k], j = pj[k];
    Rng g(seed, j * cam.W + i, sample, 0);
    double ox = g.next() - 0.5;
    double oy = g.next() - 0.5;
    V3 pixel_sample = cam.pixel00 + (((double)i + ox) * cam.du) + (((double)j + oy) * cam.dv);
    V3 o;
    if (cam.defocus_angle <= 0.0) o = cam.center;
    else { V3 p = random_in_unit_disk(g); o = cam.center + (p.x * cam.ddu) + (p.y * cam.ddv); }
    V3 d = pixel_sample - o;
    orig[3 * k] = o.x; orig[3 * k + 1] = o.y; orig[3 * k + 2] = o.z;
    dir[3 * k] = d.x; dir[3 * k + 1] = d.y; dir[3 * k + 2] = d.z;
}

// ---------------------------------------------------------------------------------------------
// extend, common epilogue.  Render mode stores (hit ? prim : -1, tp); trace mode also stores rec.t and the
// boolean, and maps the primitive back to its file-order index.
// `edge` = rec.hits_aabb_edge (draw-boxes only): bit 1 of the trace byte; in render mode it replaces the primitive, because
// ray_color returns red whether or not anything was hit (camera.rs:162-164,178-180).
template <int MODE = 2>    // 0: render format, 1: trace format, 2: decided at run time by out.trace;  +4: no tn (single-type scenes)
RT_DEV void store_hit(const ExtendOut& out, uint64_t q, bool hit, const HitState& rec, const DPrim* prims, bool edge = false) {
    if ((MODE & 3) == 1 || ((MODE & 3) == 2 && out.trace)) {
        if (out.prim) out.prim[q] = rec.prim >= 0 ? (int32_t)prims[rec.prim].orig : -1;
        if (out.t) out.t[q] = rec.t;
        if (out.tp) out.tp[q] = rec.tp;
        if (out.hit) out.hit[q] = (uint8_t)((hit ? 1 : 0) | (edge ? 2 : 0));
    } else {
        out.prim[q] = edge ? RT_HIT_EDGE : (hit ? rec.prim : -1);
        out.tp[q] = rec.tp;
        if (!(MODE & 4) && out.tn) out.tn[q] = rec.tn;
    }
}

// extend_bvh: HittableList::hit, BVH arm (hittable_list.rs:81-93) = BvhNode::hit_aabb on the root
// (bvh.rs:182-185) followed by the ordered recursive BvhNode::hit (bvh.rs:189-233), run as an explicit
// continuation machine.  Recursion state that the reference keeps on the call stack:
//   * the frame's ray_t.max: only ever replaced by rec.t when a `far` child is entered after the `close`
//     child returned true (bvh.rs:215-219); if `close` returned false no frame below it changed it, so one
//     register (`tmax`) suffices and nothing needs saving;
//   * "did close.hit() return true": a subtree returns true iff some primitive test inside it succeeded, so a
//     one-bit flag saved/cleared at push and OR-restored at pop reproduces it;
//   * the pending far child and its entry distance (bvh.rs:210-212): the 16-byte stack entry.
// The stack lives in HBM as stack[level][thread] (coalesced across the wave); `stack_cap` is the tree depth.
template <bool DRAW>
__global__ __launch_bounds__(RT_BLOCK) void k_extend_bvh(const DNode* __restrict__ nodes, const DPrim* __restrict__ prims,
                                                         RayQ in, const uint32_t* __restrict__ count, uint64_t n_fixed,
                                                         double tmin0, double tmax0, ExtendOut out,
                                                         StackEntry* __restrict__ stack, unsigned long long* stats) {
    const uint64_t n = count ? (uint64_t)*count : n_fixed;
    const uint32_t nb = gridDim.x, T = nb * RT_BLOCK;
    const uint32_t tid = xcd_chunk(blockIdx.x, nb) * RT_BLOCK + threadIdx.x;
    StackEntry* my_stack = stack + tid;
    uint64_t c_box = 0, c_prim = 0, c_steps = 0, c_overlap = 0, c_rays = 0;

    for (uint64_t q = tid; q < n; q += T) {
        V3 o = v3(in.ox[q], in.oy[q], in.oz[q]);
        V3 d = v3(in.dx[q], in.dy[q], in.dz[q]);
        V3 inv = v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
        HitState rec; rec.t = 0.0; rec.tp = 0.0; rec.tn = 0.0; rec.prim = -1;
        ++c_rays;

        DNode root = nodes[1];                       // device node i+1 = host node i (pairs 128-byte aligned)
        double tmax = tmax0;
        ++c_box;
        double e0 = tmin0, e1 = tmax0;
        bool go = slab(root.mn, root.mx, o, inv, e0, e1);
        bool edge = false;
        if (DRAW && go) edge = box_is_edge(root.mn, root.mx, o + e0 * d);      // aabb.rs:57-60
        uint32_t cur_a = root.a, cur_b = root.b;
        uint32_t sp = 0;
        bool flag = false, need_pop = false;
        while (go) {
            if (need_pop) {
                if (sp == 0) break;
                --sp;
                StackEntry e = my_stack[(uint64_t)sp * T];
                bool close_hit = flag;
                flag = flag | (e.b >> 30 & 1u);
                if (close_hit) {                         // bvh.rs:213-222
                    if (e.far_d < rec.t) { ++c_overlap; tmax = rec.t; cur_a = e.a; cur_b = e.b & ~0x40000000u; need_pop = false; }
                } else {                                 // bvh.rs:223
                    cur_a = e.a; cur_b = e.b & ~0x40000000u; need_pop = false;
                }
                continue;
            }
            ++c_steps;                                   // bvh.rs:190
            if (cur_b & RT_LEAF) {                       // bvh.rs:195-202
                uint32_t first = cur_a, cnt = cur_b & 0x3fffffffu;
                double closest = tmax;
                for (uint32_t k = 0; k < cnt; ++k) {
                    ++c_prim;
                    DPrim P = prims[first + k];
                    if (hit_prim(P, (int32_t)(first + k), o, d, tmin0, closest, rec)) { flag = true; closest = rec.t; }
                }
                need_pop = true;
            } else {                                     // bvh.rs:203-229
                DNode L = nodes[cur_a], R = nodes[cur_a + 1];
                c_box += 2;
                double ld = tmin0, lx = tmax, rd = tmin0, rx = tmax;
                bool lh = slab(L.mn, L.mx, o, inv, ld, lx);
                bool rh = slab(R.mn, R.mx, o, inv, rd, rx);
                if (DRAW) {
                    if (lh && box_is_edge(L.mn, L.mx, o + ld * d)) edge = true;
                    if (rh && box_is_edge(R.mn, R.mx, o + rd * d)) edge = true;
                }
                if (lh && rh) {
                    bool right_first = ld >= rd;         // bvh.rs:210-211 (tie -> right is `close`)
                    StackEntry e;
                    e.far_d = fmax(ld, rd);
                    e.a = right_first ? L.a : R.a;
                    e.b = (right_first ? L.b : R.b) | (flag ? 0x40000000u : 0u);
                    my_stack[(uint64_t)sp * T] = e;
                    ++sp;
                    flag = false;
                    cur_a = right_first ? R.a : L.a; cur_b = right_first ? R.b : L.b;
                } else if (lh) { cur_a = L.a; cur_b = L.b; }
                else if (rh) { cur_a = R.a; cur_b = R.b; }
                else need_pop = true;
            }
        }
        store_hit(out, q, rec.prim >= 0, rec, prims, edge);
    }
    if (stats) {
        stat_add(stats, ST_BOX, c_box); stat_add(stats, ST_PRIM, c_prim); stat_add(stats, ST_STEPS, c_steps);
        stat_add(stats, ST_OVERLAP, c_overlap); stat_add(stats, ST_SEGMENTS, c_rays);
    }
}
template __global__ void k_extend_bvh<false>(const DNode*, const DPrim*, RayQ, const uint32_t*, uint64_t, double, double, ExtendOut, StackEntry*, unsigned long long*);
template __global__ void k_extend_bvh<true>(const DNode*, const DPrim*, RayQ, const uint32_t*, uint64_t, double, double, ExtendOut, StackEntry*, unsigned long long*);

// extend_bvh, persistent form: the same continuation machine as k_extend_bvh (every decision, counter and
// result is identical), restructured for a wave64 machine whose lanes finish at very different times:
//   * a wave keeps its 64 lanes fed: when enough lanes (tune bits 16..23) are idle it pulls fresh rays for them from the
//     queue (one atomic per chunk of rays per wave), instead of waiting for its slowest lane;
//   * lanes on internal nodes and lanes on leaves / waiting to pop are served by separate phases; the wave votes which
//     one to run, so each phase's instructions execute with as many lanes as possible;
//   * the continuation stack's top RT_STK entries live in LDS ([level][lane]: conflict-free for b64 and b32
//     accesses whatever level each lane is on); deeper entries spill to HBM as stack[level][thread].
#define RT_STK 8
// Diagnostic build only (-DRT_STAMPS): per-wave cycle sums per phase of the loop, added into dbg[] at the end.
// dbg: 0 loop cycles, 1 refill cycles, 2 internal cycles, 3 leaf cycles, 4 iterations, 5 refill iterations, 6 internal
// iterations, 7 leaf iterations, 8 lanes served by internal phases, 9 lanes served by leaf phases, 10 lanes refilled,
// 11 live lanes summed over iterations, 12 waves, 13 internal phases whose lanes all stand on ONE pair (what a scalar-load
// path could serve), 14 lanes in those, 15 internal phases where at least 3/4 of the lanes stand on the first lane's pair;
// per-lane sums: 16 own-box re-reads (DPairN), 17 internal visits, 18 stack entries moved from the LDS window to HBM, 19 pushes, 20 rays
#ifdef RT_STAMPS
#define RT_STAMP_DECL unsigned long long tk0 = 0, tk_ref = 0, tk_int = 0, tk_leaf = 0, n_it = 0, n_ref = 0, n_intp = 0, n_leafp = 0, l_int = 0, l_leaf = 0, l_ref = 0, l_live = 0, n_uni = 0, l_uni = 0, n_uni2 = 0, n_refetch = 0, n_spill = 0, n_push = 0; const unsigned long long tk_begin = clock64();
#define RT_STAMP(...) __VA_ARGS__
#else
#define RT_STAMP_DECL
#define RT_STAMP(...)
#endif
// wave64 ballot of a predicate that already lives in a lane mask: one s_and with exec (HIP's __ballot(int) first
// materialises the bool as 0/1 in a VGPR and compares it again)
RT_DEV unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
RT_DEV uint32_t popc64(unsigned long long m) { return (uint32_t)__builtin_popcountll(m); }

RT_DEV double lattice(uint32_t f32_bits, double scale) { return (double)__uint_as_float(f32_bits) * scale; }
// plane - o for a lattice plane: the product is exact (rt_device.h), so the fma rounds exactly as `plane - o` does
RT_DEV double lattice_minus(uint32_t f32_bits, double scale, double o) { return __builtin_fma((double)__uint_as_float(f32_bits), scale, -o); }

#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 4
#endif
// LDS = true (small scenes with generic pairs, NODES == 0): the whole pair table and primitive table are staged in LDS by a
// 1024-thread workgroup (one per CU), so a visit's seven 16-byte reads go to LDS (128 B/clk/CU) instead of the vector L1
// (about one divergent lane access per clock); the continuation stack keeps RT_LSTK entries per lane there.
template <int NODES, int PRIMS, bool TRACE, bool LDS>
__global__ __launch_bounds__(LDS ? RT_LBLOCK : RT_BLOCK, RT_WAVES_PER_EU) void k_extend_bvh2(BvhView bvh, RayQ in, const uint32_t* __restrict__ count, uint64_t n_fixed,
                                                          double tmin0, double tmax0, ExtendOut out,
                                                          StackEntry* __restrict__ spill, uint32_t* fetch_ctr, uint32_t tune, unsigned long long* stats,
                                                          unsigned long long* dbg) {
    RT_STAMP_DECL
    constexpr int SH_MODE = (TRACE ? 1 : 0) | (PRIMS != 0 ? 4 : 0);      // single-type scenes never read rec.tn
    const uint32_t cost_leaf = tune & 0xffu, cost_int = (tune >> 8) & 0xffu;
    const uint32_t refill_min = ((tune >> 16) & 0xffu) > 64u ? 64u : ((tune >> 16) & 0xffu);   // a fully idle wave always refills
    static_assert(!LDS || NODES == 0, "the LDS variant stages generic pairs");
    constexpr uint32_t BLOCK = LDS ? RT_LBLOCK : RT_BLOCK, STK = LDS ? RT_LSTK : RT_STK;
    __shared__ uint4 s_stk_fixed[LDS ? 1 : RT_STK][LDS ? 1 : RT_BLOCK];       // {far_d lo, far_d hi, a, b | flag}: one 16-byte access per push / pop
    extern __shared__ uint4 s_dyn[];                // LDS variant: [RT_LSTK][RT_LBLOCK] stack words, then the pairs, then the primitives
    auto stk = [&](uint32_t level) -> uint4& {
        if constexpr (LDS) return s_dyn[(level & (STK - 1)) * BLOCK + threadIdx.x]; else return s_stk_fixed[level & (STK - 1)][threadIdx.x];
    };
    const uint32_t n = count ? *count : (uint32_t)n_fixed;
    // one lane per ray is all a short queue can use: surplus blocks leave before they touch the fetch counter (each wave's
    // atomic on that one word costs more than the traversal of a few hundred rays); block 0 always stays
    if (blockIdx.x != 0 && (uint64_t)blockIdx.x * BLOCK >= n) return;
    const uint32_t T = gridDim.x * BLOCK, tx = threadIdx.x, lane = tx & 63;
    StackEntry* my_spill = spill + (blockIdx.x * BLOCK + tx);
    const uint4* const s_pairs = s_dyn + STK * BLOCK;
    const uint4* const s_prims = s_pairs + bvh.lds_pairs16;
    if constexpr (LDS) {
        uint4* dst = s_dyn + STK * BLOCK;
        const uint4* src = (const uint4*)bvh.pairs64;                       // records of pair64_stride bytes in HBM, their 7 words staged packed
        const uint32_t sw = bvh.pair64_stride >> 4;
        for (uint32_t i = tx; i < bvh.lds_pairs16; i += BLOCK) dst[i] = src[(i / 7u) * sw + i % 7u];
        dst += bvh.lds_pairs16;
        src = PRIMS == 2 ? (const uint4*)bvh.spheres : (const uint4*)bvh.prims;
        for (uint32_t i = tx; i < bvh.lds_prims16; i += BLOCK) dst[i] = src[i];
        __syncthreads();
    }
    uint32_t chunk = n / ((T >> 6) * 4u);
    chunk = chunk < 64u ? 64u : (chunk > 512u ? 512u : chunk);
    uint32_t c_next = 0, c_end = 0;          // wave-uniform: the wave's current chunk of the queue
    bool exhausted = false;
    bool look_for_rays = true;               // wave-uniform: lanes went idle since the last refill decision
    // counters: box tests = 2 per internal visit + 1 per ray (root); traversal steps = internal visits + leaves
    uint32_t c_int = 0, c_leaf = 0, c_prim = 0, c_overlap = 0, c_rays = 0;

    // A lane's state is the `b` word of the node it stands on (no separate flags: every vote is one compare):
    //   0               on an internal node, cur_a = its children (internal nodes carry b == 0)
    //   RT_LEAF | k     on a leaf with k >= 1 primitives still to test, cur_a = the next one
    //   RT_LEAF         nothing left here: pop the continuation stack (a finished leaf counts down to exactly this)
    //   RT_IDLE         no ray
    constexpr uint32_t RT_IDLE = 1u, RT_POP = RT_LEAF;
    bool regular = true, all_regular = true;
    uint32_t q = 0, cur_a = 0, cur_b = RT_IDLE, sp = 0, lo = 0;
    bool flag = false;
    V3 o = v3(0, 0, 0), d = v3(0, 0, 0), inv = v3(0, 0, 0);
    double tmax = tmax0;
    HitState rec; rec.t = 0.0; rec.tp = 0.0; rec.tn = 0.0; rec.prim = -1;
    const DNode* __restrict__ nodes = bvh.nodes; const DPrim* __restrict__ prims = bvh.prims;
    const DNode root = nodes[1];                 // device node i+1 = host node i (pairs 128-byte aligned)
    const uint32_t root_a = (root.b & RT_LEAF) ? root.a : root.a >> 1;       // internal: index of its children's pair
    bool need_box = false;                       // NODES == 2: pbox must be read from the record at the next visit
    uint32_t pbox[6] = {0, 0, 0, 0, 0, 0};       // NODES == 2: the box of the node the lane stands on, f32 bits per face (DPairN)
    const double scale = bvh.scale;

    // Rays are taken from the queue 64 at a time into registers (lane i holds ray pf_base + i) while the wave is still
    // traversing; idle lanes then pick theirs up with a lane permute.  A refill serves ~11 lanes: loading, dividing and
    // testing the root box for just those would run at 17 % lane utilisation and stall the wave on HBM every time.
    V3 P_o = v3(0, 0, 0), P_d = v3(1, 1, 1), P_inv = v3(1, 1, 1);
    uint32_t pf_base = 0, pf_n = 0, pf_pos = 0;           // wave-uniform: the batch, and how much of it has been handed out
    unsigned long long pf_hit = 0, pf_reg = 0;            // per batch ray: passed the root box / regular (see slab_fast_d)
    bool pf_ready = true;
    auto issue_batch = [&]() {
        if (c_next >= c_end) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(fetch_ctr, chunk);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= n) { exhausted = true; c_next = c_end = n; }
            else { c_next = base; c_end = (n - base < chunk) ? n : base + chunk; }
        }
        pf_base = c_next; pf_n = (c_end - c_next < 64u) ? c_end - c_next : 64u; pf_pos = 0; c_next += pf_n;
        if (lane < pf_n) {
            const uint32_t pq = pf_base + lane;
            P_o = v3(in.ox[pq], in.oy[pq], in.oz[pq]);
            P_d = v3(in.dx[pq], in.dy[pq], in.dz[pq]);
        }
        pf_ready = pf_n == 0;
    };

    for (;;) {
        // ---- refill idle lanes (looked at only after lanes have finished: nothing else changes the idle set)
        RT_STAMP(tk0 = clock64(); ++n_it;)
        if (look_for_rays) {
            const unsigned long long idle = ballot64(cur_b == RT_IDLE);
            const uint32_t n_idle = popc64(idle);
            if (!(exhausted && pf_pos >= pf_n) && n_idle >= refill_min) {
                RT_STAMP(++n_ref;)
                if (pf_pos >= pf_n) issue_batch();                       // first batch only: later ones are issued when the previous runs out
                if (!pf_ready) {
                    // finish the batch with all 64 lanes: 1/d, the root box test (bvh.rs:84-85, 182-185) and its misses' results
                    const bool mine = lane < pf_n;
                    P_inv = v3(1.0 / P_d.x, 1.0 / P_d.y, 1.0 / P_d.z);
                    double e0 = tmin0, e1 = tmax0;
                    const bool rh = mine && slab(root.mn, root.mx, P_o, P_inv, e0, e1);
                    pf_hit = ballot64(rh);
                    pf_reg = ballot64(mine && ray_is_regular(P_o, P_inv));
                    if (mine) {
                        ++c_rays;
                        if (!rh) { HitState none; none.t = 0.0; none.tp = 0.0; none.tn = 0.0; none.prim = -1; store_hit<SH_MODE>(out, pf_base + lane, false, none, prims); }
                    }
                    pf_ready = true;
                }
                const uint32_t avail = pf_n - pf_pos;
                const uint32_t rank = popc64(idle & ((1ull << lane) - 1ull));
                const bool take = cur_b == RT_IDLE && rank < avail;
                const uint32_t src = take ? pf_pos + rank : lane;
                const V3 to = v3(__shfl(P_o.x, src, 64), __shfl(P_o.y, src, 64), __shfl(P_o.z, src, 64));
                const V3 td = v3(__shfl(P_d.x, src, 64), __shfl(P_d.y, src, 64), __shfl(P_d.z, src, 64));
                const V3 ti = v3(__shfl(P_inv.x, src, 64), __shfl(P_inv.y, src, 64), __shfl(P_inv.z, src, 64));
                if (take) {
                    q = pf_base + src; o = to; d = td; inv = ti;
                    regular = (pf_reg >> src) & 1ull;
                    rec.t = 0.0; rec.tp = 0.0; rec.tn = 0.0; rec.prim = -1;
                    tmax = tmax0; sp = 0; lo = 0; flag = false;
                    if ((pf_hit >> src) & 1ull) { cur_a = root_a; cur_b = root.b; }      // a root miss was answered when its batch was finished
                    if (NODES == 2) {
                        need_box = false;
#pragma unroll
                        for (int f = 0; f < 6; ++f) pbox[f] = bvh.root_own[f];
                    }
                }
                const uint32_t taken = (n_idle < avail) ? n_idle : avail;
                pf_pos += taken;
                if (pf_pos >= pf_n && !exhausted) issue_batch();         // in flight while the wave traverses
                // lanes are still idle if the batch ran out mid-refill or handed out root misses: look again next iteration
                look_for_rays = !(exhausted && pf_pos >= pf_n) && ballot64(cur_b == RT_IDLE) != 0 && (taken < n_idle || ballot64(take && cur_b == RT_IDLE) != 0);
                all_regular = ballot64(cur_b != RT_IDLE && !regular) == 0;     // min/max box test is exact only for regular rays
                RT_STAMP(l_ref += taken; tk_ref += clock64() - tk0;)
            } else {
                look_for_rays = false;
            }
            if (ballot64(cur_b != RT_IDLE) == 0) { if (exhausted && pf_pos >= pf_n) break; look_for_rays = true; continue; }
        }

        // ---- phase vote.  A lane is either on an internal node (wants the two-box visit) or on a leaf / waiting
        // to pop (wants the primitive test + stack work).  Running both every iteration leaves most lanes masked
        // in one of them; instead the wave runs the phase that serves more lanes per instruction and the others wait.
        // Internal visits come in runs, so they get a loop of their own: only their own state is carried around it.
        for (;;) {
            const uint32_t n_leaf = popc64(ballot64((int32_t)cur_b < 0));
            const uint32_t n_int = popc64(ballot64(cur_b == 0u));
            RT_STAMP(l_live += n_int + n_leaf; tk0 = clock64();)
            if (!(n_leaf == 0 || (n_int != 0 && n_int * cost_leaf >= n_leaf * cost_int))) break;   // never starve: one side is non-empty
            // ---- one internal-node visit (bvh.rs:203-229), branch-free apart from the push
            RT_STAMP({ const bool on_int = cur_b == 0u; const unsigned long long mi = ballot64(on_int);
                       const uint32_t ua = (uint32_t)__shfl((int)cur_a, (int)__builtin_ctzll(mi), 64); const unsigned long long df = ballot64(on_int && cur_a != ua);
                       n_uni += df == 0 ? 1 : 0; l_uni += df == 0 ? n_int : 0; n_uni2 += popc64(df) * 4u <= n_int ? 1 : 0; })
            if (cur_b == 0u) {
                ++c_int;                                 // bvh.rs:190 (and two box tests, bvh.rs:183)
                double dLmn[3], dLmx[3], dRmn[3], dRmx[3];          // box planes minus the ray origin (aabb.rs:36-37)
                uint32_t La, Lb, Ra, Rb;
                uint32_t nLp[6], nRp[6];                 // NODES == 2: the children's planes as f32 bits, kept for the next level's pbox
                if (NODES == 2) {                        // 32-byte record of new planes: two 16-byte accesses
                    const uint4* pp = (const uint4*)(bvh.pairsn + cur_a);
                    const uint4 r0 = pp[0], r1 = pp[1];
                    if (need_box) {                      // entered through a popped continuation: the node's own box rides in the same record
                        const uint4 b0 = pp[2], b1 = pp[3];      // (same cache line, same round trip)
                        pbox[0] = b0.x; pbox[1] = b0.y; pbox[2] = b0.z; pbox[3] = b0.w; pbox[4] = b1.x; pbox[5] = b1.y;
                        need_box = false;
                        RT_STAMP(++n_refetch;)
                    }
                    const uint32_t nw[6] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y};
#pragma unroll
                    for (int f = 0; f < 6; ++f) {
                        const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)(f < 3 ? r1.z : r1.w), 28 + f % 3, 1);   // all ones: the left child owns the new plane
                        nLp[f] = (m & nw[f]) | (~m & pbox[f]);                                                  // v_bfi_b32
                        nRp[f] = (m & pbox[f]) | (~m & nw[f]);
                    }
                    dLmn[0] = lattice_minus(nLp[0], scale, o.x); dLmn[1] = lattice_minus(nLp[1], scale, o.y); dLmn[2] = lattice_minus(nLp[2], scale, o.z);
                    dLmx[0] = lattice_minus(nLp[3], scale, o.x); dLmx[1] = lattice_minus(nLp[4], scale, o.y); dLmx[2] = lattice_minus(nLp[5], scale, o.z);
                    dRmn[0] = lattice_minus(nRp[0], scale, o.x); dRmn[1] = lattice_minus(nRp[1], scale, o.y); dRmn[2] = lattice_minus(nRp[2], scale, o.z);
                    dRmx[0] = lattice_minus(nRp[3], scale, o.x); dRmx[1] = lattice_minus(nRp[4], scale, o.y); dRmx[2] = lattice_minus(nRp[5], scale, o.z);
                    La = __builtin_amdgcn_ubfe(r1.z, 4, 24); Lb = r1.z & 0x8000000fu; Ra = __builtin_amdgcn_ubfe(r1.w, 4, 24); Rb = r1.w & 0x8000000fu;
                } else if (NODES == 1) {                 // 64-byte pair: four 16-byte accesses
                    const uint4* pp = (const uint4*)(bvh.pairs32 + cur_a);
                    const uint4 q0 = pp[0], q1 = pp[1], q2 = pp[2], q3 = pp[3];
                    dLmn[0] = lattice_minus(q0.x, scale, o.x); dLmn[1] = lattice_minus(q0.y, scale, o.y); dLmn[2] = lattice_minus(q0.z, scale, o.z);
                    dLmx[0] = lattice_minus(q0.w, scale, o.x); dLmx[1] = lattice_minus(q1.x, scale, o.y); dLmx[2] = lattice_minus(q1.y, scale, o.z);
                    dRmn[0] = lattice_minus(q1.z, scale, o.x); dRmn[1] = lattice_minus(q1.w, scale, o.y); dRmn[2] = lattice_minus(q2.x, scale, o.z);
                    dRmx[0] = lattice_minus(q2.y, scale, o.x); dRmx[1] = lattice_minus(q2.z, scale, o.y); dRmx[2] = lattice_minus(q2.w, scale, o.z);
                    La = q3.x; Lb = q3.y; Ra = q3.z; Rb = q3.w;
#ifdef RT_PROBE_L1
                    // sensitivity probe (never in the product build): RT_PROBE_L1 more 16-byte lane accesses per visit, from the previous
                    // record (another cache line half of the time); the value is consumed by an empty asm so the load stays
                    for (int e = 0; e < RT_PROBE_L1; ++e) { uint4 x = pp[-1 - e]; /* the previous record (pair indices start at 1): in bounds */ asm volatile("" :: "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w)); }
#endif
#ifdef RT_PROBE_L1SAME
                    // sensitivity probe: RT_PROBE_L1SAME more 16-byte lane accesses per visit to the record just read (the same cache line:
                    // L1 hits by construction) -- separates the cost of a lane access from the cost of touching another line
                    for (int e = 0; e < RT_PROBE_L1SAME; ++e) { typedef uint32_t u32x4 __attribute__((ext_vector_type(4))); const u32x4 x = *(const volatile __attribute__((address_space(1))) u32x4*)(uintptr_t)(pp + (3 - (e & 3))); asm volatile("" :: "v"(x)); }
#endif
#ifdef RT_PROBE_VALU
                    // sensitivity probe: RT_PROBE_VALU more 32-bit VALU instructions per visit (the v_bfe_i32 / v_bfi_b32 / v_cndmask
                    // mix a 48-byte "new planes + selector bits" record would need to rebuild the two boxes)
                    { uint32_t j0 = q3.x, j1 = q0.x, j2 = q1.x;
#pragma unroll
                      for (int e = 0; e < RT_PROBE_VALU / 3; ++e)
                          asm volatile("v_bfe_i32 %0, %1, 3, 1\n\tv_bfi_b32 %1, %0, %2, %1\n\tv_bfi_b32 %2, %0, %1, %2" : "+v"(j0), "+v"(j1), "+v"(j2));
                      asm volatile("" :: "v"(j0), "v"(j1), "v"(j2)); }
#endif
                } else {                                 // 112-byte pair: seven accesses
                    const double2* pp;
                    if constexpr (LDS) pp = (const double2*)(s_pairs + cur_a * 7u); else pp = (const double2*)((const char*)bvh.pairs64 + (size_t)cur_a * bvh.pair64_stride);
                    const double2 b0 = pp[0], b1 = pp[1], b2 = pp[2], b3 = pp[3], b4 = pp[4], b5 = pp[5];
                    const uint4 m = *(const uint4*)(pp + 6);
                    dLmn[0] = b0.x - o.x; dLmn[1] = b0.y - o.y; dLmn[2] = b1.x - o.z;
                    dLmx[0] = b1.y - o.x; dLmx[1] = b2.x - o.y; dLmx[2] = b2.y - o.z;
                    dRmn[0] = b3.x - o.x; dRmn[1] = b3.y - o.y; dRmn[2] = b4.x - o.z;
                    dRmx[0] = b4.y - o.x; dRmx[1] = b5.x - o.y; dRmx[2] = b5.y - o.z;
                    La = m.x; Lb = m.y; Ra = m.z; Rb = m.w;
                }
                double ld = tmin0, lx = tmax, rd = tmin0, rx = tmax;
                bool lh, rh;
                if (all_regular) slab_fast_d2(dLmn, dLmx, dRmn, dRmx, inv, ld, lx, rd, rx, lh, rh);
                else { lh = slab_d(dLmn, dLmx, inv, ld, lx); rh = slab_d(dRmn, dRmx, inv, rd, rx); }
                const bool both = lh && rh;
                const bool go_right = rh && (!lh || ld >= rd); // bvh.rs:210-211 (tie -> right is `close`), :226-227; as lane-mask logic
                if (both) {
                    if (sp - lo == STK) {                // LDS window full: its oldest entry goes to HBM
                        *(uint4*)&my_spill[(uint64_t)lo * T] = stk(lo); ++lo;     // same layout as StackEntry
                        RT_STAMP(++n_spill;)
                    }
                    const double far_d = ld >= rd ? ld : rd;    // bvh.rs:212 f64::max of two non-NaN values
                    stk(sp) = make_uint4((uint32_t)__double2loint(far_d), (uint32_t)__double2hiint(far_d), go_right ? La : Ra,
                                                              (go_right ? Lb : Rb) | (flag ? 0x40000000u : 0u));
                    ++sp; flag = false;
                    RT_STAMP(++n_push;)
                }
                cur_a = go_right ? Ra : La;
                cur_b = (lh || rh) ? (go_right ? Rb : Lb) : RT_POP;
                if (NODES == 2) {
#pragma unroll
                    for (int f = 0; f < 6; ++f) pbox[f] = go_right ? nRp[f] : nLp[f];
                }
            }
            RT_STAMP(++n_intp; l_int += n_int; tk_int += clock64() - tk0;)
        }

        // ---- one primitive test of the leaf (bvh.rs:195-202), then pops until the lane has a node again or its
        // ray is finished.  One primitive per call: 96 % of the leaves hold one primitive, and a lane that has more
        // keeps its place (cur_a / count in cur_b advance) and joins the next leaf phase instead of making the wave loop.
        // closest_so_far of the leaf (bvh.rs:193,200) lives in tmax: once a primitive is accepted the frame's old
        // tmax is dead (every later entry into a node goes through a pop with close_hit set, which reloads it).
        bool done = false;
        RT_STAMP(const uint32_t st_leaf = popc64(ballot64((int32_t)cur_b < 0)); tk0 = clock64();)
        if ((int32_t)cur_b < 0) {
            if (cur_b != RT_POP) {
                ++c_prim;
                bool acc;
                if (PRIMS == 1) {                    // 48-byte triangle: three 16-byte accesses
                    const uint4* tp = (const uint4*)(bvh.prims32 + cur_a);
                    const uint4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                    DPrim P;
                    P.v[0] = lattice(q0.x, scale); P.v[1] = lattice(q0.y, scale); P.v[2] = lattice(q0.z, scale);
                    P.v[3] = lattice(q0.w, scale); P.v[4] = lattice(q1.x, scale); P.v[5] = lattice(q1.y, scale);
                    P.v[6] = lattice(q1.z, scale); P.v[7] = lattice(q1.w, scale); P.v[8] = lattice(q2.x, scale);
                    acc = hit_triangle_flat(P, (int32_t)cur_a, o, d, tmin0, tmax, rec);
                } else if (PRIMS == 2) {             // 48-byte sphere: three accesses (the third only for parity of layout)
                    const double2* sp2;
                    if constexpr (LDS) sp2 = (const double2*)(s_prims + cur_a * 3u); else sp2 = (const double2*)(bvh.spheres + cur_a);
                    const double2 c0 = sp2[0], c1 = sp2[1];
                    DPrim P; P.v[0] = c0.x; P.v[1] = c0.y; P.v[2] = c1.x; P.v[3] = c1.y;
                    acc = hit_sphere(P, (int32_t)cur_a, o, d, tmin0, tmax, rec);
                } else {                             // 80-byte primitive: five accesses
                    const double2* gp;
                    if constexpr (LDS) gp = (const double2*)(s_prims + cur_a * 5u); else gp = (const double2*)(prims + cur_a);
                    const double2 g0 = gp[0], g1 = gp[1], g2 = gp[2], g3 = gp[3];
                    const uint4 g4 = *(const uint4*)(gp + 4);
                    DPrim P; P.v[0] = g0.x; P.v[1] = g0.y; P.v[2] = g1.x; P.v[3] = g1.y; P.v[4] = g2.x; P.v[5] = g2.y; P.v[6] = g3.x; P.v[7] = g3.y;
                    P.v[8] = __hiloint2double((int)g4.y, (int)g4.x); P.tm = g4.z; P.orig = g4.w;
                    acc = hit_prim_flat(P, (int32_t)cur_a, o, d, tmin0, tmax, rec);
                }
                flag = flag || acc; tmax = acc ? rec.t : tmax;
                ++cur_a; --cur_b;                    // leaves hold >= 1 primitive (checked by rt_scene_set_bvh)
                c_leaf += (cur_b == RT_POP) ? 1u : 0u;   // bvh.rs:190, once per leaf
            }
            // one continuation: `far` is entered if `close` found nothing (bvh.rs:223) or its box starts before the hit (:213-222)
            auto resume = [&](const uint4 se) {
                const double far_d = __hiloint2double((int)se.y, (int)se.x);
                const bool close_hit = flag;
                flag = flag || ((se.w >> 30) & 1u);
                const bool overlap = close_hit && far_d < rec.t;
                if (overlap) { ++c_overlap; tmax = rec.t; }
                if (overlap || !close_hit) { cur_a = se.z; cur_b = se.w & ~0x40000000u; }
            };
            // After a hit most pending far children lie behind it and are dropped one after the other; each drop used to be a
            // loop iteration of the whole wave with an LDS round trip.  The top RT_POPW entries are read at once instead.
            if (cur_b == RT_POP) {
                uint4 w0 = stk(sp - 1u), w1 = stk(sp - 2u);
                uint4 w2 = stk(sp - 3u), w3 = stk(sp - 4u);
                asm volatile("" : "+v"(w0.x), "+v"(w0.y), "+v"(w0.z), "+v"(w0.w), "+v"(w1.x), "+v"(w1.y), "+v"(w1.z), "+v"(w1.w));
                asm volatile("" : "+v"(w2.x), "+v"(w2.y), "+v"(w2.z), "+v"(w2.w), "+v"(w3.x), "+v"(w3.y), "+v"(w3.z), "+v"(w3.w));
                if (sp > lo) { --sp; resume(w0); }                        // entries [lo, sp) are the ones in LDS
                if (cur_b == RT_POP && sp > lo) { --sp; resume(w1); }
                if (cur_b == RT_POP && sp > lo) { --sp; resume(w2); }
                if (cur_b == RT_POP && sp > lo) { --sp; resume(w3); }
            }
            while (cur_b == RT_POP) {
                if (sp == 0) { done = true; cur_b = RT_IDLE; }
                else {
                    --sp;
                    uint4 se = stk(sp);
                    asm volatile("" : "+v"(se.x), "+v"(se.y), "+v"(se.z), "+v"(se.w));      // keep this an LDS read (not a flat load of a selected pointer)
                    if (sp < lo) { lo = sp; se = *(const uint4*)&my_spill[(uint64_t)sp * T]; }
                    resume(se);
                }
            }
            if (NODES == 2 && cur_b == 0u) need_box = true;     // a continuation entered an internal node: the lane no longer holds its box
            if (done) store_hit<SH_MODE>(out, q, rec.prim >= 0, rec, prims);
        }
        look_for_rays = ballot64(done) != 0;
        RT_STAMP(++n_leafp; l_leaf += st_leaf; tk_leaf += clock64() - tk0;)
    }
#ifdef RT_STAMPS
    if (dbg && lane == 0) {
        const unsigned long long v[16] = {clock64() - tk_begin, tk_ref, tk_int, tk_leaf, n_it, n_ref, n_intp, n_leafp, l_int, l_leaf, l_ref, l_live, 1, n_uni, l_uni, n_uni2};
        for (int i = 0; i < 16; ++i) atomicAdd(&dbg[i], v[i]);
    }
    if (dbg) {   // per-lane counts (every lane adds its own)
        const unsigned long long rf = wave_sum(n_refetch), ci = wave_sum(c_int);
        const unsigned long long ns = wave_sum(n_spill), np = wave_sum(n_push), nr = wave_sum((unsigned long long)c_rays);
        if (lane == 0) { atomicAdd(&dbg[16], rf); atomicAdd(&dbg[17], ci); atomicAdd(&dbg[18], ns); atomicAdd(&dbg[19], np); atomicAdd(&dbg[20], nr); }
    }
#endif
    if (stats) {
        stat_add(stats, ST_BOX, (uint64_t)c_int * 2u + c_rays); stat_add(stats, ST_PRIM, c_prim); stat_add(stats, ST_STEPS, (uint64_t)c_int + c_leaf);
        stat_add(stats, ST_OVERLAP, c_overlap); stat_add(stats, ST_SEGMENTS, c_rays);
    }
}
#define RT_INST_BVH2(N, P, T) template __global__ void k_extend_bvh2<N, P, T, false>(BvhView, RayQ, const uint32_t*, uint64_t, double, double, ExtendOut, StackEntry*, uint32_t*, uint32_t, unsigned long long*, unsigned long long*);
RT_INST_BVH2(0, 0, false) RT_INST_BVH2(0, 0, true) RT_INST_BVH2(0, 2, false) RT_INST_BVH2(0, 2, true) RT_INST_BVH2(1, 1, false) RT_INST_BVH2(1, 1, true) RT_INST_BVH2(2, 1, false) RT_INST_BVH2(2, 1, true)
#define RT_INST_BVH2L(P, T) template __global__ void k_extend_bvh2<0, P, T, true>(BvhView, RayQ, const uint32_t*, uint64_t, double, double, ExtendOut, StackEntry*, uint32_t*, uint32_t, unsigned long long*, unsigned long long*);
RT_INST_BVH2L(0, false) RT_INST_BVH2L(0, true) RT_INST_BVH2L(2, false) RT_INST_BVH2L(2, true)

// extend_naive: HittableList::hit, naive arm (hittable_list.rs:67-80): every object in file order.
// per-triangle constants of hit_triangle_pre (all-triangle scenes only)
__global__ void k_tri_pre(const DPrim* __restrict__ prims, uint32_t n, DTriPre* __restrict__ pre) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DPrim tr = prims[i];
    V3 v0 = v3(tr.v[0], tr.v[1], tr.v[2]), v1 = v3(tr.v[3], tr.v[4], tr.v[5]), v2 = v3(tr.v[6], tr.v[7], tr.v[8]);
    V3 v0v1 = v1 - v0;
    V3 v0v2 = v2 - v0;
    V3 nn = cross(v0v1, v0v2);
    DTriPre q; q.n[0] = nn.x; q.n[1] = nn.y; q.n[2] = nn.z; q.dd = -dot(nn, v0);
    pre[i] = q;
}

// per list position of the grid: plane record + index (DCellTri), gathered once when the grid is installed
__global__ void k_cell_tri(const uint32_t* __restrict__ prim_list, const DTriPre* __restrict__ pre, uint64_t n, DCellTri* __restrict__ out) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t pi = prim_list[k];
    const DTriPre q = pre[pi];
    DCellTri c; c.n[0] = q.n[0]; c.n[1] = q.n[1]; c.n[2] = q.n[2]; c.dd = q.dd; c.pi = pi; c.pad[0] = c.pad[1] = c.pad[2] = 0;
    out[k] = c;
}

__global__ __launch_bounds__(RT_BLOCK) void k_extend_naive(const DPrim* __restrict__ prims, const DTriPre* __restrict__ pre, uint32_t n_prims,
                                                           RayQ in, const uint32_t* __restrict__ count, uint64_t n_fixed,
                                                           double tmin0, double tmax0, ExtendOut out, unsigned long long* stats) {
    const uint64_t n = count ? (uint64_t)*count : n_fixed;
    const uint32_t nb = gridDim.x, T = nb * RT_BLOCK;
    const uint32_t tid = xcd_chunk(blockIdx.x, nb) * RT_BLOCK + threadIdx.x;
    uint64_t c_prim = 0, c_rays = 0;
    for (uint64_t q = tid; q < n; q += T) {
        V3 o = v3(in.ox[q], in.oy[q], in.oz[q]);
        V3 d = v3(in.dx[q], in.dy[q], in.dz[q]);
        HitState rec; rec.t = 0.0; rec.tp = 0.0; rec.tn = 0.0; rec.prim = -1;
        double closest = tmax0;
        ++c_rays;
        if (pre) {
            for (uint32_t k = 0; k < n_prims; ++k) {
                ++c_prim;
                if (hit_triangle_pre(prims, pre, k, o, d, tmin0, closest, rec)) closest = rec.t;
            }
        } else {
            for (uint32_t k = 0; k < n_prims; ++k) {
                ++c_prim;
                DPrim P = prims[k];
                if (hit_prim(P, (int32_t)k, o, d, tmin0, closest, rec)) closest = rec.t;
            }
        }
        store_hit(out, q, rec.prim >= 0, rec, prims);
    }
    if (stats) { stat_add(stats, ST_PRIM, c_prim); stat_add(stats, ST_SEGMENTS, c_rays); }
}

// ---- grid helpers: Rust's saturating float->int casts and Grid::get_index (grid.rs:64-73)
RT_DEV uint64_t sat_usize(double x) {
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551616.0) return ~0ull;
    return (uint64_t)x;
}
RT_DEV int32_t sat_i32(double x) {
    if (x != x) return 0;
    if (x <= -2147483648.0) return (int32_t)0x80000000;
    if (x >= 2147483647.0) return 0x7fffffff;
    return (int32_t)x;
}
RT_DEV uint64_t grid_index(V3 p, const DGrid& g) {
    uint64_t x = sat_usize(p.x / g.box_size.x);
    uint64_t y = sat_usize(p.y / g.box_size.y);
    uint64_t z = sat_usize(p.z / g.box_size.z);
    double size_x = g.total_size.x / g.box_size.x;
    double size_y = g.total_size.y / g.box_size.y;
    return sat_usize((double)x + (double)y * size_x + (double)z * size_x * size_y);
}
// get_index with the three per-axis cell numbers it computes on the way; `ok` = they are inside the grid (then, and only then,
// index == x + nx * (y + ny * z) names the loop cell (x, y, z) and the implicit box below can be used for a code-0 cell)
RT_DEV uint64_t grid_index_xyz(V3 p, const DGrid& g, uint32_t& cx, uint32_t& cy, uint32_t& cz, bool& ok) {
    uint64_t x = sat_usize(p.x / g.box_size.x);
    uint64_t y = sat_usize(p.y / g.box_size.y);
    uint64_t z = sat_usize(p.z / g.box_size.z);
    double size_x = g.total_size.x / g.box_size.x;
    double size_y = g.total_size.y / g.box_size.y;
    ok = x < g.nx && y < g.ny && z < g.nz;
    cx = (uint32_t)x; cy = (uint32_t)y; cz = (uint32_t)z;
    return sat_usize((double)x + (double)y * size_x + (double)z * size_x * size_y);
}
RT_DEV uint64_t grid_cell_from_point_xyz(V3 p, const DGrid& g, uint32_t& cx, uint32_t& cy, uint32_t& cz, bool& ok) {
    int32_t dx = (int32_t)((uint32_t)sat_i32(floor(p.x)) - (uint32_t)sat_i32(g.origin.x));
    int32_t dy = (int32_t)((uint32_t)sat_i32(floor(p.y)) - (uint32_t)sat_i32(g.origin.y));
    int32_t dz = (int32_t)((uint32_t)sat_i32(floor(p.z)) - (uint32_t)sat_i32(g.origin.z));
    double x = floor((double)dx / g.box_size.x) * g.box_size.x;
    double y = floor((double)dy / g.box_size.y) * g.box_size.y;
    double z = floor((double)dz / g.box_size.z) * g.box_size.z;
    return grid_index_xyz(v3(floor(x), floor(y), floor(z)), g, cx, cy, cz, ok);
}
// The box of cell (x, y, z) and whether its list is empty, from the code table where it allows, from the tables otherwise.
// Implicit box: min = origin + i * size (the loop coordinate `x as f64` of grid.rs:36-39 for an integer size), max = min + size
// (GridBox::new, grid.rs:261-264); k_grid_codes compared exactly this expression with the stored box of every cell.
RT_DEV bool grid_cell_box(const DGrid& g, uint64_t cell, uint32_t cx, uint32_t cy, uint32_t cz, bool xyz_ok, double* mn, double* mx, bool& empty) {
    uint32_t code = 3u;
    if (g.cell_code) code = (g.cell_code[cell >> 4] >> (((uint32_t)cell & 15u) * 2u)) & 3u;
    if (xyz_ok && !(code & 2u)) {
        mn[0] = g.origin.x + (double)cx * g.box_size.x; mn[1] = g.origin.y + (double)cy * g.box_size.y; mn[2] = g.origin.z + (double)cz * g.box_size.z;
        mx[0] = mn[0] + g.box_size.x; mx[1] = mn[1] + g.box_size.y; mx[2] = mn[2] + g.box_size.z;
    } else {
        const double* cb = g.cell_aabb + cell * 6;
        mn[0] = cb[0]; mn[1] = cb[1]; mn[2] = cb[2]; mx[0] = cb[3]; mx[1] = cb[4]; mx[2] = cb[5];
    }
    empty = g.cell_code && !(code & 1u);
    return xyz_ok && !(code & 2u);
}
// Grid::traverse's step to the neighbour cell (grid.rs:170-181) when the current cell's box is the implicit one: next_xyz - origin is
// then exactly (x +- 1) * size on the exit axis, the quotients of get_index (grid.rs:64-73) are exactly the integers x +- 1, y, z,
// and its f64 index expression is exact integer arithmetic — as long as the new number stays inside [0, n) (at x == n the
// reference's index runs into the next row; that case takes the literal code).  Saves three f64 divisions per step.
RT_DEV bool grid_step_regular(const DGrid& g, int ex_axis, int ex_dir, uint32_t& cx, uint32_t& cy, uint32_t& cz, uint64_t& cell) {
    uint32_t& c = ex_axis == 0 ? cx : (ex_axis == 1 ? cy : cz);
    const uint32_t lim = ex_axis == 0 ? g.nx : (ex_axis == 1 ? g.ny : g.nz);
    const uint32_t nc = c + (uint32_t)ex_dir;                              // ex_dir is +1 or -1; c == 0 with -1 wraps to 0xffffffff >= lim
    if (nc >= lim) return false;
    c = nc;
    cell = (uint64_t)cx + (uint64_t)g.nx * ((uint64_t)cy + (uint64_t)g.ny * cz);
    return true;
}
// one thread per 16 cells: the code word (DGrid::cell_code)
__global__ void k_grid_codes(DGrid g, uint32_t* __restrict__ codes) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w * 16u >= g.n_cells) return;
    uint32_t word = 0;
    for (uint32_t k = 0; k < 16u; ++k) {
        const uint64_t c = w * 16u + k;
        if (c >= g.n_cells) break;
        const uint32_t cx = (uint32_t)(c % g.nx), cy = (uint32_t)((c / g.nx) % g.ny), cz = (uint32_t)(c / ((uint64_t)g.nx * g.ny));
        const double imn[3] = {g.origin.x + (double)cx * g.box_size.x, g.origin.y + (double)cy * g.box_size.y, g.origin.z + (double)cz * g.box_size.z};
        const double imx[3] = {imn[0] + g.box_size.x, imn[1] + g.box_size.y, imn[2] + g.box_size.z};
        const double* cb = g.cell_aabb + c * 6;
        bool same = cz < g.nz;
        for (int a = 0; a < 3; ++a)
            same = same && __double_as_longlong(cb[a]) == __double_as_longlong(imn[a]) && __double_as_longlong(cb[3 + a]) == __double_as_longlong(imx[a]);
        const uint32_t code = (g.cell_offset[c + 1] > g.cell_offset[c] ? 1u : 0u) | (same ? 0u : 2u);
        word |= code << (k * 2u);
    }
    codes[w] = word;
}
RT_DEV uint64_t grid_cell_from_point(V3 p, const DGrid& g) {            // grid.rs:195-206 (i32 subtraction wraps)
    int32_t dx = (int32_t)((uint32_t)sat_i32(floor(p.x)) - (uint32_t)sat_i32(g.origin.x));
    int32_t dy = (int32_t)((uint32_t)sat_i32(floor(p.y)) - (uint32_t)sat_i32(g.origin.y));
    int32_t dz = (int32_t)((uint32_t)sat_i32(floor(p.z)) - (uint32_t)sat_i32(g.origin.z));
    double x = floor((double)dx / g.box_size.x) * g.box_size.x;
    double y = floor((double)dy / g.box_size.y) * g.box_size.y;
    double z = floor((double)dz / g.box_size.z) * g.box_size.z;
    return grid_index(v3(floor(x), floor(y), floor(z)), g);
}

// extend_grid: HittableList::hit, grid arm (hittable_list.rs:94-100) = Grid::hit (grid.rs:139-147):
// get_box_enter (:119-129) then the tail-recursive traverse (:151-183) as a loop.
// The walk is written as a loop over cell steps with a refill point: when RT_GRID_REFILL lanes of the wave are idle they take the
// next rays of the queue (one atomic per refill).  Measured on the res4 dragon, 900x506x50 (round 3): refilling early LOSES —
// size 1: 142 ms at 64 (= a fresh batch only when every lane is done), 153 at 32, 172 at 16, 198 at 8; size 2: 145 vs 226 at 16;
// weekend-final size 5: 55 vs 74 — rays that enter a wave mid-walk stand in unrelated cells, so the box / list loads of a step stop
// sharing cache lines and every step pays the set-up code for a few lanes.  (Lane utilisation at 64 is 0.28; it is the price of
// coherent loads here.)  The default is therefore 64: batches of 64 consecutive rays handed out through the counter.
#ifndef RT_GRID_REFILL
#define RT_GRID_REFILL 64
#endif
template <bool DRAW>
__global__ __launch_bounds__(RT_BLOCK) void k_extend_grid(DGrid g, const DPrim* __restrict__ prims, const DTriPre* __restrict__ pre,
                                                          RayQ in, const uint32_t* __restrict__ count, uint64_t n_fixed,
                                                          double tmin0, double tmax0, ExtendOut out, uint32_t* fetch_ctr, unsigned long long* stats) {
    const uint64_t n = count ? (uint64_t)*count : n_fixed;
    const uint32_t lane = threadIdx.x & 63;
    const double INF = __builtin_huge_val();
    uint64_t c_box = 0, c_prim = 0, c_steps = 0, c_gbox = 0, c_rays = 0;
    bool active = false, exhausted = false;                        // exhausted is wave-uniform
    uint64_t q = 0;
    V3 o = v3(0, 0, 0), d = v3(1, 1, 1), inv = v3(1, 1, 1);
    HitState rec; rec.t = 0.0; rec.tp = 0.0; rec.tn = 0.0; rec.prim = -1;
    bool hit = false, edge = false;
    uint32_t cx = 0, cy = 0, cz = 0, depth = 0; bool cok = false;  // the cell as (x, y, z) when get_index's per-axis numbers are in range
    uint64_t cell = 0;
    for (;;) {
        // ---- refill
        const unsigned long long idle = ballot64(!active);
        const uint32_t n_idle = popc64(idle);
        if (!exhausted && n_idle >= RT_GRID_REFILL) {
            uint32_t base32 = 0;
            if (lane == 0) base32 = atomicAdd(fetch_ctr, n_idle);
            const uint64_t qbase = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)base32);
            if (qbase + n_idle >= n) exhausted = true;
            const uint64_t myq = qbase + popc64(idle & ((1ull << lane) - 1ull));
            if (!active && myq < n) {
                q = myq;
                o = v3(in.ox[q], in.oy[q], in.oz[q]);
                d = v3(in.dx[q], in.dy[q], in.dz[q]);
                inv = v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
                rec.t = 0.0; rec.tp = 0.0; rec.tn = 0.0; rec.prim = -1;
                hit = false; edge = false; depth = 0;
                ++c_rays;
                // get_box_enter
                cell = grid_cell_from_point_xyz(o, g, cx, cy, cz, cok);
                bool have = cell < g.n_cells;
                if (!have) {
                    // AABB::enter_and_exit (aabb.rs:65-88) on the outer box with Interval(0.001, inf); `<=` per axis
                    double mn[3] = {g.origin.x, g.origin.y, g.origin.z}, mx[3] = {g.end.x, g.end.y, g.end.z};
                    double oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
                    double a = 0.001, b = INF; bool ok = true;
                    for (int ax = 0; ax < 3; ++ax) {
                        double t0 = (mn[ax] - oo[ax]) * ii[ax], t1 = (mx[ax] - oo[ax]) * ii[ax];
                        if (t0 < t1) { if (t0 > a) a = t0; if (t1 < b) b = t1; }
                        else { if (t1 > a) a = t1; if (t0 < b) b = t0; }
                        if (b <= a) { ok = false; break; }
                    }
                    if (ok) {
                        V3 point = o + d * a;                                  // grid.rs:132-136
                        cell = grid_cell_from_point_xyz(point, g, cx, cy, cz, cok);
                        have = cell < g.n_cells;
                    }
                }
                if (have) active = true;
                else store_hit(out, q, false, rec, prims, false);             // never entered the grid
            }
        }
        if (ballot64(active) == 0) { if (exhausted) break; continue; }
        // ---- one cell of the traverse (grid.rs:151-183) for every lane that holds a ray
        if (active) {
            bool done = false;
            if (depth > g.max_depth) done = true;                      // grid.rs:152
            else {
                ++c_steps;
                double mn[3], mx[3]; bool empty;
                const bool implicit_box = grid_cell_box(g, cell, cx, cy, cz, cok, mn, mx, empty);
                // AABB::hit with Interval::universe (aabb.rs:28-62) keeping the Exit
                double oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
                double a = -INF, b = INF; int ex_axis = 0, ex_dir = 0; bool ok = true;
                for (int ax = 0; ax < 3; ++ax) {
                    double t0 = (mn[ax] - oo[ax]) * ii[ax], t1 = (mx[ax] - oo[ax]) * ii[ax];
                    if (t0 < t1) { if (t0 > a) a = t0; if (t1 < b) { b = t1; ex_axis = ax; ex_dir = 1; } }
                    else { if (t1 > a) a = t1; if (t0 < b) { b = t0; ex_axis = ax; ex_dir = -1; } }
                    if (b < a) { ok = false; break; }
                }
                if (!ok) done = true;                                  // grid.rs:155-156
                else {
                    if (DRAW && box_is_edge(mn, mx, o + a * d)) edge = true;   // aabb.rs:57-60, on the real rec (grid.rs:155)
                    // GridBox::hit on a copy of rec (grid.rs:160-167, 281-294)
                    ++c_gbox;
                    HitState copy = rec;
                    bool any = false;
                    double closest = tmax0;
                    uint64_t lo = 0, hi = 0;
                    if (!empty) { lo = g.cell_offset[cell]; hi = g.cell_offset[cell + 1]; }
                    if (g.cell_tri) {
                        for (uint64_t k = lo; k < hi; ++k) {
                            ++c_prim;
                            const DCellTri ct = g.cell_tri[k];
                            if (hit_triangle_cell(prims, ct, o, d, tmin0, closest, copy)) { any = true; closest = copy.t; }
                        }
                    } else if (pre) {
                        for (uint64_t k = lo; k < hi; ++k) {
                            ++c_prim;
                            uint32_t pi = g.prim_list[k];
                            if (hit_triangle_pre(prims, pre, pi, o, d, tmin0, closest, copy)) { any = true; closest = copy.t; }
                        }
                    } else {
                        for (uint64_t k = lo; k < hi; ++k) {
                            ++c_prim;
                            uint32_t pi = g.prim_list[k];
                            DPrim P = prims[pi];
                            if (hit_prim(P, (int32_t)pi, o, d, tmin0, closest, copy)) { any = true; closest = copy.t; }
                        }
                    }
                    if (any) {
                        rec = copy;
                        if (rec.t <= b) { hit = true; done = true; }
                    }
                    if (!done) {
                        ++c_box;                                               // grid.rs:172
                        V3 step = v3(0, 0, 0);                                 // aabb.rs:161-165
                        double sv = (double)ex_dir * (ex_axis == 0 ? g.box_size.x : (ex_axis == 1 ? g.box_size.y : g.box_size.z));
                        if (ex_axis == 0) step.x = sv; else if (ex_axis == 1) step.y = sv; else step.z = sv;
                        V3 next_xyz = v3(mn[0], mn[1], mn[2]) + step;
                        bool inside = g.origin.x <= next_xyz.x && g.end.x >= next_xyz.x && g.origin.y <= next_xyz.y && g.end.y >= next_xyz.y
                                   && g.origin.z <= next_xyz.z && g.end.z >= next_xyz.z;   // grid.rs:186-192
                        if (!inside) done = true;
                        else {
                            if (!(implicit_box && grid_step_regular(g, ex_axis, ex_dir, cx, cy, cz, cell)))
                                cell = grid_index_xyz(next_xyz - g.origin, g, cx, cy, cz, cok);   // grid.rs:76-83
                            if (cell >= g.n_cells) done = true;
                            ++depth;
                        }
                    }
                }
            }
            if (done) { store_hit(out, q, hit, rec, prims, edge); active = false; }
        }
    }
    if (stats) {
        stat_add(stats, ST_BOX, c_box); stat_add(stats, ST_PRIM, c_prim); stat_add(stats, ST_STEPS, c_steps);
        stat_add(stats, ST_GRIDBOX, c_gbox); stat_add(stats, ST_SEGMENTS, c_rays);
    }
}
template __global__ void k_extend_grid<false>(DGrid, const DPrim*, const DTriPre*, RayQ, const uint32_t*, uint64_t, double, double, ExtendOut, uint32_t*, unsigned long long*);
template __global__ void k_extend_grid<true>(DGrid, const DPrim*, const DTriPre*, RayQ, const uint32_t*, uint64_t, double, double, ExtendOut, uint32_t*, unsigned long long*);

// ---------------------------------------------------------------------------------------------
// extend_grid, cooperative form (all-triangle scenes whose grid has well-filled cells).  The cell walk of Grid::traverse
// (grid.rs:151-183) stays one ray per lane; what changes is GridBox::hit (grid.rs:281-294) for a cell with a long list.  The
// per-lane loop walks such a list with every lane at its own entry (divergent loads) and runs the edge tests whenever any
// lane's triangle passes the plane test, i.e. always, with a third of the lanes (lane utilisation 0.27-0.33, 60-70 % of the wave
// cycles waiting: profiles/round2_res4_grid*_summary.md).  Here a lane that reaches such a cell PARKS; when no lane of the wave
// can walk on, the parked rays are served one after the other by the WHOLE wave: the ray is broadcast (readlane -> scalar
// registers), lane l tests entry l of the current 64-entry chunk of the list (coalesced SoA loads, DCellChunk), and the
// reference's sequential semantics are re-established per chunk by a short scalar loop:
//   * closest_so_far only changes when a triangle is ACCEPTED (grid.rs:288-291), and only the `n.d < closest` half of the plane
//     test reads it (triangle.rs:44-47), so every lane evaluates its entry once and the acceptance chain is resolved in list
//     order on ballots: entries up to the first accepted one are final under the current closest, the rest are re-compared
//     (one v_cmp) under the new one;
//   * rec.t is the t of the LAST entry that passed the plane test, rec.p's parameter that of the last one that also had t >= 0
//     (triangle.rs:52,59 write before acceptance), the primitive / normal those of the last accepted entry.
// Every counter is what the per-lane walk counts (one primitive test per list entry).
#ifndef RT_COOP_WAVES
#define RT_COOP_WAVES 4       // waves per SIMD the cooperative kernel is compiled for: 128 VGPRs with 43 spilled beat 161 VGPRs at 3 waves (res4 grid 25: 536 -> 503 ms, grid 5 equal)
#endif
RT_DEV double rl_f64(double x, uint32_t k) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), (int)k), __builtin_amdgcn_readlane(__double2loint(x), (int)k));
}
// one thread per (chunk, entry): gather the list entry's plane record, vertices and index
__global__ void k_cell_chunks(const uint64_t* __restrict__ cell_offset, const uint32_t* __restrict__ prim_list, const uint32_t* __restrict__ big_cells,
                              const uint32_t* __restrict__ big_first, uint32_t n_big, const DPrim* __restrict__ prims, const DTriPre* __restrict__ pre,
                              DCellChunk* __restrict__ chunks, uint32_t n_chunks) {
    const uint32_t c = blockIdx.x, l = threadIdx.x;                       // 64 threads per block
    if (c >= n_chunks) return;
    uint32_t lo = 0, hi = n_big;                                         // last big cell whose first chunk is <= c
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (big_first[mid] <= c) lo = mid; else hi = mid; }
    const uint32_t cell = big_cells[lo];
    const uint64_t e = cell_offset[cell] + (uint64_t)(c - big_first[lo]) * 64u + l, end = cell_offset[cell + 1];
    DCellChunk& ch = chunks[c];
    if (e < end) {
        const uint32_t pi = prim_list[e];
        const DTriPre q = pre[pi]; const DPrim tr = prims[pi];
        ch.n[0][l] = q.n[0]; ch.n[1][l] = q.n[1]; ch.n[2][l] = q.n[2]; ch.dd[l] = q.dd;
        for (int k = 0; k < 9; ++k) ch.v[k][l] = tr.v[k];
        ch.pi[l] = pi;
    } else {
        ch.n[0][l] = 0.0; ch.n[1][l] = 0.0; ch.n[2][l] = 0.0; ch.dd[l] = 0.0;
        for (int k = 0; k < 9; ++k) ch.v[k][l] = 0.0;
        ch.pi[l] = 0u;
    }
    ch.pad[l] = 0u;
}

__global__ __launch_bounds__(RT_BLOCK, RT_COOP_WAVES) void k_extend_grid_coop(DGrid g, const DPrim* __restrict__ prims, const DTriPre* __restrict__ pre,
                                                               RayQ in, const uint32_t* __restrict__ count, uint64_t n_fixed,
                                                               double tmin0, double tmax0, ExtendOut out, uint32_t* fetch_ctr, unsigned long long* stats) {
    __shared__ double2 s_ray[RT_BLOCK / 64][3][64];                     // origins and directions of the wave's parked rays
    const uint64_t n = count ? (uint64_t)*count : n_fixed;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double INF = __builtin_huge_val();
    uint64_t c_box = 0, c_prim = 0, c_steps = 0, c_gbox = 0, c_rays = 0;
    // 64 consecutive rays per pass, handed out through a counter: a ray that crosses a well-filled cell costs thousands of times
    // what a ray through empty cells costs, and a static split left a third of the wave slots idle (1.5 - 2.0 of 3 resident waves
    // per SIMD on the res4 dragon, SQ_WAVE_CYCLES)
    // A short queue (the last ~45 bounces of a frame carry a few thousand rays each) is spread over the waves of the grid in
    // batches of fewer than 64 rays: a wave serves its parked rays one after the other, so 64 rays in one wave are a chain of
    // hundreds of dependent chunk loads while most of the chip idles (res4 size 5: 0.8-2 ms per such launch).
    const uint64_t n_waves_grid = (uint64_t)gridDim.x * (RT_BLOCK / 64);
    uint32_t bsz = 64u;
    if (n < 64u * n_waves_grid) { const uint64_t per = (n + n_waves_grid - 1) / n_waves_grid; bsz = per < 1 ? 1u : (uint32_t)per; }
    for (;;) {
        uint32_t base32 = 0;
        if (lane == 0) base32 = atomicAdd(fetch_ctr, bsz);
        const uint64_t base = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)base32);
        if (base >= n) break;
        const uint64_t q = base + lane;
        const bool valid = lane < bsz && q < n;
        V3 o = v3(0, 0, 0), d = v3(1, 1, 1);
        if (valid) { o = v3(in.ox[q], in.oy[q], in.oz[q]); d = v3(in.dx[q], in.dy[q], in.dz[q]); }
        const V3 inv = v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
        HitState rec; rec.t = 0.0; rec.tp = 0.0; rec.tn = 0.0; rec.prim = -1;
        bool hit = false;
        uint64_t cell = 0;
        uint32_t cx = 0, cy = 0, cz = 0; bool cok = false, implicit_box = false;
        bool walking = false;
        if (valid) {
            ++c_rays;
            cell = grid_cell_from_point_xyz(o, g, cx, cy, cz, cok);    // get_box_enter (grid.rs:119-129)
            walking = cell < g.n_cells;
            if (!walking) {
                double mn[3] = {g.origin.x, g.origin.y, g.origin.z}, mx[3] = {g.end.x, g.end.y, g.end.z};
                double oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
                double a = 0.001, b = INF; bool ok = true;
                for (int ax = 0; ax < 3; ++ax) {                        // AABB::enter_and_exit (aabb.rs:65-88)
                    double t0 = (mn[ax] - oo[ax]) * ii[ax], t1 = (mx[ax] - oo[ax]) * ii[ax];
                    if (t0 < t1) { if (t0 > a) a = t0; if (t1 < b) b = t1; }
                    else { if (t1 > a) a = t1; if (t0 < b) b = t0; }
                    if (b <= a) { ok = false; break; }
                }
                if (ok) { cell = grid_cell_from_point_xyz(o + d * a, g, cx, cy, cz, cok); walking = cell < g.n_cells; }
            }
        }
        uint32_t depth = 0;
        // state of the cell the lane stands in, kept while it is parked
        bool parked = false, any = false;
        double cell_b = 0.0, cmn0 = 0.0, cmn1 = 0.0, cmn2 = 0.0; int ex_axis = 0, ex_dir = 0;
        uint32_t pk_chunk = 0, pk_len = 0;
        double cp_t = 0.0, cp_tp = 0.0, cp_tn = 0.0, cp_cl = 0.0; int32_t cp_prim = -1;   // the copy of rec after the cell's list (grid.rs:160-162); cp_cl = closest_so_far of a parked ray
        // grid.rs:163-181 after the cell's objects were tested: take the copy, stop on a hit inside the cell, else step
        auto leave_cell = [&]() {
            if (any) {
                rec.t = cp_t; rec.tp = cp_tp; rec.tn = cp_tn; rec.prim = cp_prim;
                if (rec.t <= cell_b) { hit = true; walking = false; return; }
            }
            ++c_box;                                                   // grid.rs:172
            V3 step = v3(0, 0, 0);                                     // aabb.rs:161-165
            const double sv = (double)ex_dir * (ex_axis == 0 ? g.box_size.x : (ex_axis == 1 ? g.box_size.y : g.box_size.z));
            if (ex_axis == 0) step.x = sv; else if (ex_axis == 1) step.y = sv; else step.z = sv;
            const V3 next_xyz = v3(cmn0, cmn1, cmn2) + step;
            const bool inside = g.origin.x <= next_xyz.x && g.end.x >= next_xyz.x && g.origin.y <= next_xyz.y && g.end.y >= next_xyz.y
                             && g.origin.z <= next_xyz.z && g.end.z >= next_xyz.z;   // grid.rs:186-192
            if (!inside) { walking = false; return; }
            if (!(implicit_box && grid_step_regular(g, ex_axis, ex_dir, cx, cy, cz, cell)))
                cell = grid_index_xyz(next_xyz - g.origin, g, cx, cy, cz, cok);   // grid.rs:76-83
            if (cell >= g.n_cells) walking = false;
            ++depth;
        };
        for (;;) {
            // ---- phase A: every lane walks until it is done or stands in a cell with a chunked list
            while (ballot64(walking) != 0) {
                if (walking) {
                    if (depth > g.max_depth) { walking = false; }      // grid.rs:152
                    else {
                        ++c_steps;
                        double mn[3], mx[3]; bool empty;
                        implicit_box = grid_cell_box(g, cell, cx, cy, cz, cok, mn, mx, empty);
                        double oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
                        double a = -INF, b = INF; bool ok = true; ex_axis = 0; ex_dir = 0;
                        for (int ax = 0; ax < 3; ++ax) {                // AABB::hit with Interval::universe keeping the Exit (aabb.rs:28-62)
                            double t0 = (mn[ax] - oo[ax]) * ii[ax], t1 = (mx[ax] - oo[ax]) * ii[ax];
                            if (t0 < t1) { if (t0 > a) a = t0; if (t1 < b) { b = t1; ex_axis = ax; ex_dir = 1; } }
                            else { if (t1 > a) a = t1; if (t0 < b) { b = t0; ex_axis = ax; ex_dir = -1; } }
                            if (b < a) { ok = false; break; }
                        }
                        if (!ok) { walking = false; }                  // grid.rs:155-156
                        else {
                            ++c_gbox;
                            cell_b = b; cmn0 = mn[0]; cmn1 = mn[1]; cmn2 = mn[2];
                            uint64_t lo = 0, hi = 0; uint32_t ck = RT_NO_CHUNK;
                            if (!empty) { lo = g.cell_offset[cell]; hi = g.cell_offset[cell + 1]; ck = g.cell_chunk[cell]; }
                            any = false;
                            if (ck != RT_NO_CHUNK) {                   // long list: wait for the wave
                                parked = true; walking = false; pk_chunk = ck; pk_len = (uint32_t)(hi - lo);
                            } else {                                   // short list: GridBox::hit on a copy of rec (grid.rs:160-167, 281-294), lane by lane
                                HitState copy = rec;
                                double closest = tmax0;
                                if (g.cell_tri) {
                                    for (uint64_t k = lo; k < hi; ++k) { ++c_prim; const DCellTri ct = g.cell_tri[k]; if (hit_triangle_cell(prims, ct, o, d, tmin0, closest, copy)) { any = true; closest = copy.t; } }
                                } else {
                                    for (uint64_t k = lo; k < hi; ++k) { ++c_prim; const uint32_t pi = g.prim_list[k]; if (hit_triangle_pre(prims, pre, pi, o, d, tmin0, closest, copy)) { any = true; closest = copy.t; } }
                                }
                                cp_t = copy.t; cp_tp = copy.tp; cp_tn = copy.tn; cp_prim = copy.prim;
                                leave_cell();
                            }
                        }
                    }
                }
            }
            // ---- phase B: the parked rays.  Rays that stand in the same cell form a group; each 64-entry chunk of the cell's list is
            // loaded once (lane l = entry l) and tested against every ray of the group in turn, the ray coming from its lane through
            // readlane (scalar registers) and its running GridBox::hit state (closest_so_far, the copy's t / p parameter / primitive)
            // living in that lane's own registers between chunks.
            unsigned long long pm = ballot64(parked);
            if (pm == 0) break;
            if (parked) {
                cp_t = rec.t; cp_tp = rec.tp; cp_prim = rec.prim; cp_cl = tmax0; any = false;
                s_ray[wv][0][lane] = make_double2(o.x, o.y); s_ray[wv][1][lane] = make_double2(o.z, d.x); s_ray[wv][2][lane] = make_double2(d.y, d.z);
            }
            wave_sync();
            while (pm != 0) {
                const uint32_t k = (uint32_t)__builtin_ctzll(pm);
                const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)pk_chunk, (int)k), len = (uint32_t)__builtin_amdgcn_readlane((int)pk_len, (int)k);
                const unsigned long long grp = ballot64(parked && pk_chunk == c0);      // a cell's first chunk index names the cell
                pm &= ~grp;
                const uint32_t nch = (len + 63u) >> 6;
                if ((grp >> lane) & 1ull) c_prim += len;
                for (uint32_t j = 0; j < nch; ++j) {
                    const DCellChunk* ch = g.chunks + (c0 + j);
                    const V3 nn = v3(ch->n[0][lane], ch->n[1][lane], ch->n[2][lane]);
                    const double dd = ch->dd[lane];
                    const V3 v0 = v3(ch->v[0][lane], ch->v[1][lane], ch->v[2][lane]), v1 = v3(ch->v[3][lane], ch->v[4][lane], ch->v[5][lane]);
                    const V3 v2 = v3(ch->v[6][lane], ch->v[7][lane], ch->v[8][lane]);
                    const uint32_t pi = ch->pi[lane];
                    const V3 v0v1 = v1 - v0, v1v2 = v2 - v1, v2v0 = v0 - v2;               // triangle.rs:61,66,71: ray-independent, once per chunk
                    for (unsigned long long gm = grp; gm != 0; gm &= gm - 1ull) {
                        const uint32_t r = (uint32_t)__builtin_ctzll(gm);
                        // the ray comes out of LDS (same address for every lane: a broadcast read on the LDS pipe) rather than through
                        // twelve v_readlane on the VALU; closest_so_far changes between chunks and stays a readlane
                        const double2 r0 = s_ray[wv][0][r], r1 = s_ray[wv][1][r], r2 = s_ray[wv][2][r];
                        const V3 ko = v3(r0.x, r0.y, r1.x), kd = v3(r1.y, r2.x, r2.y);
                        double closest = rl_f64(cp_cl, r);
                        const double nd = dot(nn, kd);                      // triangle.rs:43
                        const bool a = tmin0 < nd;                          // the half of :44-47 that does not read closest_so_far
                        const double t = -(dot(nn, ko) + dd) / nd;          // :50-52
                        const bool tpos = !(t < 0.0);                       // :55-57
                        bool E = false;
                        if (ballot64(a && tpos) != 0) {                     // :59-75, for every lane (a wave runs them for one lane as for all)
                            const V3 p = ko + (t * kd);
                            const V3 v0p = p - v0, v1p = p - v1, v2p = p - v2;
                            const bool e0 = !(dot(nn, cross(v0v1, v0p)) <= 0.0);
                            const bool e1 = !(dot(nn, cross(v1v2, v1p)) <= 0.0);
                            const bool e2 = !(dot(nn, cross(v2v0, v2p)) <= 0.0);
                            E = e0 && e1 && e2;
                        }
                        const unsigned long long Mtp = ballot64(tpos), Mok = ballot64(tpos && E);
                        double u_t = 0.0, u_tp = 0.0; int32_t u_prim = -1; bool u_acc = false, u_have_t = false, u_have_tp = false;
                        unsigned long long rem = ~0ull;
                        while (rem != 0) {                                  // the acceptance chain of this chunk, in list order
                            const unsigned long long C1 = ballot64(a && nd < closest) & rem;
                            const unsigned long long A = C1 & Mok;
                            unsigned long long fin = C1;
                            uint32_t k2 = 0;
                            if (A != 0) { k2 = (uint32_t)__builtin_ctzll(A); const unsigned long long upto = (k2 == 63u) ? ~0ull : ((2ull << k2) - 1ull); fin = C1 & upto; rem &= ~upto; }
                            else rem = 0;
                            if (fin != 0) {
                                u_t = rl_f64(t, 63u - (uint32_t)__builtin_clzll(fin)); u_have_t = true;
                                const unsigned long long F2 = fin & Mtp;
                                if (F2 != 0) { u_tp = rl_f64(t, 63u - (uint32_t)__builtin_clzll(F2)); u_have_tp = true; }
                            }
                            if (A != 0) { closest = rl_f64(t, k2); u_prim = __builtin_amdgcn_readlane((int)pi, (int)k2); u_acc = true; }
                        }
                        if (lane == r) {
                            if (u_have_t) cp_t = u_t;
                            if (u_have_tp) cp_tp = u_tp;
                            if (u_acc) { cp_cl = closest; cp_prim = u_prim; any = true; }
                        }
                    }
                }
            }
            if (parked) cp_tn = any ? cp_cl : rec.tn;
            if (parked) { parked = false; walking = true; leave_cell(); }
        }
        if (valid) store_hit(out, q, hit, rec, prims, false);
    }
    if (stats) {
        stat_add(stats, ST_BOX, c_box); stat_add(stats, ST_PRIM, c_prim); stat_add(stats, ST_STEPS, c_steps);
        stat_add(stats, ST_GRIDBOX, c_gbox); stat_add(stats, ST_SEGMENTS, c_rays);
    }
}

#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 6      // 80 VGPRs: three 512-thread blocks per CU instead of two (shade 59 -> 56.5 ms; 8 spills and is slower)
#endif
// ---------------------------------------------------------------------------------------------
// shade: the body of ray_color after world.hit (camera.rs:161-185) for one bounce, iteratively:
//   miss            -> radiance = throughput * sky (camera.rs:182-184), path ends
//   hit, no scatter -> radiance = 0 (camera.rs:176), path ends
//   hit, scatter    -> scatter_rays++ (camera.rs:171); throughput *= attenuation; if this was the last
//                      allowed bounce the child call returns black (camera.rs:155-157), else the scattered
//                      ray is appended to the next queue (wave ballots + a prefix sum over the workgroup, one atomic per 512 threads).
// The hit record is rebuilt from (prim, tp): rec.p = at(tp) (sphere.rs:50, triangle.rs:59), the outward
// normal and front_face as sphere.rs:51-52 / triangle.rs:39-41,77 compute them.
// The throughput is a left-to-right product where the reference nests right-to-left (camera.rs:172):
// radiance may differ from the reference's in the last ulps; hit/miss decisions are unaffected.
#ifndef RT_SHADE_WAVES_PF
#define RT_SHADE_WAVES_PF 4   // the prefetching variant holds a second queue entry (26 VGPRs): 128 VGPRs, two 512-thread blocks per CU
#endif
#define RT_SHADE_LDS_MATS 512      // materials kept in LDS (20 KB): the third dependent global load of a shade (ray -> primitive -> material) becomes an LDS read
// MODE bit 0: the next tile's queue entry (ray, throughput, id, hit) is loaded while the current one is shaded ("two rays per
//             thread" as far as loads in flight go: the experiment DESIGN carried since round 1);
// MODE bit 1: all-sphere scenes read the 48-byte DSphere (3 lane accesses) instead of the 80-byte DPrim (5).
// Both measured in round 3 (profiles/round3_shade_variants.md).
struct ShadeIn { V3 o, d, thr; uint32_t id; int32_t pi; double tp, tn; };
template <int MODE>
__global__ __launch_bounds__(RT_SBLOCK, (MODE & 1) ? RT_SHADE_WAVES_PF : RT_SHADE_WAVES) void k_shade(const DPrim* __restrict__ prims, const DSphere* __restrict__ spheres, const DPrim32* __restrict__ tris32, double tri_scale, const DMat* __restrict__ mats, uint32_t n_mats,
                                                    RayQ in, const uint32_t* __restrict__ count_in,
                                                    const int32_t* __restrict__ hit_prim_, const double* __restrict__ hit_tp, const double* __restrict__ hit_tn,
                                                    RayQ out, uint32_t* count_out,
                                                    double* __restrict__ rad, uint32_t WH, uint32_t s_base, uint32_t s_count,
                                                    uint32_t bounce, uint32_t max_depth, uint64_t seed,
                                                    unsigned long long* stats) {
    __shared__ uint32_t s_tmp[RT_SBLOCK / 64];
    __shared__ CoopLds s_coop[RT_SBLOCK / 64];
    __shared__ DMat s_mats[RT_SHADE_LDS_MATS];
    const bool mats_lds = n_mats <= RT_SHADE_LDS_MATS;
    if (mats_lds) { for (uint32_t i = threadIdx.x; i < n_mats; i += RT_SBLOCK) s_mats[i] = mats[i]; __syncthreads(); }
    const uint64_t n = (uint64_t)*count_in;
    const uint64_t T = (uint64_t)gridDim.x * RT_SBLOCK;
    uint64_t c_scatter = 0;
    auto fetch = [&](uint64_t q, ShadeIn& r) {
        r.o = v3(in.ox[q], in.oy[q], in.oz[q]); r.d = v3(in.dx[q], in.dy[q], in.dz[q]);
        r.thr = v3(in.tr[q], in.tg[q], in.tb[q]); r.id = in.id[q]; r.pi = hit_prim_[q];
        if (MODE & 1) { r.tp = hit_tp[q]; r.tn = hit_tn ? hit_tn[q] : 0.0; }      // ahead of time: not yet known whether the ray hit
        else if (r.pi >= 0) { r.tp = hit_tp[q]; r.tn = hit_tn ? hit_tn[q] : 0.0; }
    };
    ShadeIn nx; nx.o = nx.d = nx.thr = v3(0, 0, 0); nx.id = 0; nx.pi = -1; nx.tp = nx.tn = 0.0;
    if ((MODE & 1) && (uint64_t)blockIdx.x * RT_SBLOCK + threadIdx.x < n) fetch((uint64_t)blockIdx.x * RT_SBLOCK + threadIdx.x, nx);
    for (uint64_t base = (uint64_t)blockIdx.x * RT_SBLOCK; base < n; base += T) {      // uniform per block (barriers inside)
        const uint64_t q = base + threadIdx.x;
        const bool valid = q < n;
        bool alive = false;
        V3 no = v3(0, 0, 0), nd = v3(0, 0, 0), thr = v3(0, 0, 0);
        uint32_t id = 0;
        // part 1: the hit record (or the path's end); which lanes will call Vec3::random_unit()
        V3 d = v3(0, 0, 0), p = v3(0, 0, 0), normal = v3(0, 0, 0), radiance = v3(0, 0, 0);
        bool surface = false, front_face = false;
        DMat m; m.albedo[0] = m.albedo[1] = m.albedo[2] = 0.0; m.param = 0.0; m.type = 2u; m.pad = 0u;
        ShadeIn cur = nx;
        if (MODE & 1) { if (q + T < n) fetch(q + T, nx); }           // in flight while this tile is shaded
        else if (valid) fetch(q, cur);
        if (valid) {
            const V3 o = cur.o;
            d = cur.d; thr = cur.thr; id = cur.id;
            const int32_t pi = cur.pi;
            if (pi == RT_HIT_EDGE) {                                  // camera.rs:162-164,178-180: Color::red()
                radiance = thr * v3(1.0, 0.0, 0.0);
            } else if (pi < 0) {                                      // camera.rs:182-184
                radiance = thr * sky_color(d);
            } else {
                const double tp = cur.tp;
                p = o + (tp * d);
                V3 outward; uint32_t tm;
                if ((MODE & 2) && spheres) {                          // all-sphere scene: sphere.rs:50-52 from the 48-byte record
                    const double2* sp2 = (const double2*)(spheres + pi);
                    const double2 c0 = sp2[0], c1 = sp2[1]; const uint4 w2 = *(const uint4*)(sp2 + 2);
                    tm = w2.x;
                    outward = (p - v3(c0.x, c0.y, c1.x)) / c1.y;
                } else if ((MODE & 2) && tris32) {                     // all-triangle lattice scene: the 48 bytes of the 64-byte-stride record (one L1 line, three accesses)
                    const uint4* tp4 = (const uint4*)(tris32 + pi);
                    const uint4 q0 = tp4[0], q1 = tp4[1], q2 = tp4[2];
                    tm = q2.y;
                    const V3 v0 = v3(lattice(q0.x, tri_scale), lattice(q0.y, tri_scale), lattice(q0.z, tri_scale));
                    const V3 v1 = v3(lattice(q0.w, tri_scale), lattice(q1.x, tri_scale), lattice(q1.y, tri_scale));
                    const V3 v2 = v3(lattice(q1.z, tri_scale), lattice(q1.w, tri_scale), lattice(q2.x, tri_scale));
                    outward = cross(v1 - v0, v2 - v0);
                } else {
                    DPrim P = prims[pi];
                    tm = P.tm;
                    if (P.tm >> 31) {
                        V3 v0 = v3(P.v[0], P.v[1], P.v[2]), v1 = v3(P.v[3], P.v[4], P.v[5]), v2 = v3(P.v[6], P.v[7], P.v[8]);
                        outward = cross(v1 - v0, v2 - v0);
                    } else {
                        V3 pn = hit_tn ? o + (cur.tn * d) : p;            // rec.p as it was when the sphere was accepted
                        outward = (pn - v3(P.v[0], P.v[1], P.v[2])) / P.v[3];
                    }
                }
                front_face = dot(d, outward) < 0.0;                   // hittable.rs:26-29
                normal = front_face ? outward : -outward;
                m = mats_lds ? s_mats[tm & 0x7fffffffu] : mats[tm & 0x7fffffffu];
                surface = true;
            }
        }
        // Lambertian and Metal both start with Vec3::random_unit() (material.rs:36,69); its rejection loop is shared out
        // over the wave (coop_reject): the lanes of both materials, and every other lane as a helper
        const uint32_t path_lo = id % WH, path_hi = s_base + id / WH;
        const bool wants_ru = surface && m.type != 2;
        V3 ru = coop_reject<3>(wants_ru, (uint32_t)seed, (uint32_t)(seed >> 32), path_lo, path_hi, 1u + bounce, 0u, &s_coop[threadIdx.x >> 6]);
        if (valid) {
            if (surface) {
                if (wants_ru) ru = ru / sqrt(length_squared(ru));      // vec3.rs:38
                V3 attenuation; bool scattered;
                if (m.type == 0) {                                    // Lambertian material.rs:35-46
                    V3 sd = normal + ru;
                    if (near_zero(sd)) sd = normal;
                    nd = sd; attenuation = v3(m.albedo[0], m.albedo[1], m.albedo[2]); scattered = true;
                } else if (m.type == 1) {                             // Metal material.rs:67-74
                    V3 reflected = reflect(d, normal);
                    reflected = unit(reflected) + (m.param * ru);
                    nd = reflected; attenuation = v3(m.albedo[0], m.albedo[1], m.albedo[2]);
                    scattered = dot(nd, normal) > 0.0;
                } else {                                              // Dielectric material.rs:94-113
                    Rng g(seed, path_lo, path_hi, 1u + bounce);
                    attenuation = v3(1.0, 1.0, 1.0);
                    double ri = front_face ? 1.0 / m.param : m.param;
                    V3 unit_direction = unit(d);
                    double cos_theta = fmin(dot(-unit_direction, normal), 1.0);
                    double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
                    if (ri * sin_theta > 1.0 || reflectance(cos_theta, ri) > g.next()) nd = reflect(unit_direction, normal);
                    else nd = refract(unit_direction, normal, ri);
                    scattered = true;
                }
                if (scattered) {
                    ++c_scatter;
                    thr = thr * attenuation;
                    no = p;
                    alive = (bounce + 1u < max_depth);                // else ray_color(depth == 0) returns black
                }
            }
            if (!alive) {                                             // every path writes its slot exactly once
                double* slot = rad + ((uint64_t)path_lo * s_count + id / WH) * 3u;   // one 24-byte record: a dead path's write touches one line, not three
                slot[0] = radiance.x; slot[1] = radiance.y; slot[2] = radiance.z;
            }
        }
        uint32_t w = block_compact_slot(alive, count_out, s_tmp);
        if (alive) {
            out.ox[w] = no.x; out.oy[w] = no.y; out.oz[w] = no.z;
            out.dx[w] = nd.x; out.dy[w] = nd.y; out.dz[w] = nd.z;
            out.tr[w] = thr.x; out.tg[w] = thr.y; out.tb[w] = thr.z;
            out.id[w] = id;
        }
    }
    if (stats) stat_add(stats, ST_SCATTER, c_scatter);
}
#define RT_INST_SHADE(M) template __global__ void k_shade<M>(const DPrim*, const DSphere*, const DPrim32*, double, const DMat*, uint32_t, RayQ, const uint32_t*, const int32_t*, const double*, const double*, RayQ, uint32_t*, double*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint64_t, unsigned long long*);
RT_INST_SHADE(0) RT_INST_SHADE(1) RT_INST_SHADE(2) RT_INST_SHADE(3)

// resolve: `pixel_color += ray_color(...)` (camera.rs:72) for the batch's samples, in sample order.  One lane per pixel
// walks that pixel's contiguous run of records (rad[pixel][sample][rgb]).
__global__ __launch_bounds__(RT_SBLOCK) void k_accumulate(const double* __restrict__ rad, uint32_t WH, uint32_t s_count,
                                                          double* __restrict__ fb, int first_batch) {
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < WH; p += (uint64_t)gridDim.x * blockDim.x) {
        double r = first_batch ? 0.0 : fb[p * 3], g = first_batch ? 0.0 : fb[p * 3 + 1], b = first_batch ? 0.0 : fb[p * 3 + 2];
        const double* run = rad + p * s_count * 3u;
#pragma unroll 4
        for (uint32_t s = 0; s < s_count; ++s) { r = r + run[3 * s]; g = g + run[3 * s + 1]; b = b + run[3 * s + 2]; }
        fb[p * 3] = r; fb[p * 3 + 1] = g; fb[p * 3 + 2] = b;
    }
}

// interleaved xyz -> SoA queue (rt_trace input)
__global__ void k_load_rays(const double* __restrict__ orig, const double* __restrict__ dir, uint64_t n, RayQ q) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    q.ox[k] = orig[3 * k]; q.oy[k] = orig[3 * k + 1]; q.oz[k] = orig[3 * k + 2];
    q.dx[k] = dir[3 * k]; q.dy[k] = dir[3 * k + 1]; q.dz[k] = dir[3 * k + 2];
}
