// rt_build.hip — Bvh::new (src/acceleration/bvh.rs:19-180) on the GPU for all three build modes (`BvhSahPosition`, the default,
// described below; `BvhSahPlane` = the same with the median as the only candidate of an axis, bvh.rs:114-123; `BvhNaive` = median
// split of the x-sorted slice for nodes of >= 3 objects, no cost, bvh.rs:87-99), producing the
// same tree, node numbering and object order as the reference (and as the host builder in host/accel.cpp, which the tests
// compare it with byte for byte).
//
// The reference splits one node at a time: stable-sort the node's slice by centroid on each axis, evaluate every split
// position with SA(left)·|left| + SA(right)·|right| (bvh.rs:148-160), keep the first strict minimum in (axis, split) order
// (bvh.rs:124-133), re-sort the slice by the winning axis (axis 0 when nothing wins, bvh.rs:136-140), recurse left then right.
// Here all nodes of one depth are split together:
//   * the node slices ("segments") of a level are contiguous ranges of one dense array of objects.  A stable radix sort of the
//     whole array by centroid key followed by a stable sort by segment number sorts every segment by that axis, ties keeping
//     their current order — exactly what `slice::sort_by` does per node;
//   * prefix and suffix boxes are segmented scans (f64 min/max are exact and, written as BoxUnion / BoxUnionRev below, resolve
//     equal planes — signed zeros — to the object the reference's left fold would keep); the cost of every
//     split position is then one expression per element, and the first minimum per segment is a segmented min-scan on
//     (cost, position);
//   * a per-segment kernel compares the three axes in order with the reference's strict `<`, writes the children and the next
//     level's segments.
// Node ids here are in creation (level) order; the host renumbers them into the reference's order (children are numbered when
// their parent splits, left subtree first: bvh.rs:167-175) in one pass at the end.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/rt_hip.h"

namespace rtb {

struct Box6 { double mn[3], mx[3]; };
// f64::min / f64::max as the host builder and the oracle evaluate them (std::fmin / std::fmax on x86-64): of two equal operands
// — only +0 and -0 are equal and distinguishable — the SECOND is returned, and a NaN operand yields the other one.  The device's
// own fmin (v_min_f64) orders -0 below +0 instead, so a box plane that is a zero could come out with the other sign.
__host__ __device__ inline double min_tie2(double a, double b) { return (a < b || b != b) ? a : b; }
__host__ __device__ inline double max_tie2(double a, double b) { return (a > b || b != b) ? a : b; }
// objects_to_aabb (hittable_list.rs:123-133) is a left fold, acc = acc + next: `a` is the fold so far, `b` the next object.  With
// min_tie2 the fold returns the LAST of equal minima, which is associative, so a parallel scan reproduces it bit for bit.
struct BoxUnion {
    __host__ __device__ Box6 operator()(const Box6& a, const Box6& b) const {            // aabb.rs:134-147
        Box6 r;
        for (int c = 0; c < 3; ++c) { r.mn[c] = min_tie2(a.mn[c], b.mn[c]); r.mx[c] = max_tie2(a.mx[c], b.mx[c]); }
        return r;
    }
};
// The suffix boxes are a prefix scan over the REVERSED array: `a` holds objects that stand later in the slice than `b`.  The
// reference folds the right child's slice left to right (bvh.rs:151), so equal planes must again resolve to the later object: `a`.
struct BoxUnionRev {
    __host__ __device__ Box6 operator()(const Box6& a, const Box6& b) const {
        Box6 r;
        for (int c = 0; c < 3; ++c) { r.mn[c] = min_tie2(b.mn[c], a.mn[c]); r.mx[c] = max_tie2(b.mx[c], a.mx[c]); }
        return r;
    }
};
struct Cand { unsigned long long key; uint32_t pos; uint32_t pad; };                     // ordered cost, dense position
struct CandMin {                                                                         // first minimum: ties keep the lower position
    __host__ __device__ Cand operator()(const Cand& a, const Cand& b) const { return (b.key < a.key || (b.key == a.key && b.pos < a.pos)) ? b : a; }
};
struct SegEq { __host__ __device__ bool operator()(uint32_t a, uint32_t b) const { return a == b; } };

__host__ __device__ inline double area(const Box6& b) {                                  // aabb.rs:104-110
    return 2.0 * ((b.mx[0] - b.mn[0]) * (b.mx[1] - b.mn[1])
                  + (b.mx[0] - b.mn[0]) * (b.mx[2] - b.mn[2])
                  + (b.mx[1] - b.mn[1]) * (b.mx[2] - b.mn[2]));
}
// f64::total_cmp order as an unsigned integer (the radix sort's order)
__host__ __device__ inline unsigned long long total_key(double d) {
    unsigned long long x; memcpy(&x, &d, 8);
    return (x >> 63) ? ~x : (x | 0x8000000000000000ull);
}
// a cost as a key whose unsigned order is the order `<` gives on non-negative doubles; NaN (never accepted by `h < best`) last
__device__ inline unsigned long long cost_key(double h) {
    if (!(h == h)) return ~0ull;
    unsigned long long x; memcpy(&x, &h, 8);
    return (x >> 63) ? 0ull : x;            // -0.0 and (impossible) negative costs order first, like 0
}
#define RTB_NOCAND (~0ull)

// ---- per-level kernels ------------------------------------------------------------------------
// dense position -> segment (binary search in the segment bases) and the object standing there
__global__ void k_dense(const uint32_t* __restrict__ seg_base, const uint32_t* __restrict__ seg_first, uint32_t S, uint32_t M,
                        const uint32_t* __restrict__ order, uint32_t* __restrict__ segidx, uint32_t* __restrict__ obj) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    uint32_t lo = 0, hi = S;                 // last s with seg_base[s] <= i
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (seg_base[mid] <= i) lo = mid; else hi = mid; }
    segidx[i] = lo;
    obj[i] = order[seg_first[lo] + (i - seg_base[lo])];
}
__global__ void k_keys(const unsigned long long* __restrict__ axis_key, const uint32_t* __restrict__ obj, const uint32_t* __restrict__ segidx, uint32_t M,
                       unsigned long long* __restrict__ key, unsigned long long* __restrict__ val) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    key[i] = axis_key[obj[i]];
    val[i] = ((unsigned long long)segidx[i] << 32) | obj[i];
}
__global__ void k_unpack(const unsigned long long* __restrict__ val, uint32_t M, uint32_t* __restrict__ seg, uint32_t* __restrict__ obj) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    seg[i] = (uint32_t)(val[i] >> 32); obj[i] = (uint32_t)val[i];
}
// boxes of the sorted objects, forward and reversed (the suffix scan runs as a prefix scan on the reversed array)
__global__ void k_gather_boxes(const Box6* __restrict__ boxes, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ segidx, uint32_t M,
                               Box6* __restrict__ fwd, Box6* __restrict__ rev, uint32_t* __restrict__ rev_seg) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const Box6 b = boxes[sorted[i]];
    fwd[i] = b; rev[M - 1 - i] = b; rev_seg[M - 1 - i] = segidx[i];
}
// cost of splitting segment s before dense position i (bvh.rs:152-154); position `base` of a segment is no candidate
// (median_only: BvhSahPlane evaluates the one position len/2 per axis, bvh.rs:114-123)
__global__ void k_cost(const Box6* __restrict__ pre, const Box6* __restrict__ suf_rev, const uint32_t* __restrict__ segidx,
                       const uint32_t* __restrict__ seg_base, const uint32_t* __restrict__ seg_count, uint32_t M, bool median_only, Cand* __restrict__ cand) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const uint32_t s = segidx[i], j = i - seg_base[s], n = seg_count[s];
    Cand c; c.pos = i; c.pad = 0; c.key = RTB_NOCAND;
    if (j >= 1 && (!median_only || j == n / 2)) {
        const double h = area(pre[i - 1]) * (double)j + area(suf_rev[M - 1 - i]) * (double)(n - j);
        c.key = cost_key(h);
    }
    cand[i] = c;
}

struct NodeArrays {           // creation-order node table (capacity 2N)
    Box6* box; uint32_t* first; uint32_t* count; uint32_t* left;      // left = id of the left child (right = left + 1), 0 = leaf
};
// one thread per segment: the winner over the three axes in the reference's order, strict `<` (bvh.rs:101,127)
__global__ void k_decide(uint32_t S, const uint32_t* __restrict__ seg_node, const uint32_t* __restrict__ seg_base, const uint32_t* __restrict__ seg_count,
                         const Cand* __restrict__ best0, const Cand* __restrict__ best1, const Cand* __restrict__ best2,
                         NodeArrays nodes, bool naive, uint32_t* __restrict__ best_axis, uint32_t* __restrict__ split, uint32_t* __restrict__ is_split) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const uint32_t base = seg_base[s], n = seg_count[s], last = base + n - 1;
    if (naive) { best_axis[s] = 0; split[s] = n / 2; is_split[s] = 1u; return; }        // bvh.rs:87-99: every node of >= 3 objects (the only active ones) splits at its median
    const Box6 nb = nodes.box[seg_node[s]];
    double best = area(nb) * (double)n;                                   // bvh.rs:101
    uint32_t ax = 0, sp = 0;
    const Cand c[3] = {best0[last], best1[last], best2[last]};            // inclusive min-scan: the segment's minimum stands at its end
    for (int a = 0; a < 3; ++a) {
        if (c[a].key == RTB_NOCAND) continue;
        double h; unsigned long long k = c[a].key; memcpy(&h, &k, 8);
        if (h < best) { best = h; ax = (uint32_t)a; sp = c[a].pos - base; }
    }
    best_axis[s] = ax; split[s] = sp; is_split[s] = sp ? 1u : 0u;
}
// the node's slice takes the order of the winning axis, split or not (bvh.rs:136-140)
__global__ void k_apply_order(uint32_t M, const uint32_t* __restrict__ segidx, const uint32_t* __restrict__ seg_base, const uint32_t* __restrict__ seg_first,
                              const uint32_t* __restrict__ best_axis, const uint32_t* __restrict__ s0, const uint32_t* __restrict__ s1, const uint32_t* __restrict__ s2,
                              uint32_t* __restrict__ order) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const uint32_t s = segidx[i], a = best_axis[s];
    order[seg_first[s] + (i - seg_base[s])] = a == 0 ? s0[i] : (a == 1 ? s1[i] : s2[i]);
}
// children of the segments that split (BvhNode::sub_divide bvh.rs:162-180): node records, and the flags / sizes of the next
// level's segments (children with fewer than two objects can never split: get_split has no candidate for them; BvhNaive
// leaves nodes of fewer than three alone, slice order included: bvh.rs:88-90)
__global__ void k_children(uint32_t S, const uint32_t* __restrict__ seg_node, const uint32_t* __restrict__ seg_base, const uint32_t* __restrict__ seg_first,
                           const uint32_t* __restrict__ seg_count, const uint32_t* __restrict__ best_axis, const uint32_t* __restrict__ split,
                           const uint32_t* __restrict__ split_rank, uint32_t node_base, uint32_t M,
                           const Box6* __restrict__ pre0, const Box6* __restrict__ pre1, const Box6* __restrict__ pre2,
                           const Box6* __restrict__ suf0, const Box6* __restrict__ suf1, const Box6* __restrict__ suf2,
                           NodeArrays nodes, uint32_t min_active, uint32_t* __restrict__ child_active, uint32_t* __restrict__ child_count) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const uint32_t sp = split[s];
    if (!sp) return;
    const uint32_t j = split_rank[s], l = node_base + 2 * j, r = l + 1, base = seg_base[s], n = seg_count[s], a = best_axis[s];
    const Box6* pre = a == 0 ? pre0 : (a == 1 ? pre1 : pre2);
    const Box6* suf = a == 0 ? suf0 : (a == 1 ? suf1 : suf2);
    nodes.left[seg_node[s]] = l;
    nodes.first[l] = seg_first[s]; nodes.count[l] = sp; nodes.box[l] = pre[base + sp - 1]; nodes.left[l] = 0;
    nodes.first[r] = seg_first[s] + sp; nodes.count[r] = n - sp; nodes.box[r] = suf[M - 1 - (base + sp)]; nodes.left[r] = 0;
    child_active[2 * j] = sp >= min_active ? 1u : 0u; child_count[2 * j] = sp >= min_active ? sp : 0u;
    child_active[2 * j + 1] = (n - sp) >= min_active ? 1u : 0u; child_count[2 * j + 1] = (n - sp) >= min_active ? (n - sp) : 0u;
}
// compact the active children into the next level's segment table
__global__ void k_next_segments(uint32_t n_children, uint32_t node_base, const uint32_t* __restrict__ child_active, const uint32_t* __restrict__ child_slot,
                                const uint32_t* __restrict__ child_dense, NodeArrays nodes,
                                uint32_t* __restrict__ seg_node, uint32_t* __restrict__ seg_first, uint32_t* __restrict__ seg_count, uint32_t* __restrict__ seg_base) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_children || !child_active[c]) return;
    const uint32_t t = child_slot[c], id = node_base + c;
    seg_node[t] = id; seg_first[t] = nodes.first[id]; seg_count[t] = nodes.count[id]; seg_base[t] = child_dense[c];
}
__global__ void k_prim_setup(const rt_prim* __restrict__ prims, const double* __restrict__ centroids, uint32_t n,
                             Box6* __restrict__ boxes, unsigned long long* __restrict__ k0, unsigned long long* __restrict__ k1, unsigned long long* __restrict__ k2,
                             uint32_t* __restrict__ order) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rt_prim p = prims[i];
    Box6 b;
    if (p.type == RT_SPHERE) {                           // sphere.rs:58-67
        for (int c = 0; c < 3; ++c) { b.mn[c] = p.v[c] - p.v[3]; b.mx[c] = p.v[c] + p.v[3]; }
    } else {                                             // triangle.rs:83-92
        for (int c = 0; c < 3; ++c) {
            b.mn[c] = min_tie2(min_tie2(p.v[c], p.v[3 + c]), p.v[6 + c]);
            b.mx[c] = max_tie2(max_tie2(p.v[c], p.v[3 + c]), p.v[6 + c]);
        }
    }
    boxes[i] = b;
    k0[i] = total_key(centroids[3 * (size_t)i]); k1[i] = total_key(centroids[3 * (size_t)i + 1]); k2[i] = total_key(centroids[3 * (size_t)i + 2]);
    order[i] = i;
}

// ---- host driver ---------------------------------------------------------------------------
struct DevBuf {
    std::vector<void*> all;
    template <class T> T* get(size_t n) { void* p = nullptr; if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr; all.push_back(p); return (T*)p; }
    ~DevBuf() { for (void* p : all) (void)hipFree(p); }
};

#define RTB_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + " failed: " + hipGetErrorString(e_); return RT_E_HIP; } } while (0)
#define RTB_ALLOC(ptr, T, n) do { ptr = buf.get<T>(n); if (!ptr) { err = "out of device memory in the BVH build"; return RT_E_HIP; } } while (0)

int build(const rt_prim* prims, const double* centroids, uint32_t n, int mode, rt_bvh_node* nodes_out, uint32_t* n_nodes_out,
          uint32_t* order_out, uint32_t* depth_out, hipStream_t st, std::string& err) {
    if (n == 0) { err = "scene has no objects"; return RT_E_ARG; }
    if (mode < RT_BVH_SAH_POSITION || mode > RT_BVH_NAIVE) { err = "unknown BVH build mode"; return RT_E_ARG; }
    const bool naive = mode == RT_BVH_NAIVE, median_only = mode == RT_BVH_SAH_PLANE;
    const uint32_t min_active = naive ? 3u : 2u;         // the smallest node get_split can split
    const int n_axes = naive ? 1 : 3;
    if (n > 0x3fffffffu) { err = "too many objects"; return RT_E_LIMIT; }
    DevBuf buf;
    const uint32_t TB = 256;
    auto blocks = [&](uint32_t m) { return dim3((m + TB - 1) / TB); };

    rt_prim* d_prims; double* d_cent; Box6* boxes; unsigned long long* akey[3]; uint32_t* order;
    RTB_ALLOC(d_prims, rt_prim, n); RTB_ALLOC(d_cent, double, 3 * (size_t)n); RTB_ALLOC(boxes, Box6, n);
    for (int a = 0; a < 3; ++a) RTB_ALLOC(akey[a], unsigned long long, n);
    RTB_ALLOC(order, uint32_t, n);
    RTB_CHECK(hipMemcpyAsync(d_prims, prims, (size_t)n * sizeof(rt_prim), hipMemcpyHostToDevice, st));
    RTB_CHECK(hipMemcpyAsync(d_cent, centroids, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_prim_setup, blocks(n), dim3(TB), 0, st, d_prims, d_cent, n, boxes, akey[0], akey[1], akey[2], order);

    NodeArrays nodes;
    const size_t cap = 2 * (size_t)n;
    RTB_ALLOC(nodes.box, Box6, cap); RTB_ALLOC(nodes.first, uint32_t, cap); RTB_ALLOC(nodes.count, uint32_t, cap); RTB_ALLOC(nodes.left, uint32_t, cap);

    // level state (segments) and per-level scratch, all sized for the worst case
    uint32_t *seg_node[2], *seg_first[2], *seg_count[2], *seg_base[2];
    for (int i = 0; i < 2; ++i) { RTB_ALLOC(seg_node[i], uint32_t, n); RTB_ALLOC(seg_first[i], uint32_t, n); RTB_ALLOC(seg_count[i], uint32_t, n); RTB_ALLOC(seg_base[i], uint32_t, n); }
    uint32_t *segidx, *obj, *seg_tmp, *obj_tmp, *sorted[3], *rev_seg, *best_axis, *split, *is_split, *split_rank, *child_active, *child_count, *child_slot, *child_dense;
    RTB_ALLOC(segidx, uint32_t, n); RTB_ALLOC(obj, uint32_t, n); RTB_ALLOC(seg_tmp, uint32_t, n); RTB_ALLOC(obj_tmp, uint32_t, n); RTB_ALLOC(rev_seg, uint32_t, n);
    for (int a = 0; a < 3; ++a) RTB_ALLOC(sorted[a], uint32_t, n);
    RTB_ALLOC(best_axis, uint32_t, n); RTB_ALLOC(split, uint32_t, n); RTB_ALLOC(is_split, uint32_t, n); RTB_ALLOC(split_rank, uint32_t, n);
    RTB_ALLOC(child_active, uint32_t, n + 2); RTB_ALLOC(child_count, uint32_t, n + 2); RTB_ALLOC(child_slot, uint32_t, n + 2); RTB_ALLOC(child_dense, uint32_t, n + 2);
    unsigned long long *key_a, *key_b, *val_a, *val_b;
    RTB_ALLOC(key_a, unsigned long long, n); RTB_ALLOC(key_b, unsigned long long, n); RTB_ALLOC(val_a, unsigned long long, n); RTB_ALLOC(val_b, unsigned long long, n);
    Box6 *fwd, *rev, *pre[3], *suf[3];
    RTB_ALLOC(fwd, Box6, n); RTB_ALLOC(rev, Box6, n);
    for (int a = 0; a < 3; ++a) { RTB_ALLOC(pre[a], Box6, n); RTB_ALLOC(suf[a], Box6, n); }
    Cand *cand, *best[3];
    RTB_ALLOC(cand, Cand, n); for (int a = 0; a < 3; ++a) RTB_ALLOC(best[a], Cand, n);
    uint32_t* d_scalars; RTB_ALLOC(d_scalars, uint32_t, 8);

    // hipCUB temporary storage: the largest request of the calls below at full size
    size_t tmp_bytes = 0, need = 0;
    RTB_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, key_a, key_b, val_a, val_b, (int)n, 0, 64, st)); tmp_bytes = std::max(tmp_bytes, need);
    RTB_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, seg_tmp, segidx, obj_tmp, obj, (int)n, 0, 32, st)); tmp_bytes = std::max(tmp_bytes, need);
    RTB_CHECK(hipcub::DeviceScan::InclusiveScanByKey(nullptr, need, segidx, fwd, pre[0], BoxUnion(), (int)n, SegEq(), st)); tmp_bytes = std::max(tmp_bytes, need);
    RTB_CHECK(hipcub::DeviceScan::InclusiveScanByKey(nullptr, need, segidx, cand, best[0], CandMin(), (int)n, SegEq(), st)); tmp_bytes = std::max(tmp_bytes, need);
    RTB_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, is_split, split_rank, (int)n + 2, st)); tmp_bytes = std::max(tmp_bytes, need);
    void* tmp; RTB_ALLOC(tmp, unsigned char, tmp_bytes);

    // root (Bvh::new bvh.rs:22): a leaf over everything, box = fold of all object boxes
    {
        RTB_CHECK(hipMemsetAsync(segidx, 0, (size_t)n * sizeof(uint32_t), st));
        size_t tb = tmp_bytes;
        RTB_CHECK(hipcub::DeviceScan::InclusiveScanByKey(tmp, tb, segidx, boxes, pre[0], BoxUnion(), (int)n, SegEq(), st));
        RTB_CHECK(hipMemcpyAsync(nodes.box, pre[0] + (n - 1), sizeof(Box6), hipMemcpyDeviceToDevice, st));
        const uint32_t h[4] = {0u, n, 0u, 0u};   // first, count, left, seg_base
        RTB_CHECK(hipMemcpyAsync(nodes.first, &h[0], 4, hipMemcpyHostToDevice, st));
        RTB_CHECK(hipMemcpyAsync(nodes.count, &h[1], 4, hipMemcpyHostToDevice, st));
        RTB_CHECK(hipMemcpyAsync(nodes.left, &h[2], 4, hipMemcpyHostToDevice, st));
        RTB_CHECK(hipMemcpyAsync(seg_node[0], &h[2], 4, hipMemcpyHostToDevice, st));
        RTB_CHECK(hipMemcpyAsync(seg_first[0], &h[0], 4, hipMemcpyHostToDevice, st));
        RTB_CHECK(hipMemcpyAsync(seg_count[0], &h[1], 4, hipMemcpyHostToDevice, st));
        RTB_CHECK(hipMemcpyAsync(seg_base[0], &h[3], 4, hipMemcpyHostToDevice, st));
        RTB_CHECK(hipStreamSynchronize(st));
    }

    uint32_t S = n >= min_active ? 1u : 0u, M = n >= min_active ? n : 0u, n_nodes = 1;
    int cur = 0;
    for (uint32_t level = 0; S > 0; ++level) {
        if (level > 4096) { err = "BVH deeper than 4096 levels"; return RT_E_LIMIT; }
        const uint32_t *sn = seg_node[cur], *sf = seg_first[cur], *sc = seg_count[cur], *sb = seg_base[cur];
        hipLaunchKernelGGL(k_dense, blocks(M), dim3(TB), 0, st, sb, sf, S, M, order, segidx, obj);
        int seg_bits = 1; while ((1ull << seg_bits) < S) ++seg_bits;
        for (int a = 0; a < n_axes; ++a) {
            hipLaunchKernelGGL(k_keys, blocks(M), dim3(TB), 0, st, akey[a], obj, segidx, M, key_a, val_a);
            size_t tb = tmp_bytes;
            RTB_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, key_a, key_b, val_a, val_b, (int)M, 0, 64, st));
            if (S > 1) {
                hipLaunchKernelGGL(k_unpack, blocks(M), dim3(TB), 0, st, val_b, M, seg_tmp, obj_tmp);
                tb = tmp_bytes;
                RTB_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, seg_tmp, rev_seg /* scratch: sorted segment numbers == segidx */, obj_tmp, sorted[a], (int)M, 0, seg_bits, st));
            } else {
                hipLaunchKernelGGL(k_unpack, blocks(M), dim3(TB), 0, st, val_b, M, seg_tmp, sorted[a]);
            }
            hipLaunchKernelGGL(k_gather_boxes, blocks(M), dim3(TB), 0, st, boxes, sorted[a], segidx, M, fwd, rev, rev_seg);
            tb = tmp_bytes;
            RTB_CHECK(hipcub::DeviceScan::InclusiveScanByKey(tmp, tb, segidx, fwd, pre[a], BoxUnion(), (int)M, SegEq(), st));
            tb = tmp_bytes;
            RTB_CHECK(hipcub::DeviceScan::InclusiveScanByKey(tmp, tb, rev_seg, rev, suf[a], BoxUnionRev(), (int)M, SegEq(), st));
            if (naive) continue;                         // no cost: the median of the x order
            hipLaunchKernelGGL(k_cost, blocks(M), dim3(TB), 0, st, pre[a], suf[a], segidx, sb, sc, M, median_only, cand);
            tb = tmp_bytes;
            RTB_CHECK(hipcub::DeviceScan::InclusiveScanByKey(tmp, tb, segidx, cand, best[a], CandMin(), (int)M, SegEq(), st));
        }
        hipLaunchKernelGGL(k_decide, blocks(S), dim3(TB), 0, st, S, sn, sb, sc, best[0], best[1], best[2], nodes, naive, best_axis, split, is_split);
        hipLaunchKernelGGL(k_apply_order, blocks(M), dim3(TB), 0, st, M, segidx, sb, sf, best_axis, sorted[0], sorted[1], sorted[2], order);
        // ranks of the splitting segments (+ their number in the extra element)
        RTB_CHECK(hipMemsetAsync(is_split + S, 0, sizeof(uint32_t), st));
        size_t tb = tmp_bytes;
        RTB_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, is_split, split_rank, (int)S + 1, st));
        uint32_t n_split = 0;
        RTB_CHECK(hipMemcpyAsync(&n_split, split_rank + S, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RTB_CHECK(hipStreamSynchronize(st));
        if (n_split == 0) break;
        if ((size_t)n_nodes + 2 * (size_t)n_split > cap) { err = "node table overflow"; return RT_E_LIMIT; }
        const uint32_t n_children = 2 * n_split;
        RTB_CHECK(hipMemsetAsync(child_active, 0, ((size_t)n_children + 1) * sizeof(uint32_t), st));
        RTB_CHECK(hipMemsetAsync(child_count, 0, ((size_t)n_children + 1) * sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_children, blocks(S), dim3(TB), 0, st, S, sn, sb, sf, sc, best_axis, split, split_rank, n_nodes, M,
                           pre[0], pre[1], pre[2], suf[0], suf[1], suf[2], nodes, min_active, child_active, child_count);
        tb = tmp_bytes;
        RTB_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, child_active, child_slot, (int)n_children + 1, st));
        tb = tmp_bytes;
        RTB_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, child_count, child_dense, (int)n_children + 1, st));
        hipLaunchKernelGGL(k_next_segments, blocks(n_children), dim3(TB), 0, st, n_children, n_nodes, child_active, child_slot, child_dense, nodes,
                           seg_node[cur ^ 1], seg_first[cur ^ 1], seg_count[cur ^ 1], seg_base[cur ^ 1]);
        uint32_t next[2] = {0, 0};
        RTB_CHECK(hipMemcpyAsync(&next[0], child_slot + n_children, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RTB_CHECK(hipMemcpyAsync(&next[1], child_dense + n_children, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RTB_CHECK(hipStreamSynchronize(st));
        n_nodes += n_children; S = next[0]; M = next[1]; cur ^= 1;
    }
    RTB_CHECK(hipGetLastError());

    // download the creation-order table and renumber it into the reference's order (bvh.rs:167-175): a node's children take
    // the next two indices at the moment it splits, and the left subtree is finished before the right child is looked at
    std::vector<Box6> h_box(n_nodes); std::vector<uint32_t> h_first(n_nodes), h_count(n_nodes), h_left(n_nodes);
    RTB_CHECK(hipMemcpy(h_box.data(), nodes.box, (size_t)n_nodes * sizeof(Box6), hipMemcpyDeviceToHost));
    RTB_CHECK(hipMemcpy(h_first.data(), nodes.first, (size_t)n_nodes * 4, hipMemcpyDeviceToHost));
    RTB_CHECK(hipMemcpy(h_count.data(), nodes.count, (size_t)n_nodes * 4, hipMemcpyDeviceToHost));
    RTB_CHECK(hipMemcpy(h_left.data(), nodes.left, (size_t)n_nodes * 4, hipMemcpyDeviceToHost));
    RTB_CHECK(hipMemcpy(order_out, order, (size_t)n * 4, hipMemcpyDeviceToHost));
    auto put = [&](uint32_t dst, uint32_t src) {
        rt_bvh_node& o = nodes_out[dst]; memset(&o, 0, sizeof o);
        for (int c = 0; c < 3; ++c) { o.min[c] = h_box[src].mn[c]; o.max[c] = h_box[src].mx[c]; }
        o.is_leaf = 1; o.first = h_first[src]; o.count = h_count[src];
    };
    struct Item { uint32_t src, dst, depth; };
    std::vector<Item> todo; todo.push_back({0u, 0u, 1u});
    put(0, 0);
    uint32_t next_index = 1, depth = 1;
    while (!todo.empty()) {
        const Item it = todo.back(); todo.pop_back();
        if (it.depth > depth) depth = it.depth;
        const uint32_t l = h_left[it.src];
        if (!l) continue;
        const uint32_t dl = next_index, dr = next_index + 1; next_index += 2;
        put(dl, l); put(dr, l + 1);
        nodes_out[it.dst].is_leaf = 0; nodes_out[it.dst].left = dl; nodes_out[it.dst].right = dr;
        todo.push_back({l + 1, dr, it.depth + 1}); todo.push_back({l, dl, it.depth + 1});
    }
    *n_nodes_out = n_nodes; *depth_out = depth;
    return RT_OK;
}

} // namespace rtb

// =================================================================================================
// Grid::new (src/acceleration/grid.rs:23-61, with the fixed bounds of hittable_list.rs:43) on the GPU: the same table as the
// host builder in host/accel.cpp (which the tests compare it with byte for byte).
//   * the reference loops over cell corners x,y,z in (-100..100).step_by(size as usize) and stores each cell at
//     get_index(corner - origin) (grid.rs:64-73), later cells overwriting earlier ones; slots nobody writes keep the
//     clone of the origin cell (box and object list, grid.rs:24-27), and corners within distance 1 of the origin are
//     skipped (grid.rs:41-45).  Here: one thread per loop cell, atomicMax of the loop number per slot;
//   * an object belongs to a cell iff their boxes overlap inclusively (grid.rs:267-271, aabb.rs:113-117), in ascending
//     object order (grid.rs:274-278).  Here: one wave per object walks the cells its box reaches and emits
//     (slot, object) pairs; a radix sort of the pairs gives every slot's list in ascending object order.
namespace rtb {

__host__ __device__ inline unsigned long long sat_usize64(double x) { if (!(x > 0.0)) return 0; if (x >= 18446744073709551616.0) return ~0ull; return (unsigned long long)x; }
__host__ __device__ inline int sat_i32d(double x) { if (x != x) return 0; if (x <= -2147483648.0) return (int)0x80000000; if (x >= 2147483647.0) return 0x7fffffff; return (int)x; }
__host__ __device__ inline unsigned long long grid_get_index(const double p[3], const double bs[3], const double total[3]) {   // grid.rs:64-73
    unsigned long long x = sat_usize64(p[0] / bs[0]), y = sat_usize64(p[1] / bs[1]), z = sat_usize64(p[2] / bs[2]);
    double size_x = total[0] / bs[0], size_y = total[1] / bs[1];
    return sat_usize64((double)x + (double)y * size_x + (double)z * size_x * size_y);
}
struct GridGeom { double bs[3], origin[3], total[3]; long long start[3]; unsigned long long step[3]; uint32_t dim[3]; unsigned long long n_cells; };
__host__ __device__ inline double grid_coord(const GridGeom& g, int c, uint32_t k) { return (double)(g.start[c] + (long long)(k * g.step[c])); }

// owner[slot] = 1 + the largest loop number that maps to the slot (0 = nobody: origin clone)
__global__ void k_grid_owner(GridGeom g, uint32_t n_loop, uint32_t* __restrict__ owner1, uint32_t* __restrict__ bad) {
    uint32_t lc = blockIdx.x * blockDim.x + threadIdx.x;
    if (lc >= n_loop) return;
    const uint32_t iz = lc % g.dim[2], iy = (lc / g.dim[2]) % g.dim[1], ix = lc / (g.dim[2] * g.dim[1]);
    const double mn[3] = {grid_coord(g, 0, ix), grid_coord(g, 1, iy), grid_coord(g, 2, iz)};
    const double dc[3] = {mn[0] - g.origin[0], mn[1] - g.origin[1], mn[2] - g.origin[2]};
    if (dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2] <= 1.0) return;                  // grid.rs:41-45
    const double fin[3] = {(double)sat_i32d(dc[0]), (double)sat_i32d(dc[1]), (double)sat_i32d(dc[2])};
    const unsigned long long index = grid_get_index(fin, g.bs, g.total);
    if (index >= g.n_cells) { *bad = 1; return; }
    atomicMax(&owner1[index], lc + 1u);
}
// slot of each loop cell if it is the final owner (else ~0); box of every slot
__global__ void k_grid_slots(GridGeom g, unsigned long long n_cells, const uint32_t* __restrict__ owner1, unsigned long long* __restrict__ slot_of, double* __restrict__ cell_aabb) {
    unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_cells) return;
    double* a = cell_aabb + idx * 6;
    const uint32_t o = owner1[idx];
    if (!o) { for (int c = 0; c < 3; ++c) { a[c] = g.origin[c]; a[3 + c] = g.origin[c] + g.bs[c]; } return; }   // origin clone
    const uint32_t lc = o - 1, iz = lc % g.dim[2], iy = (lc / g.dim[2]) % g.dim[1], ix = lc / (g.dim[2] * g.dim[1]);
    slot_of[lc] = idx;
    const double mn[3] = {grid_coord(g, 0, ix), grid_coord(g, 1, iy), grid_coord(g, 2, iz)};
    for (int c = 0; c < 3; ++c) { a[c] = mn[c]; a[3 + c] = mn[c] + g.bs[c]; }          // GridBox::new grid.rs:261-264
}
// the cells along one axis whose [co, co + size] overlaps [lo, hi] inclusively: a contiguous run [k0, k1)
__device__ inline void grid_axis_run(const GridGeom& g, int c, double lo, double hi, uint32_t& k0, uint32_t& k1) {
    k0 = k1 = 0; bool open = false;
    for (uint32_t k = 0; k < g.dim[c]; ++k) {
        const double cmin = grid_coord(g, c, k), cmax = cmin + g.bs[c];
        const bool hit = cmax >= lo && cmin <= hi;
        if (hit && !open) { k0 = k; open = true; }
        if (hit) k1 = k + 1;
    }
}
// one wave per object: MODE 0 counts its (slot, object) pairs, MODE 1 writes them (key = slot << 32 | object) from `base[obj]`;
// in both modes the origin flag (overlap with the origin cell's box) is written
template <int MODE>
__global__ __launch_bounds__(64) void k_grid_pairs(GridGeom g, const Box6* __restrict__ boxes, uint32_t n, const unsigned long long* __restrict__ slot_of,
                                                   unsigned long long* __restrict__ pair_count, const unsigned long long* __restrict__ base,
                                                   unsigned long long* __restrict__ keys, uint32_t* __restrict__ in_origin) {
    const uint32_t obj = blockIdx.x, lane = threadIdx.x;
    if (obj >= n) return;
    const Box6 b = boxes[obj];
    __shared__ uint32_t run[6];
    if (lane < 3) { uint32_t k0, k1; grid_axis_run(g, (int)lane, b.mn[lane], b.mx[lane], k0, k1); run[2 * lane] = k0; run[2 * lane + 1] = k1; }
    __syncthreads();
    const uint32_t cx = run[1] - run[0], cy = run[3] - run[2], cz = run[5] - run[4];
    const unsigned long long total = (unsigned long long)cx * cy * cz;
    unsigned long long mine = 0, out_pos = MODE ? base[obj] : 0;
    for (unsigned long long t0 = 0; t0 < total; t0 += 64) {
        const unsigned long long t = t0 + lane;
        bool valid = false; unsigned long long slot = 0;
        if (t < total) {
            const uint32_t iz = run[4] + (uint32_t)(t % cz), iy = run[2] + (uint32_t)((t / cz) % cy), ix = run[0] + (uint32_t)(t / ((unsigned long long)cz * cy));
            slot = slot_of[((unsigned long long)ix * g.dim[1] + iy) * g.dim[2] + iz];
            valid = slot != ~0ull;
        }
        const unsigned long long m = __ballot(valid);
        if (MODE) {
            if (valid) keys[out_pos + __popcll(m & ((1ull << lane) - 1ull))] = (slot << 32) | obj;
            out_pos += __popcll(m);
        } else mine += valid ? 1 : 0;
    }
    if (!MODE) {
        for (int off = 32; off; off >>= 1) mine += __shfl_down(mine, off, 64);
        if (lane == 0) {
            pair_count[obj] = mine;
            bool o = true;                                           // aabb.rs:113-117 against the origin cell's box
            for (int c = 0; c < 3; ++c) o = o && (g.origin[c] + g.bs[c] >= b.mn[c]) && (g.origin[c] <= b.mx[c]);
            in_origin[obj] = o ? 1u : 0u;
        }
    }
}
__global__ void k_grid_count(const unsigned long long* __restrict__ keys, unsigned long long L, unsigned long long* __restrict__ cell_count) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L) atomicAdd(&cell_count[keys[i] >> 32], 1ull);
}
__global__ void k_grid_clone_counts(unsigned long long n_cells, const uint32_t* __restrict__ owner1, unsigned long long n_origin, unsigned long long* __restrict__ cell_count) {
    unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n_cells && !owner1[idx]) cell_count[idx] = n_origin;
}
// sorted pairs -> lists: pair i of slot s lands at offset[s] + (i - first pair of s); the first pair of s is found by a search
__global__ void k_grid_fill(const unsigned long long* __restrict__ keys, unsigned long long L, const unsigned long long* __restrict__ cell_offset,
                            const unsigned long long* __restrict__ pair_start, uint32_t* __restrict__ prim_list) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const unsigned long long s = keys[i] >> 32;
    prim_list[cell_offset[s] + (i - pair_start[s])] = (uint32_t)keys[i];
}
__global__ void k_grid_clone_fill(unsigned long long n_cells, const uint32_t* __restrict__ owner1, const unsigned long long* __restrict__ cell_offset,
                                  const uint32_t* __restrict__ origin_list, uint32_t n_origin, uint32_t* __restrict__ prim_list) {
    unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_cells || owner1[idx]) return;
    for (uint32_t k = 0; k < n_origin; ++k) prim_list[cell_offset[idx] + k] = origin_list[k];
}
__global__ void k_grid_boxes(const rt_prim* __restrict__ prims, uint32_t n, Box6* __restrict__ boxes) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rt_prim p = prims[i]; Box6 b;
    if (p.type == RT_SPHERE) { for (int c = 0; c < 3; ++c) { b.mn[c] = p.v[c] - p.v[3]; b.mx[c] = p.v[c] + p.v[3]; } }
    else for (int c = 0; c < 3; ++c) { b.mn[c] = rtb::min_tie2(rtb::min_tie2(p.v[c], p.v[3 + c]), p.v[6 + c]); b.mx[c] = rtb::max_tie2(rtb::max_tie2(p.v[c], p.v[3 + c]), p.v[6 + c]); }
    boxes[i] = b;
}
__global__ void k_grid_origin_list(const uint32_t* __restrict__ in_origin, const uint32_t* __restrict__ rank, uint32_t n, uint32_t* __restrict__ origin_list) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && in_origin[i]) origin_list[rank[i]] = i;
}

struct GridResult { rt_grid_desc desc; double* cell_aabb; unsigned long long* cell_offset; uint32_t* prim_list; unsigned long long list_len; };

// Device buffers of the result are owned by the caller afterwards (hipFree); everything else is freed here.
int build_grid(const rt_prim* prims, uint32_t n, double s, GridResult& out, hipStream_t st, std::string& err) {
    if (!(s >= 1.0)) { err = "grid size must be >= 1 (the reference's step_by panics on 0)"; return RT_E_ARG; }
    GridGeom g;
    for (int c = 0; c < 3; ++c) { g.bs[c] = s; g.origin[c] = -100.0; g.total[c] = 200.0; }                         // hittable_list.rs:43
    g.n_cells = sat_usize64((g.total[0] / s) * (g.total[1] / s) * (g.total[2] / s));                                 // grid.rs:27
    if (g.n_cells == 0) { err = "grid has no cells"; return RT_E_ARG; }
    unsigned long long n_loop = 1;
    for (int c = 0; c < 3; ++c) {                                                                                     // grid.rs:29-38
        const long long a = sat_i32d(g.origin[c]), b = sat_i32d(100.0); const unsigned long long stp = sat_usize64(g.bs[c]);
        g.start[c] = a; g.step[c] = stp; g.dim[c] = (uint32_t)((b - a + (long long)stp - 1) / (long long)stp);
        n_loop *= g.dim[c];
    }
    if (n_loop >= 0xffffffffull || g.n_cells >= 0xffffffffull) { err = "grid too fine"; return RT_E_LIMIT; }
    memset(&out, 0, sizeof out);
    for (int c = 0; c < 3; ++c) { out.desc.box_size[c] = s; out.desc.origin[c] = -100.0; out.desc.end[c] = 100.0; out.desc.total_size[c] = 200.0; }
    out.desc.n_cells = g.n_cells;

    DevBuf buf;
    const uint32_t TB = 256;
    auto blocks = [&](unsigned long long m) { return dim3((unsigned)((m + TB - 1) / TB)); };
    rt_prim* d_prims; Box6* boxes; uint32_t *owner1, *bad, *in_origin, *rank, *origin_list; unsigned long long *slot_of, *pair_count, *pair_base;
    RTB_ALLOC(d_prims, rt_prim, n); RTB_ALLOC(boxes, Box6, n); RTB_ALLOC(owner1, uint32_t, g.n_cells); RTB_ALLOC(bad, uint32_t, 1);
    RTB_ALLOC(in_origin, uint32_t, (size_t)n + 1); RTB_ALLOC(rank, uint32_t, (size_t)n + 1); RTB_ALLOC(origin_list, uint32_t, n);
    RTB_ALLOC(slot_of, unsigned long long, n_loop); RTB_ALLOC(pair_count, unsigned long long, (size_t)n + 1); RTB_ALLOC(pair_base, unsigned long long, (size_t)n + 1);
    unsigned long long *cell_count, *pair_start;
    RTB_ALLOC(cell_count, unsigned long long, g.n_cells + 1); RTB_ALLOC(pair_start, unsigned long long, g.n_cells + 1);
    // results (kept on success)
    double* cell_aabb = nullptr; unsigned long long* cell_offset = nullptr; uint32_t* prim_list = nullptr;
    auto drop = [&]() { if (cell_aabb) (void)hipFree(cell_aabb); if (cell_offset) (void)hipFree(cell_offset); if (prim_list) (void)hipFree(prim_list); };
#define RTB_GCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + " failed: " + hipGetErrorString(e_); drop(); return RT_E_HIP; } } while (0)
    RTB_GCHECK(hipMalloc(&cell_aabb, g.n_cells * 6 * sizeof(double)));
    RTB_GCHECK(hipMalloc(&cell_offset, (g.n_cells + 1) * sizeof(unsigned long long)));

    if (n) RTB_GCHECK(hipMemcpyAsync(d_prims, prims, (size_t)n * sizeof(rt_prim), hipMemcpyHostToDevice, st));
    RTB_GCHECK(hipMemsetAsync(owner1, 0, g.n_cells * sizeof(uint32_t), st));
    RTB_GCHECK(hipMemsetAsync(bad, 0, sizeof(uint32_t), st));
    RTB_GCHECK(hipMemsetAsync(slot_of, 0xff, n_loop * sizeof(unsigned long long), st));
    RTB_GCHECK(hipMemsetAsync(cell_count, 0, (g.n_cells + 1) * sizeof(unsigned long long), st));
    RTB_GCHECK(hipMemsetAsync(pair_count, 0, ((size_t)n + 1) * sizeof(unsigned long long), st));
    RTB_GCHECK(hipMemsetAsync(in_origin, 0, ((size_t)n + 1) * sizeof(uint32_t), st));
    if (n) hipLaunchKernelGGL(k_grid_boxes, blocks(n), dim3(TB), 0, st, d_prims, n, boxes);
    hipLaunchKernelGGL(k_grid_owner, blocks(n_loop), dim3(TB), 0, st, g, (uint32_t)n_loop, owner1, bad);
    hipLaunchKernelGGL(k_grid_slots, blocks(g.n_cells), dim3(TB), 0, st, g, g.n_cells, owner1, slot_of, cell_aabb);
    if (n) hipLaunchKernelGGL(k_grid_pairs<0>, dim3(n), dim3(64), 0, st, g, boxes, n, slot_of, pair_count, pair_base, (unsigned long long*)nullptr, in_origin);

    size_t tmp_bytes = 0, need = 0;
    RTB_GCHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, pair_count, pair_base, (int)n + 1, st)); tmp_bytes = std::max(tmp_bytes, need);
    RTB_GCHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, in_origin, rank, (int)n + 1, st)); tmp_bytes = std::max(tmp_bytes, need);
    RTB_GCHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, cell_count, cell_offset, (int)(g.n_cells + 1), st)); tmp_bytes = std::max(tmp_bytes, need);
    void* tmp; RTB_ALLOC(tmp, unsigned char, tmp_bytes);
    size_t tb = tmp_bytes;
    RTB_GCHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, pair_count, pair_base, (int)n + 1, st));
    tb = tmp_bytes;
    RTB_GCHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, in_origin, rank, (int)n + 1, st));
    unsigned long long L = 0; uint32_t n_origin = 0, h_bad = 0;
    RTB_GCHECK(hipMemcpyAsync(&L, pair_base + n, sizeof L, hipMemcpyDeviceToHost, st));
    RTB_GCHECK(hipMemcpyAsync(&n_origin, rank + n, sizeof n_origin, hipMemcpyDeviceToHost, st));
    RTB_GCHECK(hipMemcpyAsync(&h_bad, bad, sizeof h_bad, hipMemcpyDeviceToHost, st));
    RTB_GCHECK(hipStreamSynchronize(st));
    if (h_bad) { err = "grid size does not tile the 200^3 world (the reference panics: index out of bounds)"; drop(); return RT_E_ARG; }
    if (L >= 0x7fffffffull) { err = "grid object lists too long for the device builder"; drop(); return RT_E_LIMIT; }
    if (n) hipLaunchKernelGGL(k_grid_origin_list, blocks(n), dim3(TB), 0, st, in_origin, rank, n, origin_list);

    unsigned long long *keys_a = nullptr, *keys_b = nullptr;
    if (L) {
        RTB_ALLOC(keys_a, unsigned long long, L); RTB_ALLOC(keys_b, unsigned long long, L);
        hipLaunchKernelGGL(k_grid_pairs<1>, dim3(n), dim3(64), 0, st, g, boxes, n, slot_of, pair_count, pair_base, keys_a, in_origin);
        size_t sort_bytes = 0;
        RTB_GCHECK(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, keys_a, keys_b, (int)L, 0, 64, st));
        void* sort_tmp; RTB_ALLOC(sort_tmp, unsigned char, sort_bytes);
        RTB_GCHECK(hipcub::DeviceRadixSort::SortKeys(sort_tmp, sort_bytes, keys_a, keys_b, (int)L, 0, 64, st));
        hipLaunchKernelGGL(k_grid_count, blocks(L), dim3(TB), 0, st, keys_b, L, cell_count);
    }
    // first pair of every slot (before the clones are counted in), then the final CSR offsets with the clones
    tb = tmp_bytes;
    RTB_GCHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, cell_count, pair_start, (int)(g.n_cells + 1), st));
    hipLaunchKernelGGL(k_grid_clone_counts, blocks(g.n_cells), dim3(TB), 0, st, g.n_cells, owner1, (unsigned long long)n_origin, cell_count);
    tb = tmp_bytes;
    RTB_GCHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, cell_count, cell_offset, (int)(g.n_cells + 1), st));
    unsigned long long len = 0;
    RTB_GCHECK(hipMemcpyAsync(&len, cell_offset + g.n_cells, sizeof len, hipMemcpyDeviceToHost, st));
    RTB_GCHECK(hipStreamSynchronize(st));
    RTB_GCHECK(hipMalloc(&prim_list, (len ? len : 1) * sizeof(uint32_t)));
    if (L) hipLaunchKernelGGL(k_grid_fill, blocks(L), dim3(TB), 0, st, keys_b, L, cell_offset, pair_start, prim_list);
    if (n_origin) hipLaunchKernelGGL(k_grid_clone_fill, blocks(g.n_cells), dim3(TB), 0, st, g.n_cells, owner1, cell_offset, origin_list, n_origin, prim_list);
    RTB_GCHECK(hipStreamSynchronize(st));
    RTB_GCHECK(hipGetLastError());
#undef RTB_GCHECK
    out.cell_aabb = cell_aabb; out.cell_offset = cell_offset; out.prim_list = prim_list; out.list_len = len;
    return RT_OK;
}

} // namespace rtb
