// rt_device.h — device-side math, RNG, primitive tests and scatter functions (gfx950, f64).
//
// Every function cites the reference lines whose arithmetic it reproduces (paths under the reference
// repo's src/).  The file must be compiled with -ffp-contract=off: the reference never fuses a
// multiply-add, and hit/miss decisions are compared bit-for-bit against the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_DEV __device__ __forceinline__

struct V3 { double x, y, z; };
RT_DEV V3 v3(double a, double b, double c) { V3 r; r.x = a; r.y = b; r.z = c; return r; }
RT_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }            // value/vec3.rs:164-169
RT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }            // :171-177
RT_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }                                 // :114-120
RT_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }            // :179-185
RT_DEV V3 operator*(V3 a, double t) { return v3(a.x * t, a.y * t, a.z * t); }              // :187-193
RT_DEV V3 operator*(double t, V3 a) { return a * t; }                                      // :195-201
RT_DEV V3 operator/(V3 a, double t) { return (1.0 / t) * a; }                              // :203-209 (reciprocal, then multiply)
RT_DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                // :97-99
RT_DEV double length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }           // :88-90
RT_DEV V3 unit(V3 a) { return a / sqrt(length_squared(a)); }                               // :16-18, :84-86
RT_DEV V3 cross(V3 a, V3 b) {                                                              // :101-107
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_DEV bool near_zero(V3 a) { const double s = 1e-8; return fabs(a.x) < s && fabs(a.y) < s && fabs(a.z) < s; } // :92-95
RT_DEV V3 reflect(V3 v, V3 n) { return v - 2.0 * dot(v, n) * n; }                          // :53-55
RT_DEV V3 refract(V3 uv, V3 n, double etai_over_etat) {                                    // :57-62
    double cos_theta = fmin(dot(-uv, n), 1.0);
    V3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
    V3 r_out_parallel = -sqrt(fabs(1.0 - length_squared(r_out_perp))) * n;
    return r_out_perp + r_out_parallel;
}

// ---------------------------------------------------------------------------------------------
// Counter-based RNG replacing thread_rng() (utils/rtweekend.rs:166-169): Philox4x32-10,
// key = seed, counter = (path_lo = pixel index j*W+i, path_hi = global sample index, stream, block).
// stream 0 = camera ray, stream 1+b = scatter at bounce b.  Block n yields draws 2n and 2n+1.
// A draw is 53 bits * 2^-53 in [0,1), the rand 0.8 `Standard` f64 distribution.
RT_DEV void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r) { k0 += W0; k1 += W1; }
        const uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];      // one v_mad_u64_u32 each instead of mul_hi + mul_lo
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        // three-input xor in one instruction (gfx950 v_bitop3_b32, truth table 0x96); the compiler emits two v_xor_b32 otherwise
        uint32_t n0 = __builtin_amdgcn_bitop3_b32(h1, c[1], k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(h0, c[3], k1, 0x96);
        c[0] = n0; c[1] = l1; c[2] = n2; c[3] = l0;
    }
}
struct Rng {
    uint32_t k0, k1, lo, hi, stream, k; uint32_t blk[4];
    RT_DEV Rng(uint64_t seed, uint32_t path_lo, uint32_t path_hi, uint32_t st)
        : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), lo(path_lo), hi(path_hi), stream(st), k(0) {}
    RT_DEV double next() {
        if (!(k & 1)) {
            blk[0] = lo; blk[1] = hi; blk[2] = stream; blk[3] = k >> 1;
            philox4x32_10(blk, k0, k1);
        }
        uint64_t u = (k & 1) ? (((uint64_t)blk[3] << 32) | blk[2]) : (((uint64_t)blk[1] << 32) | blk[0]);
        ++k;
        return (double)(u >> 11) * (1.0 / 9007199254740992.0);
    }
    RT_DEV double range(double a, double b) { return a + (b - a) * next(); }               // rtweekend.rs:171-173
};
RT_DEV V3 random_in_unit_disk(Rng& g) {                                                    // vec3.rs:20-31
    for (;;) {
        double a = g.range(-1.0, 1.0);
        double b = g.range(-1.0, 1.0);
        V3 p = v3(a, b, 0.0);
        if (length_squared(p) < 1.0) return p;
    }
}
RT_DEV V3 random_unit(Rng& g) {                                                            // vec3.rs:33-41
    for (;;) {
        double a = g.range(-1.0, 1.0);
        double b = g.range(-1.0, 1.0);
        double c = g.range(-1.0, 1.0);
        V3 p = v3(a, b, c);
        double lsq = length_squared(p);
        if (1e-160 < lsq && lsq <= 1.0) return p / sqrt(lsq);
    }
}


// ---------------------------------------------------------------------------------------------
// Wave-cooperative form of the two rejection samplers.  A wave that runs `random_unit` lane by lane loops as long as its
// unluckiest lane (6.4 tries expected for 64 lanes, 1.9 for one); here the lanes that are done compute further tries
// of the lanes that are not.  The tries of a path are fixed by its counter (try t of random_unit = draws 3t..3t+2, try t
// of random_in_unit_disk = draws first+2t, first+2t+1), so any lane can evaluate any path's try, and the owner takes its
// first accepted try in order: the result is bit for bit what the sequential loop returns.
RT_DEV double draw_of(uint32_t lo32, uint32_t hi32) { return (double)((((uint64_t)hi32 << 32) | lo32) >> 11) * (1.0 / 9007199254740992.0); }
RT_DEV void rng_block(uint32_t k0, uint32_t k1, uint32_t lo, uint32_t hi, uint32_t stream, uint32_t n, double& a, double& b) {
    uint32_t c[4] = {lo, hi, stream, n};
    philox4x32_10(c, k0, k1);
    a = draw_of(c[0], c[1]); b = draw_of(c[2], c[3]);
}
struct CoopLds { uint2 owner[64]; double px[64], py[64], pz[64]; };       // one per wave
RT_DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// One work item of a path's sampler.  DIM 3: tries 2j and 2j+1 of random_unit (blocks 3j..3j+2 of the stream, first draw 0),
// returns the first accepted candidate p (not yet normalised).  DIM 2: try j of random_in_unit_disk (block base + j).
template <int DIM>
RT_DEV bool reject_item(uint32_t k0, uint32_t k1, uint32_t lo, uint32_t hi, uint32_t stream, uint32_t base, uint32_t j, V3& p) {
    if (DIM == 2) {
        double a, b; rng_block(k0, k1, lo, hi, stream, base + j, a, b);
        p = v3(-1.0 + 2.0 * a, -1.0 + 2.0 * b, 0.0);
        return length_squared(p) < 1.0;                                    // vec3.rs:27
    } else {
        double d0, d1, d2, d3, d4, d5;
        rng_block(k0, k1, lo, hi, stream, base + 3u * j, d0, d1);
        rng_block(k0, k1, lo, hi, stream, base + 3u * j + 1u, d2, d3);
        rng_block(k0, k1, lo, hi, stream, base + 3u * j + 2u, d4, d5);
        const V3 pa = v3(-1.0 + 2.0 * d0, -1.0 + 2.0 * d1, -1.0 + 2.0 * d2), pb = v3(-1.0 + 2.0 * d3, -1.0 + 2.0 * d4, -1.0 + 2.0 * d5);
        const double la = length_squared(pa), lb = length_squared(pb);
        const bool aa = 1e-160 < la && la <= 1.0, ab = 1e-160 < lb && lb <= 1.0;   // vec3.rs:37
        p = aa ? pa : pb;
        return aa || ab;
    }
}
// Every lane of the wave must call this (converged); `need` says whether the lane wants a sample.  Returns the accepted
// candidate of the lane's own path.
template <int DIM>
RT_DEV V3 coop_reject(bool need, uint32_t k0, uint32_t k1, uint32_t lo, uint32_t hi, uint32_t stream, uint32_t base, CoopLds* L) {
    V3 res = v3(0, 0, 0);
    if (__builtin_amdgcn_ballot_w64(need) == 0) return res;
    const uint32_t lane = threadIdx.x & 63;
    bool undone = need;
    { V3 p; const bool acc = reject_item<DIM>(k0, k1, lo, hi, stream, base, 0, p); if (undone && acc) { res = p; undone = false; } }
    uint32_t next = 1;
#ifdef RT_NO_COOP
    while (undone) { V3 p; if (reject_item<DIM>(k0, k1, lo, hi, stream, base, next, p)) { res = p; undone = false; } ++next; }   // A/B only
#endif
    for (;;) {
        const unsigned long long U = __builtin_amdgcn_ballot_w64(undone);
        if (U == 0) break;
        const uint32_t n_u = (uint32_t)__builtin_popcountll(U);
        const uint32_t lg = 31u - (uint32_t)__builtin_clz(64u / n_u);     // 2^lg helpers per unfinished path
        const uint32_t rank = (uint32_t)__builtin_popcountll(U & ((1ull << lane) - 1ull));
        if (undone) L->owner[rank] = make_uint2(lo, hi);
        wave_sync();
        const uint32_t orank = lane >> lg, off = lane & ((1u << lg) - 1u);
        const bool helper = orank < n_u;
        const uint2 ow = L->owner[helper ? orank : 0u];
        V3 p; const bool acc = reject_item<DIM>(k0, k1, ow.x, ow.y, stream, base, next + off, p) && helper;
        L->px[lane] = p.x; L->py[lane] = p.y; L->pz[lane] = p.z;
        const unsigned long long A = __builtin_amdgcn_ballot_w64(acc);
        wave_sync();
        if (undone) {
            const unsigned long long mine = (A >> (rank << lg)) & (lg == 6u ? ~0ull : ((1ull << (1u << lg)) - 1ull));
            if (mine) { const uint32_t h = (rank << lg) + (uint32_t)__builtin_ctzll(mine); res = v3(L->px[h], L->py[h], L->pz[h]); undone = false; }
        }
        wave_sync();
        next += 1u << lg;
    }
    return res;
}
// ---------------------------------------------------------------------------------------------
// Device scene records.
struct __align__(16) DPrim {       // 80 B: one Hittable (sphere.rs:13-17 / triangle.rs:12-18)
    double v[9];                   // sphere: center xyz, radius.  triangle: v0, v1, v2
    uint32_t tm;                   // bit 31 = type (1 triangle), bits 0..30 = material index
    uint32_t orig;                 // file-order object index
};
struct __align__(8) DMat { double albedo[3]; double param; uint32_t type; uint32_t pad; };  // material.rs:23-25,54-57,82-84
// 64 B BVH node (bvh.rs:35-42 flattened).  Siblings are adjacent (bvh.rs:167-168: left = ptr+1, right = ptr+2).
//   internal: a = index of the left child (right = a + 1), b = 0
//   leaf:     a = first object (index into the BVH-ordered DPrim array), b = count | RT_LEAF
struct __align__(16) DNode { double mn[3]; double mx[3]; uint32_t a, b, pad0, pad1; };
#define RT_LEAF 0x80000000u
// Compact forms for scenes whose coordinates all lie on an f32 lattice, c == (f64)f32 * scale bit for bit (PLY meshes:
// parser.rs:67-73 widens f32 vertices and multiplies by 100; box planes are min/max of vertex coordinates, so they lie on
// the same lattice).  The device widens and scales with the same two IEEE operations, so every value it computes with is
// identical to the f64 one — the point is that a divergent 16-byte lane access costs the vector L1 one cycle each, and a
// sibling pair is then 4 accesses instead of 8, a triangle 3 instead of 6.
// scale is 100 or 1: an f32 (24 significant bits) times either is exact in f64, so plane - o can also be formed as one
// fma(f, scale, -o): with an exact product the single rounding of the fma is the rounding of the subtraction.
//   DPair32: children of one internal node.  a/b as in DNode, except that an internal child's `a` is a pair index.
//   DPair64: the same pair with f64 planes for scenes off the lattice: 112 B = 7 accesses (two 64-B DNodes would be 8)
//   DSphere: 48-B sphere for all-sphere scenes (3 accesses; the generic 80-B DPrim is 5)
struct __align__(16) DPair64 { double box[12]; uint32_t la, lb, ra, rb; };     // box: L.mn, L.mx, R.mn, R.mx
struct __align__(16) DSphere { double c[3]; double r; uint32_t tm, orig, pad0, pad1; };
struct __align__(16) DPair32 { float box[12]; uint32_t la, lb, ra, rb; };      // box: L.mn, L.mx, R.mn, R.mx
// 64-byte stride (48 bytes read): two records per 128-byte line, none straddling a line boundary (at 48-byte stride a quarter of them
// did; extend 549.7 -> 540.9 ms on the headline workload)
struct __align__(16) DPrim32 { float v[9]; uint32_t tm, orig, pad; uint32_t unused[4]; };   // triangles only
#define RT_PRIM32_READ_BYTES 48u
// DPairN: a sibling pair in TWO 16-byte accesses.  A node's box is the union of its children's boxes (objects_to_aabb folds
// min/max, hittable_list.rs:123-133), so on each of the six faces (mn.x, mn.y, mn.z, mx.x, mx.y, mx.z) at least one child's
// plane IS the parent's plane, bit for bit; the record keeps only the other child's plane on that face ("new" plane) and one
// selector bit saying which child owns it.  The lane carries the box of the node it stands on (six f32 words, the planes it
// tested on the way down) and rebuilds both children's planes with one bit-field select each; the values it then computes with
// are the very f32 lattice planes the 64-byte DPair32 holds.  A divergent 16-byte lane access is what the vector L1 serves at
// about one per clock and CU (profiles/round1_mb_gather.txt), and one more access per visit costs the extend kernel +28 %.
// MEASURED (profiles/round3_extend_variants.md): the waves wait less (0.54 -> 0.43 of their cycles) but the rebuild's ~44 VALU
// per visit push VALU issue from 0.67 to 0.80 and the step is 8.5 % slower.  Opt-in (RT_PAIRN=1), not the default.
//   nw[0..5]  new plane per face, f32 bits
//   wl, wr    child words: bit 31 leaf | bits 28-30 selector bits (wl: faces 0-2, wr: faces 3-5; 1 = the LEFT child owns the
//             new plane) | bits 4-27 pair index of the child's children / first primitive of a leaf | bits 0-3 leaf count
//   own[0..5] the box of the node whose children this record describes (f32 bits): read only when a popped continuation
//             enters an internal node, whose box the lane no longer holds
// Only for lattice scenes whose tree fits the fields (< 2^24 pairs and primitives, leaves of <= 15) and whose planes inherit
// bit for bit (a +0 / -0 pair of equal planes would not): anything else keeps DPair32.
struct __align__(16) DPairN { uint32_t nw[6]; uint32_t wl, wr; uint32_t own[6]; uint32_t pad0, pad1; };

// The part of HitRecord (hittable.rs:13-21) that survives a hit() call and that later code reads:
// rec.t (read back as closest_so_far, bvh.rs:200,215 / grid.rs:164), the t at which rec.p was last written
// (rec.p == at(tp)), and the last accepted primitive (fixes normal, front_face and mat).
// `tn` is the t at which that primitive was accepted: a sphere's normal is built from the hit point of ITS
// acceptance (sphere.rs:50-52) even if a later failed triangle test moved rec.p (triangle.rs:59).
struct HitState { double t; double tp; double tn; int32_t prim; };

// sphere.rs:27-56.  Writes the record only on success.
RT_DEV bool hit_sphere(const DPrim& s, int32_t index, V3 o, V3 d, double tmin, double tmax, HitState& rec) {
    V3 center = v3(s.v[0], s.v[1], s.v[2]);
    double radius = s.v[3];
    V3 oc = center - o;
    double a = length_squared(d);
    double h = dot(d, oc);
    double c = length_squared(oc) - radius * radius;
    double discriminant = h * h - a * c;
    if (discriminant < 0.0) return false;
    double sq = sqrt(discriminant);
    double root = (h - sq) / a;
    if (!(tmin < root && root < tmax)) {
        root = (h + sq) / a;
        if (!(tmin < root && root < tmax)) return false;
    }
    rec.t = root; rec.tp = root; rec.tn = root; rec.prim = index;
    return true;
}
// triangle.rs:37-81, quirks included: the interval is tested against n.d (:44-47), rec.t is written before
// acceptance (:52), rec.p is written once t >= 0 (:59), and only then the three edge tests run (:62-75).
RT_DEV bool hit_triangle(const DPrim& tr, int32_t index, V3 o, V3 d, double tmin, double tmax, HitState& rec) {
    V3 v0 = v3(tr.v[0], tr.v[1], tr.v[2]), v1 = v3(tr.v[3], tr.v[4], tr.v[5]), v2 = v3(tr.v[6], tr.v[7], tr.v[8]);
    V3 v0v1 = v1 - v0;
    V3 v0v2 = v2 - v0;
    V3 n = cross(v0v1, v0v2);
    double n_dot_dir = dot(n, d);
    if (!(tmin < n_dot_dir && n_dot_dir < tmax)) return false;
    double dd = -dot(n, v0);
    double t = -(dot(n, o) + dd) / n_dot_dir;
    rec.t = t;
    if (t < 0.0) return false;
    rec.tp = t;
    V3 p = o + (t * d);
    V3 v0p = p - v0;
    if (dot(n, cross(v0v1, v0p)) <= 0.0) return false;
    V3 v1v2 = v2 - v1;
    V3 v1p = p - v1;
    if (dot(n, cross(v1v2, v1p)) <= 0.0) return false;
    V3 v2v0 = v0 - v2;
    V3 v2p = p - v2;
    if (dot(n, cross(v2v0, v2p)) <= 0.0) return false;
    rec.prim = index; rec.tn = t;
    return true;
}
// hit_triangle without early returns, for the persistent extend kernel: a wave runs the test with dozens of lanes, so a
// lane's early-out saves nothing (the wave executes every level anyway) and each nested exit costs mask bookkeeping.
// Same arithmetic, same order; the record is updated through selects.  (NaN: `t < 0.0` and `x <= 0.0` are false for
// NaN in the reference as well, so the negated forms below keep its behaviour.)
RT_DEV bool hit_triangle_flat(const DPrim& tr, int32_t index, V3 o, V3 d, double tmin, double tmax, HitState& rec) {
    V3 v0 = v3(tr.v[0], tr.v[1], tr.v[2]), v1 = v3(tr.v[3], tr.v[4], tr.v[5]), v2 = v3(tr.v[6], tr.v[7], tr.v[8]);
    V3 v0v1 = v1 - v0;
    V3 v0v2 = v2 - v0;
    V3 n = cross(v0v1, v0v2);
    double n_dot_dir = dot(n, d);
    const bool c1 = tmin < n_dot_dir && n_dot_dir < tmax;                 // :44-47
    double dd = -dot(n, v0);
    double t = -(dot(n, o) + dd) / n_dot_dir;                             // :52
    const bool c2 = c1 && !(t < 0.0);                                     // :55-57
    V3 p = o + (t * d);
    V3 v0p = p - v0;
    const bool e0 = !(dot(n, cross(v0v1, v0p)) <= 0.0);                   // :62-65
    V3 v1v2 = v2 - v1;
    V3 v1p = p - v1;
    const bool e1 = !(dot(n, cross(v1v2, v1p)) <= 0.0);                   // :67-70
    V3 v2v0 = v0 - v2;
    V3 v2p = p - v2;
    const bool e2 = !(dot(n, cross(v2v0, v2p)) <= 0.0);                   // :72-75
    const bool acc = c2 && e0 && e1 && e2;
    rec.t = c1 ? t : rec.t;
    rec.tp = c2 ? t : rec.tp;
    rec.tn = acc ? t : rec.tn;
    rec.prim = acc ? index : rec.prim;
    return acc;
}
RT_DEV bool hit_prim_flat(const DPrim& p, int32_t index, V3 o, V3 d, double tmin, double tmax, HitState& rec) {
    return (p.tm >> 31) ? hit_triangle_flat(p, index, o, d, tmin, tmax, rec) : hit_sphere(p, index, o, d, tmin, tmax, rec);
}
// Triangle::hit with the ray-independent part hoisted: n = (v1-v0)x(v2-v0) (triangle.rs:39-41) and d = -n.v0 (:50) are pure
// functions of the triangle, computed once per scene by k_tri_pre with the very expressions hit_triangle uses, so every value
// is the one the reference recomputes per test.  The naive and grid walks test hundreds to thousands of triangles per ray and
// half of them leave at the n.dir test (:44-47): those now cost one 32-byte read and a dot product, and never touch the vertices.
struct __align__(16) DTriPre { double n[3]; double dd; };
RT_DEV bool hit_triangle_pre(const DPrim* __restrict__ prims, const DTriPre* __restrict__ pre, uint32_t pi, V3 o, V3 d, double tmin, double tmax, HitState& rec) {
    const DTriPre q = pre[pi];
    const V3 n = v3(q.n[0], q.n[1], q.n[2]);
    double n_dot_dir = dot(n, d);
    if (!(tmin < n_dot_dir && n_dot_dir < tmax)) return false;
    double t = -(dot(n, o) + q.dd) / n_dot_dir;
    rec.t = t;
    if (t < 0.0) return false;
    rec.tp = t;
    const DPrim tr = prims[pi];
    V3 v0 = v3(tr.v[0], tr.v[1], tr.v[2]), v1 = v3(tr.v[3], tr.v[4], tr.v[5]), v2 = v3(tr.v[6], tr.v[7], tr.v[8]);
    V3 p = o + (t * d);
    V3 v0v1 = v1 - v0;
    V3 v0p = p - v0;
    if (dot(n, cross(v0v1, v0p)) <= 0.0) return false;
    V3 v1v2 = v2 - v1;
    V3 v1p = p - v1;
    if (dot(n, cross(v1v2, v1p)) <= 0.0) return false;
    V3 v2v0 = v0 - v2;
    V3 v2p = p - v2;
    if (dot(n, cross(v2v0, v2p)) <= 0.0) return false;
    rec.prim = (int32_t)pi; rec.tn = t;
    return true;
}

// The same test for the grid walk, with the triangle's plane record stored IN the cell's list (one 48-byte entry per list
// position: the plane and the primitive index): a list entry is then one sequential load instead of the chain
// list entry -> plane record, and the vertices are fetched only when the n.dir test passes.
struct __align__(16) DCellTri { double n[3]; double dd; uint32_t pi; uint32_t pad[3]; };
RT_DEV bool hit_triangle_cell(const DPrim* __restrict__ prims, const DCellTri& q, V3 o, V3 d, double tmin, double tmax, HitState& rec) {
    const V3 n = v3(q.n[0], q.n[1], q.n[2]);
    double n_dot_dir = dot(n, d);
    if (!(tmin < n_dot_dir && n_dot_dir < tmax)) return false;
    double t = -(dot(n, o) + q.dd) / n_dot_dir;
    rec.t = t;
    if (t < 0.0) return false;
    rec.tp = t;
    const DPrim tr = prims[q.pi];
    V3 v0 = v3(tr.v[0], tr.v[1], tr.v[2]), v1 = v3(tr.v[3], tr.v[4], tr.v[5]), v2 = v3(tr.v[6], tr.v[7], tr.v[8]);
    V3 p = o + (t * d);
    V3 v0v1 = v1 - v0;
    V3 v0p = p - v0;
    if (dot(n, cross(v0v1, v0p)) <= 0.0) return false;
    V3 v1v2 = v2 - v1;
    V3 v1p = p - v1;
    if (dot(n, cross(v1v2, v1p)) <= 0.0) return false;
    V3 v2v0 = v0 - v2;
    V3 v2p = p - v2;
    if (dot(n, cross(v2v0, v2p)) <= 0.0) return false;
    rec.prim = (int32_t)q.pi; rec.tn = t;
    return true;
}

// The list of a well-filled grid cell, 64 entries at a time, one field after the other (SoA): lane l of a wave reads entry l
// of every field, so each load instruction of the cooperative cell test (k_extend_grid_coop) is one contiguous 256- or
// 512-byte run.  n / dd are the triangle's plane as k_tri_pre computes it (triangle.rs:39-41,50), v its vertices, pi its index.
// Entries past the end of the list have n = 0: n.d = 0 fails `0.001 < n.d` (triangle.rs:44-47), so they do nothing.
struct __align__(128) DCellChunk { double n[3][64]; double dd[64]; double v[9][64]; uint32_t pi[64]; uint32_t pad[64]; };

RT_DEV bool hit_prim(const DPrim& p, int32_t index, V3 o, V3 d, double tmin, double tmax, HitState& rec) {
    return (p.tm >> 31) ? hit_triangle(p, index, o, d, tmin, tmax, rec) : hit_sphere(p, index, o, d, tmin, tmax, rec);
}

// AABB::hit (acceleration/aabb.rs:28-62) with 1/dir hoisted out (1.0/dir[axis] is a pure function of the ray,
// so computing it once per ray gives the same bits as computing it per box, aabb.rs:34).  The per-axis
// early return (:54) is folded into one test at the end: ray_t.min only grows and ray_t.max only shrinks, and
// neither can become NaN (a NaN t0/t1 fails every comparison and updates nothing), so `max < min` after axis k
// implies `max < min` after axis 2.  Returns the clipped interval.
RT_DEV bool slab(const double* mn, const double* mx, V3 o, V3 inv, double& tmin, double& tmax) {
    double t0, t1, lo, hi;
    t0 = (mn[0] - o.x) * inv.x; t1 = (mx[0] - o.x) * inv.x;
    lo = (t0 < t1) ? t0 : t1; hi = (t0 < t1) ? t1 : t0;
    if (lo > tmin) tmin = lo; if (hi < tmax) tmax = hi;
    t0 = (mn[1] - o.y) * inv.y; t1 = (mx[1] - o.y) * inv.y;
    lo = (t0 < t1) ? t0 : t1; hi = (t0 < t1) ? t1 : t0;
    if (lo > tmin) tmin = lo; if (hi < tmax) tmax = hi;
    t0 = (mn[2] - o.z) * inv.z; t1 = (mx[2] - o.z) * inv.z;
    lo = (t0 < t1) ? t0 : t1; hi = (t0 < t1) ? t1 : t0;
    if (lo > tmin) tmin = lo; if (hi < tmax) tmax = hi;
    return !(tmax < tmin);
}

// The same test written with v_min_f64 / v_max_f64 (4 instructions per axis instead of 11).  Identical to slab()
// whenever no t0/t1 is NaN: lo/hi are then min/max of (t0,t1), and `if (lo > tmin) tmin = lo` is max(tmin, lo).
// A NaN t0/t1 needs 0 * inf, i.e. a non-finite 1/dir component (dir component zero, denormal or NaN) or a
// non-finite origin: rays for which ray_is_regular() is false must use slab().  (Signed zeros: min/max may pick
// either zero of an equal pair; the interval ends are only ever compared, and tmin >= 0.001, so it cannot matter.)
// The instructions are named directly: fmin()/fmax() make the compiler canonicalise every operand first (one extra
// v_max_f64 x,x each), which only matters for signalling NaNs and no NaN reaches this path.
RT_DEV double vmin_f64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_DEV double vmax_f64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_DEV bool slab_fast(const double* mn, const double* mx, V3 o, V3 inv, double& tmin, double& tmax) {
    double t0, t1;
    t0 = (mn[0] - o.x) * inv.x; t1 = (mx[0] - o.x) * inv.x;
    tmin = vmax_f64(tmin, vmin_f64(t0, t1)); tmax = vmin_f64(tmax, vmax_f64(t0, t1));
    t0 = (mn[1] - o.y) * inv.y; t1 = (mx[1] - o.y) * inv.y;
    tmin = vmax_f64(tmin, vmin_f64(t0, t1)); tmax = vmin_f64(tmax, vmax_f64(t0, t1));
    t0 = (mn[2] - o.z) * inv.z; t1 = (mx[2] - o.z) * inv.z;
    tmin = vmax_f64(tmin, vmin_f64(t0, t1)); tmax = vmin_f64(tmax, vmax_f64(t0, t1));
    return !(tmax < tmin);
}
// The same two tests on precomputed differences dmn = mn - o, dmx = mx - o (the first operation of every axis above)
RT_DEV bool slab_d(const double* dmn, const double* dmx, V3 inv, double& tmin, double& tmax) {
    double t0, t1, lo, hi;
    t0 = dmn[0] * inv.x; t1 = dmx[0] * inv.x;
    lo = (t0 < t1) ? t0 : t1; hi = (t0 < t1) ? t1 : t0;
    if (lo > tmin) tmin = lo; if (hi < tmax) tmax = hi;
    t0 = dmn[1] * inv.y; t1 = dmx[1] * inv.y;
    lo = (t0 < t1) ? t0 : t1; hi = (t0 < t1) ? t1 : t0;
    if (lo > tmin) tmin = lo; if (hi < tmax) tmax = hi;
    t0 = dmn[2] * inv.z; t1 = dmx[2] * inv.z;
    lo = (t0 < t1) ? t0 : t1; hi = (t0 < t1) ? t1 : t0;
    if (lo > tmin) tmin = lo; if (hi < tmax) tmax = hi;
    return !(tmax < tmin);
}
RT_DEV bool slab_fast_d(const double* dmn, const double* dmx, V3 inv, double& tmin, double& tmax) {
    double t0, t1;
    t0 = dmn[0] * inv.x; t1 = dmx[0] * inv.x;
    tmin = vmax_f64(tmin, vmin_f64(t0, t1)); tmax = vmin_f64(tmax, vmax_f64(t0, t1));
    t0 = dmn[1] * inv.y; t1 = dmx[1] * inv.y;
    tmin = vmax_f64(tmin, vmin_f64(t0, t1)); tmax = vmin_f64(tmax, vmax_f64(t0, t1));
    t0 = dmn[2] * inv.z; t1 = dmx[2] * inv.z;
    tmin = vmax_f64(tmin, vmin_f64(t0, t1)); tmax = vmin_f64(tmax, vmax_f64(t0, t1));
    return !(tmax < tmin);
}
// slab_fast_d for the two boxes of a sibling pair with the two dependency chains interleaved: consecutive v_min/v_max_f64
// then never depend on each other, so the hazard wait states (s_nop) the compiler puts between dependent ones disappear.
// Same operations on the same operands as two slab_fast_d calls.
RT_DEV void slab_fast_d2(const double* Lmn, const double* Lmx, const double* Rmn, const double* Rmx, V3 inv,
                         double& ltmin, double& ltmax, double& rtmin, double& rtmax, bool& lh, bool& rh) {
    const double iv[3] = {inv.x, inv.y, inv.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double l0 = Lmn[a] * iv[a], l1 = Lmx[a] * iv[a], r0 = Rmn[a] * iv[a], r1 = Rmx[a] * iv[a];
        const double ln = vmin_f64(l0, l1), rn = vmin_f64(r0, r1), lf = vmax_f64(l0, l1), rf = vmax_f64(r0, r1);
        ltmin = vmax_f64(ltmin, ln); rtmin = vmax_f64(rtmin, rn);
        ltmax = vmin_f64(ltmax, lf); rtmax = vmin_f64(rtmax, rf);
    }
    lh = !(ltmax < ltmin); rh = !(rtmax < rtmin);
}
RT_DEV bool is_finite(double x) { return fabs(x) < __builtin_huge_val(); }     // false for inf and NaN
RT_DEV bool ray_is_regular(V3 o, V3 inv) {
    return is_finite(o.x) && is_finite(o.y) && is_finite(o.z) && is_finite(inv.x) && is_finite(inv.y) && is_finite(inv.z);
}

// AABB::is_edge (aabb.rs:90-97): the point lies within 0.01 of a min or max face on at least two axes (draw-boxes only)
RT_DEV bool box_is_edge(const double* mn, const double* mx, V3 p) {
    const double pt[3] = {p.x, p.y, p.z};
    int count = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
        if ((pt[a] <= mn[a] + 0.01 && pt[a] >= mn[a] - 0.01) || (pt[a] <= mx[a] + 0.01 && pt[a] >= mx[a] - 0.01)) ++count;
    return count >= 2;
}

// value/material.rs:120-124; powi(5) expands to x * ((x*x)*(x*x))
RT_DEV double reflectance(double cosine, double refraction_index) {
    double r0 = (1.0 - refraction_index) / (1.0 + refraction_index);
    r0 = r0 * r0;
    double x = 1.0 - cosine;
    double x2 = x * x;
    double x4 = x2 * x2;
    return r0 + (1.0 - r0) * (x * x4);
}
