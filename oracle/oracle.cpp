/*
 * oracle.cpp — CPU restatement of Davvos11/raytracer's ray_color path loop.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED by the reference itself: its tests
 * assert nothing (src/test.rs:1-310) and it cannot be compiled here (Rust; no cargo/rustc).
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 * Arithmetic is f64, evaluated in the reference's order; build with -ffp-contract=off so no FMA is
 * formed.  The only deliberate departure is the random source: the reference draws from an OS-seeded
 * thread_rng() (src/utils/rtweekend.rs:166-169), which no test can pin; the oracle draws the same
 * *distributions* from a counter-based Philox4x32-10 stream shared bit-for-bit with the HIP kernels.
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

const double INF = std::numeric_limits<double>::infinity();

/* ---------------------------------------------------------------- Vec3: src/value/vec3.rs */
struct V3 { double x, y, z; };
inline V3 v3(double a, double b, double c) { V3 r = {a, b, c}; return r; }
inline double idx(const V3& v, unsigned a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }   /* vec3.rs:122-128 */
inline void set_idx(V3& v, unsigned a, double t) { if (a == 0) v.x = t; else if (a == 1) v.y = t; else v.z = t; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }               /* :164-169 */
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }               /* :171-177 */
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }                                    /* :114-120 */
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }               /* :179-185 */
inline V3 operator*(V3 a, double t) { return v3(a.x * t, a.y * t, a.z * t); }                 /* :187-193 */
inline V3 operator*(double t, V3 a) { return a * t; }                                         /* :195-201 */
inline V3 operator/(V3 a, double t) { return (1.0 / t) * a; }                                 /* :203-209 multiply by reciprocal */
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                   /* :97-99 */
inline double length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }              /* :88-90 */
inline double length(V3 a) { return std::sqrt(length_squared(a)); }                            /* :84-86 */
inline V3 unit(V3 a) { return a / length(a); }                                                /* :16-18 */
inline V3 cross(V3 a, V3 b) {                                                                 /* :101-107 */
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline bool near_zero(V3 a) {                                                                 /* :92-95 */
    const double s = 1e-8;
    return std::fabs(a.x) < s && std::fabs(a.y) < s && std::fabs(a.z) < s;
}
inline V3 reflect(V3 v, V3 n) { return v - 2.0 * dot(v, n) * n; }                             /* :53-55 ((2*dot)*n) */
inline V3 refract(V3 uv, V3 n, double etai_over_etat) {                                       /* :57-62 */
    double cos_theta = std::fmin(dot(-uv, n), 1.0);
    V3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
    V3 r_out_parallel = -std::sqrt(std::fabs(1.0 - length_squared(r_out_perp))) * n;
    return r_out_perp + r_out_parallel;
}

/* ---------------------------------------------------------------- Ray / Interval */
struct Ray { V3 orig, dir; };                                                                  /* ray.rs:4-7 */
inline V3 at(const Ray& r, double t) { return r.orig + (t * r.dir); }                          /* ray.rs:17-19 */
struct Interval { double min, max; };                                                          /* interval.rs:2-5 */
inline bool surrounds(Interval i, double x) { return i.min < x && x < i.max; }                 /* :32-34 */
inline double clampi(Interval i, double x) { return x < i.min ? i.min : (x > i.max ? i.max : x); } /* :36-44 */

/* ---------------------------------------------------------------- counter-based RNG (replaces rtweekend.rs:166-169) */
inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        if (r) { k0 += W0; k1 += W1; }
        uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
}
struct Rng {
    uint64_t seed; uint32_t path_lo, path_hi, stream, k; uint32_t blk[4]; bool have;
    Rng(uint64_t s, uint32_t lo, uint32_t hi, uint32_t st) : seed(s), path_lo(lo), path_hi(hi), stream(st), k(0), have(false) {}
    /* rand 0.8 Standard f64: 53 random bits * 2^-53 in [0,1) */
    double next() {
        if (!(k & 1) || !have) {
            blk[0] = path_lo; blk[1] = path_hi; blk[2] = stream; blk[3] = k >> 1;
            philox4x32_10(blk, (uint32_t)seed, (uint32_t)(seed >> 32));
            have = true;
        }
        uint64_t u = (k & 1) ? (((uint64_t)blk[3] << 32) | blk[2]) : (((uint64_t)blk[1] << 32) | blk[0]);
        ++k;
        return (double)(u >> 11) * (1.0 / 9007199254740992.0);
    }
    double range(double lo, double hi) { return lo + (hi - lo) * next(); }                     /* rtweekend.rs:171-173 */
};
inline V3 random_in_unit_disk(Rng& g) {                                                        /* vec3.rs:20-31 */
    for (;;) {
        double a = g.range(-1.0, 1.0);
        double b = g.range(-1.0, 1.0);
        V3 p = v3(a, b, 0.0);
        if (length_squared(p) < 1.0) return p;
    }
}
inline V3 random_unit(Rng& g) {                                                                /* vec3.rs:33-41, :68-74 */
    for (;;) {
        double a = g.range(-1.0, 1.0);
        double b = g.range(-1.0, 1.0);
        double c = g.range(-1.0, 1.0);
        V3 p = v3(a, b, c);
        double lsq = length_squared(p);
        if (1e-160 < lsq && lsq <= 1.0) return p / std::sqrt(lsq);
    }
}

/* ---------------------------------------------------------------- HitRecord: src/hittable.rs:13-30 */
struct HitRecord {
    V3 p; V3 normal; int mat; double t; bool front_face;
    /* bookkeeping that is not in the reference: which object was accepted last, and the t at which
     * rec.p was last written (rec.p == at(tp) bit-for-bit whenever prim >= 0 or a triangle reached :59) */
    int32_t prim; double tp;
    bool hits_aabb_edge;                                                                       /* hittable.rs:20 */
    HitRecord() : p(v3(0, 0, 0)), normal(v3(0, 0, 0)), mat(-1), t(0.0), front_face(false), prim(-1), tp(0.0), hits_aabb_edge(false) {}
};
inline void set_face_normal(HitRecord& rec, const Ray& r, V3 outward) {                        /* hittable.rs:26-29 */
    rec.front_face = dot(r.dir, outward) < 0.0;
    rec.normal = rec.front_face ? outward : -outward;
}

/* ---------------------------------------------------------------- AABB: src/acceleration/aabb.rs */
struct AABB { V3 min, max; };                                                                  /* :12-15 (Default = zeros) */
inline AABB aabb_default() { AABB b = {v3(0, 0, 0), v3(0, 0, 0)}; return b; }
struct Exit { uint32_t axis; int direction; };                                                 /* :149-153 */
inline bool aabb_is_edge(const AABB& b, V3 point) {                                            /* :90-97 */
    int count = 0;
    for (unsigned axis = 0; axis < 3; ++axis) {
        double p = idx(point, axis), mn = idx(b.min, axis), mx = idx(b.max, axis);
        if ((p <= mn + 0.01 && p >= mn - 0.01) || (p <= mx + 0.01 && p >= mx - 0.01)) ++count;
    }
    return count >= 2;
}
/* :28-62.  Returns false for None.  `draw_boxes` is Options.draw_boxes (rtweekend.rs:92): sticky edge flag on rec (:57-60). */
inline bool aabb_hit(const AABB& b, const Ray& ray, Interval ray_t, Interval& out_t, Exit& exit_, HitRecord& rec, bool draw_boxes) {
    exit_.axis = 0; exit_.direction = 0;
    for (unsigned axis = 0; axis < 3; ++axis) {
        double amin = idx(b.min, axis), amax = idx(b.max, axis);
        double ad_inverse = 1.0 / idx(ray.dir, axis);
        double t0 = (amin - idx(ray.orig, axis)) * ad_inverse;
        double t1 = (amax - idx(ray.orig, axis)) * ad_inverse;
        if (t0 < t1) {
            if (t0 > ray_t.min) ray_t.min = t0;
            if (t1 < ray_t.max) { ray_t.max = t1; exit_.axis = axis; exit_.direction = 1; }
        } else {
            if (t1 > ray_t.min) ray_t.min = t1;
            if (t0 < ray_t.max) { ray_t.max = t0; exit_.axis = axis; exit_.direction = -1; }
        }
        if (ray_t.max < ray_t.min) return false;
    }
    V3 hit_point = ray.orig + ray_t.min * ray.dir;                                            /* :57 */
    if (draw_boxes && aabb_is_edge(b, hit_point)) rec.hits_aabb_edge = true;
    out_t = ray_t;
    return true;
}
inline bool aabb_enter_and_exit(const AABB& b, const Ray& ray, Interval ray_t, double& tmin, double& tmax) { /* :65-88 */
    for (unsigned axis = 0; axis < 3; ++axis) {
        double amin = idx(b.min, axis), amax = idx(b.max, axis);
        double ad_inverse = 1.0 / idx(ray.dir, axis);
        double t0 = (amin - idx(ray.orig, axis)) * ad_inverse;
        double t1 = (amax - idx(ray.orig, axis)) * ad_inverse;
        if (t0 < t1) {
            if (t0 > ray_t.min) ray_t.min = t0;
            if (t1 < ray_t.max) ray_t.max = t1;
        } else {
            if (t1 > ray_t.min) ray_t.min = t1;
            if (t0 < ray_t.max) ray_t.max = t0;
        }
        if (ray_t.max <= ray_t.min) return false;
    }
    tmin = ray_t.min; tmax = ray_t.max;
    return true;
}
inline double surface_area(const AABB& b) {                                                    /* :104-110 */
    return 2.0 * ((b.max.x - b.min.x) * (b.max.y - b.min.y)
                  + (b.max.x - b.min.x) * (b.max.z - b.min.z)
                  + (b.max.y - b.min.y) * (b.max.z - b.min.z));
}
inline bool aabb_inside(const AABB& a, const AABB& o) {                                        /* :113-117 */
    return a.max.x >= o.min.x && a.min.x <= o.max.x
        && a.max.y >= o.min.y && a.min.y <= o.max.y
        && a.max.z >= o.min.z && a.min.z <= o.max.z;
}
inline bool point_inside(const AABB& a, V3 p) {                                                /* :119-123 */
    return a.min.x <= p.x && a.max.x >= p.x && a.min.y <= p.y && a.max.y >= p.y && a.min.z <= p.z && a.max.z >= p.z;
}
inline AABB aabb_union(const AABB& a, const AABB& b) {                                         /* :134-147 */
    AABB r;
    r.min = v3(std::fmin(a.min.x, b.min.x), std::fmin(a.min.y, b.min.y), std::fmin(a.min.z, b.min.z));
    r.max = v3(std::fmax(a.max.x, b.max.x), std::fmax(a.max.y, b.max.y), std::fmax(a.max.z, b.max.z));
    return r;
}

/* ---------------------------------------------------------------- scene */
struct Counters : orc_counters {
    Counters() { std::memset(static_cast<orc_counters*>(this), 0, sizeof(orc_counters)); }
    void add(const orc_counters& o) {
        primary_rays += o.primary_rays; scatter_rays += o.scatter_rays; intersection_checks += o.intersection_checks;
        traversal_steps += o.traversal_steps; overlapping_aabb += o.overlapping_aabb;
        gridbox_intersection_checks += o.gridbox_intersection_checks; box_tests += o.box_tests; prim_tests += o.prim_tests;
    }
};

struct BvhNode { AABB aabb; bool is_leaf; size_t left, right, first, count; };                /* bvh.rs:35-42 */
inline BvhNode node_default() { BvhNode n = {aabb_default(), true, 0, 0, 0, 0}; return n; }   /* bvh.rs:44-55 */

struct GridBox { AABB aabb; std::vector<uint32_t> objects; };                                  /* grid.rs:253-257 */

} // namespace

struct orc_scene {
    std::vector<orc_prim> prims;
    std::vector<V3> centroids;
    std::vector<AABB> prim_aabb;             /* Hittable::to_aabb, cached (pure function of the primitive) */
    std::vector<orc_material> mats;
    /* Bvh (bvh.rs:12-16) */
    bool have_bvh; int bvh_mode;
    std::vector<uint32_t> bvh_objects;       /* Bvh.objects as indices into prims */
    std::vector<BvhNode> nodes; size_t node_pointer;
    /* Grid (grid.rs:11-18) */
    bool have_grid;
    std::vector<GridBox> boxes; V3 box_size, origin, total_size; AABB grid_aabb;
    bool draw_boxes;                          /* Options.draw_boxes rtweekend.rs:92,128 */
    orc_scene() : have_bvh(false), bvh_mode(0), node_pointer(0), have_grid(false), draw_boxes(false) {}
};

namespace {

/* ---------------------------------------------------------------- primitives */
inline bool sphere_hit(const orc_prim& s, int32_t index, const Ray& r, Interval ray_t, HitRecord& rec) { /* sphere.rs:27-56 */
    V3 center = v3(s.v[0], s.v[1], s.v[2]);
    double radius = s.v[3];
    V3 oc = center - r.orig;
    double a = length_squared(r.dir);
    double h = dot(r.dir, oc);
    double c = length_squared(oc) - radius * radius;
    double discriminant = h * h - a * c;
    if (discriminant < 0.0) return false;
    double sqrt_discriminant = std::sqrt(discriminant);
    double root = (h - sqrt_discriminant) / a;
    if (!surrounds(ray_t, root)) {
        root = (h + sqrt_discriminant) / a;
        if (!surrounds(ray_t, root)) return false;
    }
    rec.t = root;
    rec.p = at(r, rec.t); rec.tp = rec.t;
    V3 outward_normal = (rec.p - center) / radius;
    set_face_normal(rec, r, outward_normal);
    rec.mat = (int)s.material; rec.prim = index;
    return true;
}
inline bool triangle_hit(const orc_prim& tr, int32_t index, const Ray& r, Interval ray_t, HitRecord& rec) { /* triangle.rs:37-81 */
    V3 v0 = v3(tr.v[0], tr.v[1], tr.v[2]), v1 = v3(tr.v[3], tr.v[4], tr.v[5]), v2 = v3(tr.v[6], tr.v[7], tr.v[8]);
    V3 v0v1 = v1 - v0;
    V3 v0v2 = v2 - v0;
    V3 n = cross(v0v1, v0v2);
    double n_dot_dir = dot(n, r.dir);
    if (!surrounds(ray_t, n_dot_dir)) return false;          /* :44-47: the t-interval is compared against n.d */
    double d = -dot(n, v0);
    rec.t = -(dot(n, r.orig) + d) / n_dot_dir;               /* :52: written before acceptance */
    if (rec.t < 0.0) return false;
    rec.p = at(r, rec.t); rec.tp = rec.t;                    /* :59: also before acceptance */
    V3 v0p = rec.p - v0;
    if (dot(n, cross(v0v1, v0p)) <= 0.0) return false;
    V3 v1v2 = v2 - v1;
    V3 v1p = rec.p - v1;
    if (dot(n, cross(v1v2, v1p)) <= 0.0) return false;
    V3 v2v0 = v0 - v2;
    V3 v2p = rec.p - v2;
    if (dot(n, cross(v2v0, v2p)) <= 0.0) return false;
    set_face_normal(rec, r, n);                               /* :77: non-unit n */
    rec.mat = (int)tr.material; rec.prim = index;
    return true;
}
inline bool object_hit(const orc_scene& sc, uint32_t i, const Ray& r, Interval ray_t, HitRecord& rec) {
    const orc_prim& p = sc.prims[i];
    return p.type == ORC_SPHERE ? sphere_hit(p, (int32_t)i, r, ray_t, rec) : triangle_hit(p, (int32_t)i, r, ray_t, rec);
}
inline AABB object_aabb(const orc_prim& p) {
    if (p.type == ORC_SPHERE) {                                                                 /* sphere.rs:58-67 */
        double r = p.v[3];
        AABB b = {v3(p.v[0] - r, p.v[1] - r, p.v[2] - r), v3(p.v[0] + r, p.v[1] + r, p.v[2] + r)};
        return b;
    }
    AABB b;                                                                                     /* triangle.rs:83-92 */
    b.min = v3(std::fmin(std::fmin(p.v[0], p.v[3]), p.v[6]), std::fmin(std::fmin(p.v[1], p.v[4]), p.v[7]), std::fmin(std::fmin(p.v[2], p.v[5]), p.v[8]));
    b.max = v3(std::fmax(std::fmax(p.v[0], p.v[3]), p.v[6]), std::fmax(std::fmax(p.v[1], p.v[4]), p.v[7]), std::fmax(std::fmax(p.v[2], p.v[5]), p.v[8]));
    return b;
}
/* hittable_list.rs:123-133 over a slice of object indices */
inline AABB objects_to_aabb(const orc_scene& sc, const uint32_t* objs, size_t count) {
    if (count == 0) return aabb_default();
    AABB acc = sc.prim_aabb[objs[0]];
    for (size_t i = 1; i < count; ++i) acc = aabb_union(acc, sc.prim_aabb[objs[i]]);
    return acc;
}

/* ---------------------------------------------------------------- BVH build: bvh.rs:19-180 */
inline bool total_less(double a, double b) {                  /* f64::total_cmp(a,b) == Less */
    int64_t x, y; std::memcpy(&x, &a, 8); std::memcpy(&y, &b, 8);
    x ^= (int64_t)(((uint64_t)(x >> 63)) >> 1); y ^= (int64_t)(((uint64_t)(y >> 63)) >> 1);
    return x < y;
}
struct Builder {
    orc_scene& sc; int mode;
    Builder(orc_scene& s, int m) : sc(s), mode(m) {}
    void sort_axis(uint32_t* objs, size_t n, unsigned axis) { /* slice::sort_by is stable */
        std::stable_sort(objs, objs + n, [&](uint32_t a, uint32_t b) { return total_less(idx(sc.centroids[a], axis), idx(sc.centroids[b], axis)); });
    }
    BvhNode new_leaf(size_t first, size_t offset, size_t count, const uint32_t* objs) { /* bvh.rs:61-64 */
        BvhNode n = node_default();
        n.aabb = objects_to_aabb(sc, objs + first, count);
        n.is_leaf = true; n.first = offset + first; n.count = count; n.left = 0; n.right = 0;
        return n;
    }
    bool check_split_sah(const BvhNode& self, size_t split, const uint32_t* objs, double current, double& h, BvhNode& l, BvhNode& r) { /* :148-160 */
        l = new_leaf(0, self.first, split, objs);
        r = new_leaf(split, self.first, self.count - split, objs);
        double lh = surface_area(l.aabb) * (double)split;
        double rh = surface_area(r.aabb) * (double)(self.count - split);
        h = lh + rh;
        return h < current;
    }
    bool get_split(const BvhNode& self, uint32_t* objs, size_t len, BvhNode& bl, BvhNode& br) { /* :86-142 */
        if (mode == ORC_BVH_NAIVE) {
            if (len < 3) return false;
            sort_axis(objs, len, 0);
            size_t split = self.count / 2;
            bl = new_leaf(0, self.first, split, objs);
            br = new_leaf(split, self.first, self.count - split, objs);
            return true;
        }
        double best_heuristic = surface_area(self.aabb) * (double)self.count;
        bool have = false; unsigned best_axis = 0;
        std::vector<uint32_t> axis_objects(len);
        for (unsigned axis = 0; axis < 3; ++axis) {
            std::copy(objs, objs + len, axis_objects.begin());
            sort_axis(axis_objects.data(), len, axis);
            double h; BvhNode l, r;
            if (mode == ORC_BVH_SAH_PLANE) {
                size_t split = self.count / 2;
                if (check_split_sah(self, split, axis_objects.data(), best_heuristic, h, l, r) && h < best_heuristic) {
                    best_heuristic = h; bl = l; br = r; have = true; best_axis = axis;
                }
            } else {
                for (size_t split = 1; split < len; ++split) {
                    if (check_split_sah(self, split, axis_objects.data(), best_heuristic, h, l, r) && h < best_heuristic) {
                        best_heuristic = h; bl = l; br = r; have = true; best_axis = axis;
                    }
                }
            }
        }
        sort_axis(objs, len, best_axis);                       /* :136-140: even when no split is taken */
        return have;
    }
    void sub_divide(BvhNode& self) {                           /* :162-180 */
        BvhNode l, r;
        if (get_split(self, sc.bvh_objects.data() + self.first, self.count, l, r)) {
            self.left = sc.node_pointer + 1;
            self.right = sc.node_pointer + 2;
            sc.node_pointer += 2;
            sub_divide(l); sc.nodes[self.left] = l;
            sub_divide(r); sc.nodes[self.right] = r;
            self.is_leaf = false;
        }
    }
};

/* ---------------------------------------------------------------- BVH build, second form (BvhSahPosition only)
 * The literal get_split above re-folds both sides of every candidate split (bvh.rs:148-160), O(len^2) box unions per node,
 * which does not end for meshes much beyond 10^4 triangles.  This form produces the same nodes from the same definitions in
 * O(len log len) per node; tests/test_oracle_sweep.py holds it to byte equality with the literal form on every scene the
 * literal form finishes, and the larger meshes' trees (res3, the 764 704-triangle bench mesh) come from here.
 *
 * Why it is the same tree, not merely a similar one:
 *  - candidate `split` on an axis costs SA(fold(sorted[..split])) * split + SA(fold(sorted[split..])) * (count - split), where
 *    fold chains AABB::add = per-coordinate f64::min / f64::max (hittable_list.rs:123-133, aabb.rs:134-147);
 *  - before[s] below IS the reference's left fold (same chain, same order), bit for bit;
 *  - after[s] chains the same operands from the other end.  min/max of non-NaN doubles return one of their operands and the
 *    value does not depend on grouping; only the sign of a zero can (min(-0, +0)).  A zero of either sign gives the same
 *    surface area up to the sign of a zero product, so `h < best` (bvh.rs:127,155) decides identically;
 *  - nothing from after[] is stored: the two children are rebuilt with the literal new_leaf (left-to-right fold) on the
 *    node's final order, which is what bvh.rs:152-153 built from axis_objects for the winning (axis, split). */
struct SweepBuilder {
    orc_scene& sc; Builder lit;
    std::vector<uint32_t> by_axis;
    std::vector<AABB> before, after;
    explicit SweepBuilder(orc_scene& s) : sc(s), lit(s, ORC_BVH_SAH_POSITION) {}

    bool get_split(const BvhNode& self, uint32_t* objs, size_t len, BvhNode& bl, BvhNode& br) {
        double best_heuristic = surface_area(self.aabb) * (double)self.count;          /* bvh.rs:101 */
        bool have = false; unsigned best_axis = 0; size_t best_split = 0;
        if (len > 1) {
            by_axis.resize(len); before.resize(len); after.resize(len + 1);
            for (unsigned axis = 0; axis < 3; ++axis) {                                /* :104 */
                std::copy(objs, objs + len, by_axis.begin());                          /* :105-106 */
                lit.sort_axis(by_axis.data(), len, axis);                              /* :107-112 */
                before[1] = sc.prim_aabb[by_axis[0]];
                for (size_t s = 2; s < len; ++s) before[s] = aabb_union(before[s - 1], sc.prim_aabb[by_axis[s - 1]]);
                after[len - 1] = sc.prim_aabb[by_axis[len - 1]];
                for (size_t s = len - 1; s-- > 1;) after[s] = aabb_union(sc.prim_aabb[by_axis[s]], after[s + 1]);
                for (size_t split = 1; split < len; ++split) {                         /* :125 */
                    double lh = surface_area(before[split]) * (double)split;           /* :152-154 */
                    double rh = surface_area(after[split]) * (double)(self.count - split);
                    double h = lh + rh;
                    if (h < best_heuristic) { best_heuristic = h; have = true; best_axis = axis; best_split = split; }   /* :155,127-131 */
                }
            }
        }
        lit.sort_axis(objs, len, best_axis);                                           /* :136-140 */
        if (have) {
            bl = lit.new_leaf(0, self.first, best_split, objs);
            br = lit.new_leaf(best_split, self.first, self.count - best_split, objs);
        }
        return have;
    }
    void sub_divide(BvhNode& self) {                                                   /* :162-180 */
        BvhNode l, r;
        if (get_split(self, sc.bvh_objects.data() + self.first, self.count, l, r)) {
            self.left = sc.node_pointer + 1;
            self.right = sc.node_pointer + 2;
            sc.node_pointer += 2;
            sub_divide(l); sc.nodes[self.left] = l;
            sub_divide(r); sc.nodes[self.right] = r;
            self.is_leaf = false;
        }
    }
};

/* ---------------------------------------------------------------- traversal */
struct Trace {
    const orc_scene& sc; Counters& data;
    Trace(const orc_scene& s, Counters& d) : sc(s), data(d) {}

    bool node_hit_aabb(const BvhNode& n, const Ray& r, Interval ray_t, HitRecord& rec, double& dist) { /* bvh.rs:182-185 */
        data.intersection_checks++; data.box_tests++;
        Interval t; Exit e;
        if (!aabb_hit(n.aabb, r, ray_t, t, e, rec, sc.draw_boxes)) return false;
        dist = t.min;
        return true;
    }
    bool node_hit(const BvhNode& n, const Ray& r, Interval ray_t, HitRecord& rec) { /* bvh.rs:189-233 */
        data.traversal_steps++;
        bool hit_anything = false;
        double closest_so_far = ray_t.max;
        if (n.is_leaf) {
            for (size_t k = n.first; k < n.first + n.count; ++k) {
                data.intersection_checks++; data.prim_tests++;
                Interval it = {ray_t.min, closest_so_far};
                if (object_hit(sc, sc.bvh_objects[k], r, it, rec)) { hit_anything = true; closest_so_far = rec.t; }
            }
        } else {
            const BvhNode& left = sc.nodes[n.left];
            double ld = 0, rd = 0;
            bool lh = node_hit_aabb(left, r, ray_t, rec, ld);
            const BvhNode& right = sc.nodes[n.right];
            bool rh = node_hit_aabb(right, r, ray_t, rec, rd);
            if (lh && rh) {
                const BvhNode& close = (ld >= rd) ? right : left;
                const BvhNode& far_ = (ld >= rd) ? left : right;
                double far_distance = std::fmax(ld, rd);
                if (node_hit(close, r, ray_t, rec)) {
                    hit_anything = true;
                    closest_so_far = rec.t;
                    if (far_distance < closest_so_far) {
                        data.overlapping_aabb++;
                        Interval it = {ray_t.min, closest_so_far};
                        if (node_hit(far_, r, it, rec)) hit_anything = true;
                    }
                } else if (node_hit(far_, r, ray_t, rec)) {
                    hit_anything = true;
                }
            } else if ((lh && node_hit(left, r, ray_t, rec)) || (rh && node_hit(right, r, ray_t, rec))) {
                hit_anything = true;
            }
        }
        return hit_anything;
    }

    /* ---- grid: grid.rs */
    static size_t sat_usize(double x) {                         /* Rust `as usize` saturates, NaN -> 0 */
        if (!(x > 0.0)) return 0;
        if (x >= 18446744073709551616.0) return std::numeric_limits<size_t>::max();
        return (size_t)x;
    }
    static int32_t sat_i32(double x) {                          /* Rust `as i32` */
        if (x != x) return 0;
        if (x <= -2147483648.0) return std::numeric_limits<int32_t>::min();
        if (x >= 2147483647.0) return std::numeric_limits<int32_t>::max();
        return (int32_t)x;
    }
    static size_t get_index(V3 point, V3 box_size, V3 total_size) { /* grid.rs:64-73 */
        size_t x = sat_usize(point.x / box_size.x);
        size_t y = sat_usize(point.y / box_size.y);
        size_t z = sat_usize(point.z / box_size.z);
        double size_x = total_size.x / box_size.x;
        double size_y = total_size.y / box_size.y;
        return sat_usize((double)x + (double)y * size_x + (double)z * size_x * size_y);
    }
    const GridBox* get_grid_box_from_point(V3 point) {         /* grid.rs:195-206; i32 subtraction wraps (release build) */
        auto q = [](double p, double o, double b) {
            int32_t d = (int32_t)((uint32_t)sat_i32(std::floor(p)) - (uint32_t)sat_i32(o));
            return std::floor((double)d / b) * b;
        };
        double x = q(point.x, sc.origin.x, sc.box_size.x);
        double y = q(point.y, sc.origin.y, sc.box_size.y);
        double z = q(point.z, sc.origin.z, sc.box_size.z);
        size_t index = get_index(v3(std::floor(x), std::floor(y), std::floor(z)), sc.box_size, sc.total_size);
        if (index >= sc.boxes.size()) return nullptr;
        return &sc.boxes[index];
    }
    const GridBox* get_box_enter(const Ray& ray) {             /* grid.rs:119-129 */
        if (const GridBox* b = get_grid_box_from_point(ray.orig)) return b;
        double tmin, tmax; Interval it = {0.001, INF};
        if (aabb_enter_and_exit(sc.grid_aabb, ray, it, tmin, tmax)) {
            V3 point = ray.orig + ray.dir * tmin;              /* grid.rs:132-136 */
            return get_grid_box_from_point(point);
        }
        return nullptr;
    }
    bool gridbox_hit(const GridBox& gb, const Ray& ray, Interval ray_t, HitRecord& rec) { /* grid.rs:281-294 */
        data.gridbox_intersection_checks++;
        bool hit_anything = false;
        double closest_so_far = ray_t.max;
        for (uint32_t i : gb.objects) {
            data.intersection_checks++; data.prim_tests++;
            Interval it = {ray_t.min, closest_so_far};
            if (object_hit(sc, i, ray, it, rec)) { hit_anything = true; closest_so_far = rec.t; }
        }
        return hit_anything;
    }
    bool grid_traverse(const GridBox* gb, const Ray& ray, Interval ray_t, HitRecord& rec) { /* grid.rs:151-183, tail recursion as a loop */
        for (uint32_t depth = 0;; ++depth) {
            if (depth > (uint32_t)sat_i32(sc.total_size.z)) return false;   /* `as u32` of 200.0 */
            data.traversal_steps++;
            Interval universe = {-INF, INF}, t_hit; Exit ex;
            if (!aabb_hit(gb->aabb, ray, universe, t_hit, ex, rec, sc.draw_boxes)) return false;   /* grid.rs:155 */
            V3 step = v3(0, 0, 0);                              /* aabb.rs:161-165 */
            set_idx(step, ex.axis, (double)ex.direction * idx(sc.box_size, ex.axis));
            HitRecord rec_copy = rec;
            if (gridbox_hit(*gb, ray, ray_t, rec_copy)) {
                rec = rec_copy;
                if (rec.t <= t_hit.max) return true;
            }
            V3 xyz = gb->aabb.min;
            data.intersection_checks++; data.box_tests++;
            V3 next_xyz = xyz + step;
            if (!point_inside(sc.grid_aabb, next_xyz)) return false;        /* grid.rs:186-192 */
            size_t next_idx = get_index(next_xyz - sc.origin, sc.box_size, sc.total_size); /* grid.rs:76-83 */
            if (next_idx >= sc.boxes.size()) return false;
            gb = &sc.boxes[next_idx];
        }
    }

    /* HittableList::hit hittable_list.rs:65-102 */
    bool world_hit(int algorithm, const Ray& r, Interval ray_t, HitRecord& rec) {
        if (algorithm == ORC_ALGO_NAIVE) {
            bool hit_anything = false;
            double closest_so_far = ray_t.max;
            for (uint32_t i = 0; i < sc.prims.size(); ++i) {
                data.intersection_checks++; data.prim_tests++;
                Interval it = {ray_t.min, closest_so_far};
                if (object_hit(sc, i, r, it, rec)) { hit_anything = true; closest_so_far = rec.t; }
            }
            return hit_anything;
        } else if (algorithm == ORC_ALGO_BVH) {
            const BvhNode& root = sc.nodes[0];
            double d;
            bool root_hit = node_hit_aabb(root, r, ray_t, rec, d);
            return root_hit && node_hit(root, r, ray_t, rec);
        } else {
            const GridBox* initial = get_box_enter(r);          /* grid.rs:139-147 */
            if (initial) return grid_traverse(initial, r, ray_t, rec);
            return false;
        }
    }

    /* ---- materials: src/value/material.rs */
    bool scatter(const orc_material& m, const Ray& r_in, const HitRecord& rec, V3& attenuation, Ray& scattered, Rng& g) {
        if (m.type == ORC_LAMBERTIAN) {                         /* :35-46 */
            V3 scatter_direction = rec.normal + random_unit(g);
            if (near_zero(scatter_direction)) scatter_direction = rec.normal;
            scattered.orig = rec.p; scattered.dir = scatter_direction;
            attenuation = v3(m.albedo[0], m.albedo[1], m.albedo[2]);
            return true;
        } else if (m.type == ORC_METAL) {                       /* :67-74 */
            V3 reflected = reflect(r_in.dir, rec.normal);
            reflected = unit(reflected) + (m.param * random_unit(g));
            scattered.orig = rec.p; scattered.dir = reflected;
            attenuation = v3(m.albedo[0], m.albedo[1], m.albedo[2]);
            return dot(scattered.dir, rec.normal) > 0.0;
        } else {                                                /* :94-113 */
            attenuation = v3(1.0, 1.0, 1.0);
            double ri = rec.front_face ? 1.0 / m.param : m.param;
            V3 unit_direction = unit(r_in.dir);
            double cos_theta = std::fmin(dot(-unit_direction, rec.normal), 1.0);
            double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
            V3 direction;
            if (ri * sin_theta > 1.0 || reflectance(cos_theta, ri) > g.next())
                direction = reflect(unit_direction, rec.normal);
            else
                direction = refract(unit_direction, rec.normal, ri);
            scattered.orig = rec.p; scattered.dir = direction;
            return true;
        }
    }
    static double reflectance(double cosine, double refraction_index) { /* :120-124; powi(5) = x*((x*x)*(x*x)) */
        double r0 = (1.0 - refraction_index) / (1.0 + refraction_index);
        r0 = r0 * r0;
        double x = 1.0 - cosine;
        double x2 = x * x;
        double x4 = x2 * x2;
        return r0 + (1.0 - r0) * (x * x4);
    }

    /* ray_color camera.rs:153-186 */
    V3 ray_color(const Ray& r, uint32_t depth, uint32_t max_depth, int algorithm, uint64_t seed, uint32_t path_lo, uint32_t path_hi) {
        if (depth == 0) return v3(0, 0, 0);
        HitRecord rec;
        Interval it = {0.001, INF};
        if (world_hit(algorithm, r, it, rec)) {
            if (rec.hits_aabb_edge) return v3(1.0, 0.0, 0.0);      /* :162-164 Color::red() vec3.rs:109 */
            Ray scattered = {v3(0, 0, 0), v3(0, 0, 0)};
            V3 attenuation = v3(0, 0, 0);
            if (rec.mat >= 0) {
                Rng g(seed, path_lo, path_hi, 1u + (max_depth - depth));   /* stream = 1 + bounce index */
                if (scatter(sc.mats[rec.mat], r, rec, attenuation, scattered, g)) {
                    data.scatter_rays++;
                    return attenuation * ray_color(scattered, depth - 1, max_depth, algorithm, seed, path_lo, path_hi);
                }
            }
            return v3(0, 0, 0);
        } else {
            if (rec.hits_aabb_edge) return v3(1.0, 0.0, 0.0);      /* :178-180 */
            V3 unit_direction = unit(r.dir);
            double a = 0.5 * (unit_direction.y + 1.0);
            return (1.0 - a) * v3(1.0, 1.0, 1.0) + a * v3(0.5, 0.7, 1.0);
        }
    }
};

/* Camera::get_ray camera.rs:132-148, sample_square :189-191.  RNG stream 0 of the path. */
inline Ray get_ray(const orc_camera& c, uint32_t i, uint32_t j, uint64_t seed, uint32_t s) {
    Rng g(seed, j * c.image_width + i, s, 0);
    double ox = g.next() - 0.5;
    double oy = g.next() - 0.5;
    V3 pixel00 = v3(c.pixel00[0], c.pixel00[1], c.pixel00[2]);
    V3 du = v3(c.du[0], c.du[1], c.du[2]), dv = v3(c.dv[0], c.dv[1], c.dv[2]);
    V3 center = v3(c.center[0], c.center[1], c.center[2]);
    V3 pixel_sample = pixel00 + (((double)i + ox) * du) + (((double)j + oy) * dv);
    V3 ray_origin;
    if (c.defocus_angle <= 0.0) {
        ray_origin = center;
    } else {
        V3 p = random_in_unit_disk(g);
        V3 ddu = v3(c.ddu[0], c.ddu[1], c.ddu[2]), ddv = v3(c.ddv[0], c.ddv[1], c.ddv[2]);
        ray_origin = center + (p.x * ddu) + (p.y * ddv);
    }
    Ray r = {ray_origin, pixel_sample - ray_origin};
    return r;
}

} // namespace

/* ================================================================ C ABI */
extern "C" {

orc_scene* orc_scene_create(const orc_prim* prims, const double* centroids, uint32_t n, const orc_material* mats, uint32_t m) {
    orc_scene* sc = new orc_scene();
    sc->prims.assign(prims, prims + n);
    sc->mats.assign(mats, mats + m);
    sc->centroids.resize(n); sc->prim_aabb.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        sc->centroids[i] = v3(centroids[3 * i], centroids[3 * i + 1], centroids[3 * i + 2]);
        sc->prim_aabb[i] = object_aabb(sc->prims[i]);
    }
    return sc;
}
void orc_scene_destroy(orc_scene* sc) { delete sc; }

int orc_build_bvh(orc_scene* sc, int mode) {                   /* Bvh::new bvh.rs:19-26 */
    size_t n = sc->prims.size();
    sc->bvh_objects.resize(n);
    for (size_t i = 0; i < n; ++i) sc->bvh_objects[i] = (uint32_t)i;
    sc->nodes.assign(n * 4, node_default());
    sc->node_pointer = 0; sc->bvh_mode = mode;
    if (n == 0) { sc->have_bvh = false; return -1; }           /* reference: nodes.first() is None -> every ray misses */
    Builder b(*sc, mode);
    BvhNode root = b.new_leaf(0, 0, n, sc->bvh_objects.data());
    b.sub_divide(root);
    sc->nodes[0] = root;
    sc->have_bvh = true;
    return 0;
}
int orc_build_bvh_sweep(orc_scene* sc) {                       /* Bvh::new bvh.rs:19-26, BvhSahPosition through SweepBuilder */
    size_t n = sc->prims.size();
    sc->bvh_objects.resize(n);
    for (size_t i = 0; i < n; ++i) sc->bvh_objects[i] = (uint32_t)i;
    sc->nodes.assign(n * 4, node_default());
    sc->node_pointer = 0; sc->bvh_mode = ORC_BVH_SAH_POSITION;
    if (n == 0) { sc->have_bvh = false; return -1; }
    SweepBuilder b(*sc);
    BvhNode root = b.lit.new_leaf(0, 0, n, sc->bvh_objects.data());
    b.sub_divide(root);
    sc->nodes[0] = root;
    sc->have_bvh = true;
    return 0;
}
int orc_set_bvh(orc_scene* sc, const orc_bvh_node* nodes, uint32_t n_nodes, const uint32_t* order, uint32_t n) {
    if (n != sc->prims.size() || n_nodes == 0) return -1;
    sc->bvh_objects.assign(order, order + n);
    sc->nodes.resize(n_nodes);
    for (uint32_t i = 0; i < n_nodes; ++i) {
        BvhNode b;
        b.aabb.min = v3(nodes[i].min[0], nodes[i].min[1], nodes[i].min[2]);
        b.aabb.max = v3(nodes[i].max[0], nodes[i].max[1], nodes[i].max[2]);
        b.is_leaf = nodes[i].is_leaf != 0; b.left = nodes[i].left; b.right = nodes[i].right; b.first = nodes[i].first; b.count = nodes[i].count;
        if (!b.is_leaf && (b.left >= n_nodes || b.right >= n_nodes)) return -2;
        if (b.is_leaf && b.first + b.count > n) return -3;
        sc->nodes[i] = b;
    }
    sc->node_pointer = n_nodes - 1; sc->have_bvh = true;
    return 0;
}
uint32_t orc_bvh_node_count(const orc_scene* sc) { return sc->have_bvh ? (uint32_t)(sc->node_pointer + 1) : 0; }
int orc_get_bvh(const orc_scene* sc, orc_bvh_node* out, uint32_t* order_out) {
    if (!sc->have_bvh) return -1;
    for (size_t i = 0; i <= sc->node_pointer; ++i) {
        const BvhNode& b = sc->nodes[i];
        orc_bvh_node o;
        o.min[0] = b.aabb.min.x; o.min[1] = b.aabb.min.y; o.min[2] = b.aabb.min.z;
        o.max[0] = b.aabb.max.x; o.max[1] = b.aabb.max.y; o.max[2] = b.aabb.max.z;
        o.is_leaf = b.is_leaf; o.left = (uint32_t)b.left; o.right = (uint32_t)b.right; o.first = (uint32_t)b.first; o.count = (uint32_t)b.count; o.pad = 0;
        out[i] = o;
    }
    std::copy(sc->bvh_objects.begin(), sc->bvh_objects.end(), order_out);
    return 0;
}

int orc_build_grid(orc_scene* sc, double s) {                  /* hittable_list.rs:40-43 -> Grid::new grid.rs:23-61 */
    if (!(s >= 1.0)) return -1;                                 /* step_by(0) panics in the reference */
    V3 box_size = v3(s, s, s), origin = v3(-100.0, -100.0, -100.0), end = v3(100.0, 100.0, 100.0), total = v3(200.0, 200.0, 200.0);
    auto try_add_all = [&](GridBox& gb) {                       /* grid.rs:267-278 */
        for (uint32_t i = 0; i < sc->prims.size(); ++i) {
            const AABB& o = sc->prim_aabb[i];
            if (aabb_inside(gb.aabb, o) || aabb_inside(o, gb.aabb)) gb.objects.push_back(i);
        }
    };
    GridBox origin_box; origin_box.aabb.min = origin; origin_box.aabb.max = origin + box_size;   /* grid.rs:261-264 */
    try_add_all(origin_box);
    size_t n_cells = Trace::sat_usize((total.x / box_size.x) * (total.y / box_size.y) * (total.z / box_size.z));
    sc->boxes.assign(n_cells, origin_box);
    int32_t sx = Trace::sat_i32(origin.x), sy = Trace::sat_i32(origin.y), sz = Trace::sat_i32(origin.z);
    int32_t ex = Trace::sat_i32(end.x), ey = Trace::sat_i32(end.y), ez = Trace::sat_i32(end.z);
    size_t stx = Trace::sat_usize(box_size.x), sty = Trace::sat_usize(box_size.y), stz = Trace::sat_usize(box_size.z);
    for (int64_t x = sx; x < ex; x += (int64_t)stx)
        for (int64_t y = sy; y < ey; y += (int64_t)sty)
            for (int64_t z = sz; z < ez; z += (int64_t)stz) {
                V3 mn = v3((double)x, (double)y, (double)z);
                V3 distance_check = mn - origin;
                if (length_squared(distance_check) <= 1.0) continue;
                GridBox gb; gb.aabb.min = mn; gb.aabb.max = mn + box_size;
                try_add_all(gb);
                int32_t fx = Trace::sat_i32((double)x - origin.x), fy = Trace::sat_i32((double)y - origin.y), fz = Trace::sat_i32((double)z - origin.z);
                size_t index = Trace::get_index(v3((double)fx, (double)fy, (double)fz), box_size, total);
                if (index >= n_cells) { sc->boxes.clear(); sc->have_grid = false; return -2; }   /* reference panics (index out of bounds) */
                sc->boxes[index] = gb;
            }
    sc->box_size = box_size; sc->origin = origin; sc->total_size = total;
    sc->grid_aabb.min = origin; sc->grid_aabb.max = end;
    sc->have_grid = true;
    return 0;
}
uint64_t orc_grid_cell_count(const orc_scene* sc) { return sc->boxes.size(); }
uint64_t orc_grid_list_len(const orc_scene* sc) { uint64_t t = 0; for (const GridBox& b : sc->boxes) t += b.objects.size(); return t; }
int orc_get_grid(const orc_scene* sc, double* cell_aabb, uint64_t* cell_offset, uint32_t* prim_list) {
    if (!sc->have_grid) return -1;
    uint64_t off = 0;
    for (size_t i = 0; i < sc->boxes.size(); ++i) {
        const GridBox& b = sc->boxes[i];
        double* a = cell_aabb + 6 * i;
        a[0] = b.aabb.min.x; a[1] = b.aabb.min.y; a[2] = b.aabb.min.z; a[3] = b.aabb.max.x; a[4] = b.aabb.max.y; a[5] = b.aabb.max.z;
        cell_offset[i] = off;
        for (uint32_t o : b.objects) prim_list[off++] = o;
    }
    cell_offset[sc->boxes.size()] = off;
    return 0;
}

void orc_set_draw_boxes(orc_scene* sc, int on) { sc->draw_boxes = on != 0; }

int orc_trace(const orc_scene* sc, int algorithm, const double* orig, const double* dir, uint64_t n, double tmin, double tmax,
              int32_t* out_prim, double* out_t, double* out_tp, uint8_t* out_hit, orc_counters* counters) {
    if (algorithm == ORC_ALGO_BVH && !sc->have_bvh) return -1;
    if (algorithm == ORC_ALGO_GRID && !sc->have_grid) return -1;
    Counters c; Trace tr(*sc, c);
    for (uint64_t i = 0; i < n; ++i) {
        Ray r = {v3(orig[3 * i], orig[3 * i + 1], orig[3 * i + 2]), v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2])};
        HitRecord rec; Interval it = {tmin, tmax};
        bool h = tr.world_hit(algorithm, r, it, rec);
        if (out_prim) out_prim[i] = rec.prim;
        if (out_t) out_t[i] = rec.t;
        if (out_tp) out_tp[i] = rec.tp;
        if (out_hit) out_hit[i] = (uint8_t)((h ? 1 : 0) | (rec.hits_aabb_edge ? 2 : 0));
    }
    if (counters) *counters = c;
    return 0;
}

void orc_camera_init(const orc_camera_in* in, orc_camera* out) { /* camera.rs:84-128 */
    uint32_t image_height = (uint32_t)Trace::sat_usize((double)in->image_width / in->aspect_ratio);
    image_height = image_height < 1 ? 1 : image_height;
    V3 look_from = v3(in->look_from[0], in->look_from[1], in->look_from[2]);
    V3 look_at = v3(in->look_at[0], in->look_at[1], in->look_at[2]);
    V3 v_up = v3(in->v_up[0], in->v_up[1], in->v_up[2]);
    V3 center = look_from;
    const double PI = 3.14159265358979323846264338327950288;
    double theta = in->vfov * PI / 180.0;                       /* rtweekend.rs:162-164 */
    double h = std::tan(theta / 2.0);
    double viewport_height = 2.0 * h * in->focus_dist;
    double viewport_width = viewport_height * ((double)in->image_width / (double)image_height);
    V3 w = unit(look_from - look_at);
    V3 u = unit(cross(v_up, w));
    V3 v = cross(w, u);
    V3 viewport_u = viewport_width * u;
    V3 viewport_v = viewport_height * -v;
    V3 du = viewport_u / (double)in->image_width;
    V3 dv = viewport_v / (double)image_height;
    V3 viewport_upper_left = center - (in->focus_dist * w) - (viewport_u / 2.0) - (viewport_v / 2.0);
    V3 pixel00 = viewport_upper_left + 0.5 * (du + dv);
    double defocus_radius = in->focus_dist * std::tan((in->defocus_angle / 2.0) * PI / 180.0);
    V3 ddu = u * defocus_radius;
    V3 ddv = v * defocus_radius;
    out->image_width = in->image_width; out->image_height = image_height;
    auto st = [](double* d, V3 a) { d[0] = a.x; d[1] = a.y; d[2] = a.z; };
    st(out->center, center); st(out->pixel00, pixel00); st(out->du, du); st(out->dv, dv); st(out->ddu, ddu); st(out->ddv, ddv);
    out->defocus_angle = in->defocus_angle;
}

void orc_generate(const orc_camera* c, uint64_t seed, uint32_t s, const uint32_t* pi, const uint32_t* pj, uint64_t n, double* orig, double* dir) {
    for (uint64_t k = 0; k < n; ++k) {
        Ray r = get_ray(*c, pi[k], pj[k], seed, s);
        orig[3 * k] = r.orig.x; orig[3 * k + 1] = r.orig.y; orig[3 * k + 2] = r.orig.z;
        dir[3 * k] = r.dir.x; dir[3 * k + 1] = r.dir.y; dir[3 * k + 2] = r.dir.z;
    }
}

int orc_render(const orc_scene* sc, const orc_camera* cam, int algorithm, uint32_t s_begin, uint32_t s_count, uint32_t max_depth,
               uint64_t seed, uint32_t row_begin, uint32_t row_end, double* out_sum, orc_counters* counters, int nthreads) {
    if (algorithm == ORC_ALGO_BVH && !sc->have_bvh) return -1;
    if (algorithm == ORC_ALGO_GRID && !sc->have_grid) return -1;
    if (row_end > cam->image_height) row_end = cam->image_height;
    if (nthreads < 1) nthreads = 1;
    std::vector<Counters> cs((size_t)nthreads);
    auto work = [&](int tid) {
        Trace tr(*sc, cs[(size_t)tid]);
        uint32_t W = cam->image_width;
        for (uint32_t j = row_begin + (uint32_t)tid; j < row_end; j += (uint32_t)nthreads)   /* camera.rs:64 */
            for (uint32_t i = 0; i < W; ++i) {                                               /* :67 */
                V3 pixel_color = v3(0, 0, 0);
                for (uint32_t s = s_begin; s < s_begin + s_count; ++s) {                     /* :69 */
                    cs[(size_t)tid].primary_rays++;
                    Ray r = get_ray(*cam, i, j, seed, s);
                    V3 c = tr.ray_color(r, max_depth, max_depth, algorithm, seed, j * W + i, s);
                    pixel_color = pixel_color + c;                                           /* += :72 */
                }
                double* o = out_sum + ((size_t)j * W + i) * 3;
                o[0] = pixel_color.x; o[1] = pixel_color.y; o[2] = pixel_color.z;
            }
    };
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    if (counters) { Counters tot; for (auto& c : cs) tot.add(c); *counters = tot; }
    return 0;
}

void orc_resolve(const double* sum, uint64_t n_pixels, uint32_t spp_total, uint8_t* rgb) { /* camera.rs:93,75 ; color.rs:6-30 */
    double scale = 1.0 / (double)spp_total;
    Interval intensity = {0.0, 0.999};
    for (uint64_t p = 0; p < n_pixels * 3; ++p) {
        double lin = sum[p] * scale;
        double g = lin > 0.0 ? std::sqrt(lin) : 0.0;
        double q = 256.0 * clampi(intensity, g);
        rgb[p] = (uint8_t)(uint32_t)q;
    }
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    for (int i = 0; i < 4; ++i) out[i] = c[i];
}

double orc_rng_draw(uint64_t seed, uint32_t path_lo, uint32_t path_hi, uint32_t stream, uint32_t k) {
    Rng g(seed, path_lo, path_hi, stream);
    double v = 0;
    for (uint32_t i = 0; i <= k; ++i) v = g.next();
    return v;
}

} // extern "C"
