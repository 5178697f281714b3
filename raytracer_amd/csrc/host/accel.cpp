// accel.cpp — host builders for the acceleration structures the device consumes.
//
// Both builders must reproduce the reference's *layout* exactly (node numbering, leaf object order,
// per-cell object order), because the reference's traversal returns the LAST accepted primitive in
// traversal order, so a differently shaped tree yields different images.  They do not
// reproduce its cost: the SAH sweep uses prefix/suffix boxes instead of re-folding both sides for every
// candidate split, and the grid bins each object into the cells its box touches instead of testing every
// cell against every object.
#include "scene.h"

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <thread>

namespace rth {
namespace {

struct Box { double mn[3], mx[3]; };

Box prim_box(const rt_prim& p) {
    Box b;
    if (p.type == RT_SPHERE) {                           // sphere.rs:58-67
        for (int c = 0; c < 3; ++c) { b.mn[c] = p.v[c] - p.v[3]; b.mx[c] = p.v[c] + p.v[3]; }
    } else {                                             // triangle.rs:83-92
        for (int c = 0; c < 3; ++c) {
            b.mn[c] = std::fmin(std::fmin(p.v[c], p.v[3 + c]), p.v[6 + c]);
            b.mx[c] = std::fmax(std::fmax(p.v[c], p.v[3 + c]), p.v[6 + c]);
        }
    }
    return b;
}
inline void grow(Box& a, const Box& b) {                 // aabb.rs:134-147
    for (int c = 0; c < 3; ++c) { a.mn[c] = std::fmin(a.mn[c], b.mn[c]); a.mx[c] = std::fmax(a.mx[c], b.mx[c]); }
}
inline double area(const Box& b) {                       // aabb.rs:104-110
    return 2.0 * ((b.mx[0] - b.mn[0]) * (b.mx[1] - b.mn[1])
                  + (b.mx[0] - b.mn[0]) * (b.mx[2] - b.mn[2])
                  + (b.mx[1] - b.mn[1]) * (b.mx[2] - b.mn[2]));
}
inline int64_t total_key(double d) {                     // f64::total_cmp ordering as a signed integer
    int64_t x; std::memcpy(&x, &d, 8);
    return x ^ (int64_t)(((uint64_t)(x >> 63)) >> 1);
}
struct Keyed { int64_t key; uint32_t obj; };

// Scratch of one get_split call chain (one per thread)
struct SplitScratch { std::vector<Keyed> sorted[3]; std::vector<Box> suffix[3]; };

struct BvhBuilder {
    const Scene& sc; BvhMode mode; Bvh& out;
    std::vector<Box> boxes;
    size_t par_min;                                      // nodes with at least this many objects are split with helper threads

    BvhBuilder(const Scene& s, BvhMode m, Bvh& o) : sc(s), mode(m), out(o), par_min(32768) {
        if (const char* e = getenv("RT_BVH_PAR_MIN")) { long v = atol(e); if (v > 0) par_min = (size_t)v; }
    }

    Box fold(const uint32_t* objs, size_t n) const {     // objects_to_aabb hittable_list.rs:123-133
        Box b; std::memset(&b, 0, sizeof b);
        if (n == 0) return b;
        b = boxes[objs[0]];
        for (size_t i = 1; i < n; ++i) grow(b, boxes[objs[i]]);
        return b;
    }
    rt_bvh_node leaf(size_t first, size_t count) const { // BvhNode::new_leaf bvh.rs:61-64
        Box b = fold(out.order.data() + first, count);
        rt_bvh_node n; std::memset(&n, 0, sizeof n);
        for (int c = 0; c < 3; ++c) { n.min[c] = b.mn[c]; n.max[c] = b.mx[c]; }
        n.is_leaf = 1; n.first = (uint32_t)first; n.count = (uint32_t)count;
        return n;
    }
    void sort_axis(const uint32_t* objs, size_t n, int axis, std::vector<Keyed>& v) const {   // slice::sort_by is stable
        v.resize(n);
        for (size_t i = 0; i < n; ++i) { v[i].obj = objs[i]; v[i].key = total_key(sc.centroids[3 * (size_t)objs[i] + axis]); }
        std::stable_sort(v.begin(), v.end(), [](const Keyed& a, const Keyed& b) { return a.key < b.key; });
    }
    // One axis of get_split (bvh.rs:104-134): the stable sort and the best candidate on that axis, as (cost, split);
    // split 0 = no candidate below `limit`.  Strict `<` keeps the first minimum in split order (bvh.rs:127,152-156).
    void sweep_axis(const uint32_t* objs, size_t n, int axis, double limit, SplitScratch& sx, double& best, size_t& best_split) const {
        std::vector<Keyed>& v = sx.sorted[axis];
        sort_axis(objs, n, axis, v);
        best = limit; best_split = 0;
        if (mode == BVH_SAH_PLANE) {                     // :114-123
            size_t split = n / 2;
            std::vector<uint32_t> tmp(n); for (size_t i = 0; i < n; ++i) tmp[i] = v[i].obj;
            Box l = fold(tmp.data(), split), r = fold(tmp.data() + split, n - split);
            double h = area(l) * (double)split + area(r) * (double)(n - split);
            if (h < best) { best = h; best_split = split; }
        } else if (n > 1) {                              // :124-133 as a sweep with suffix boxes
            std::vector<Box>& suffix = sx.suffix[axis];
            suffix.resize(n);
            suffix[n - 1] = boxes[v[n - 1].obj];
            for (size_t i = n - 1; i-- > 0;) { suffix[i] = boxes[v[i].obj]; grow(suffix[i], suffix[i + 1]); }
            Box pre = boxes[v[0].obj];
            for (size_t split = 1; split < n; ++split) {
                if (split > 1) grow(pre, boxes[v[split - 1].obj]);
                double h = area(pre) * (double)split + area(suffix[split]) * (double)(n - split);   // :152-154
                if (h < best) { best = h; best_split = split; }
            }
        }
    }
    // BvhNode::get_split bvh.rs:86-142.  Returns the split position (0 = none); objs is reordered as the
    // reference reorders the node's slice.  The three axes are independent given the node's cost: with helpers they run
    // concurrently and are combined in axis order, each axis having to beat the best of the axes before it (:127).
    size_t get_split(const rt_bvh_node& self, uint32_t* objs, size_t n, SplitScratch& sx, bool helpers) const {
        if (mode == BVH_NAIVE) {                         // :87-99
            if (n < 3) return 0;
            sort_axis(objs, n, 0, sx.sorted[0]);
            for (size_t i = 0; i < n; ++i) objs[i] = sx.sorted[0][i].obj;
            return n / 2;
        }
        Box sb; for (int c = 0; c < 3; ++c) { sb.mn[c] = self.min[c]; sb.mx[c] = self.max[c]; }
        const double node_cost = area(sb) * (double)n;   // :101
        double best = node_cost; size_t best_split = 0; int best_axis = 0;
        if (helpers) {
            // each axis finds its own first minimum below the NODE's cost; an axis only wins if it is strictly below the
            // running best of the earlier axes — the same winner as the sequential scan, because an axis's first minimum
            // below a smaller limit is the same candidate whenever that candidate is below the limit at all
            double ab[3]; size_t as[3];
            std::thread t1([&] { sweep_axis(objs, n, 1, node_cost, sx, ab[1], as[1]); });
            std::thread t2([&] { sweep_axis(objs, n, 2, node_cost, sx, ab[2], as[2]); });
            sweep_axis(objs, n, 0, node_cost, sx, ab[0], as[0]);
            t1.join(); t2.join();
            for (int axis = 0; axis < 3; ++axis) if (as[axis] != 0 && ab[axis] < best) { best = ab[axis]; best_split = as[axis]; best_axis = axis; }
        } else {
            for (int axis = 0; axis < 3; ++axis) {
                double b; size_t sp;
                sweep_axis(objs, n, axis, best, sx, b, sp);
                if (sp != 0) { best = b; best_split = sp; best_axis = axis; }
            }
        }
        for (size_t i = 0; i < n; ++i) objs[i] = sx.sorted[best_axis][i].obj;                              // :136-140
        return best_split;
    }

    // BvhNode::sub_divide (bvh.rs:162-180) of one subtree without recursion, into `nodes` with LOCAL numbering
    // (nodes[0] = the subtree's root).  The reference numbers children at the moment the parent splits
    // (left = ptr+1, right = ptr+2) and finishes the left subtree first; a LIFO of pending nodes with the right child
    // pushed below the left reproduces that order.
    uint32_t build_seq(size_t first, size_t count, std::vector<rt_bvh_node>& nodes, SplitScratch& sx) const {
        struct Item { uint32_t node, depth; };
        nodes.clear(); nodes.push_back(leaf(first, count));
        std::vector<Item> todo; todo.push_back({0u, 1u});
        uint32_t depth = 1;
        while (!todo.empty()) {
            Item it = todo.back(); todo.pop_back();
            rt_bvh_node self = nodes[it.node];
            size_t split = get_split(self, out.order.data() + self.first, self.count, sx, false);
            if (it.depth > depth) depth = it.depth;
            if (split == 0) continue;
            uint32_t l = (uint32_t)nodes.size(), r = l + 1;
            nodes.push_back(leaf(self.first, split));
            nodes.push_back(leaf(self.first + split, self.count - split));
            nodes[it.node].is_leaf = 0; nodes[it.node].left = l; nodes[it.node].right = r;
            todo.push_back({r, it.depth + 1}); todo.push_back({l, it.depth + 1});
        }
        return depth;
    }
    // The same numbering built from independently built subtrees: with P at local 0 its children sit at 1 and 2, the rest
    // of the left subtree follows from 3 (the reference finishes it before touching the right child), then the rest of
    // the right subtree.  Inside a subtree the same rule applies recursively, so remapping indices is all a merge needs.
    uint32_t build_par(size_t first, size_t count, std::vector<rt_bvh_node>& nodes, unsigned threads) const {
        SplitScratch sx;
        if (count < par_min || threads < 2) return build_seq(first, count, nodes, sx);
        nodes.clear(); nodes.push_back(leaf(first, count));
        const size_t split = get_split(nodes[0], out.order.data() + first, count, sx, threads >= 3);
        if (split == 0) return 1;
        sx = SplitScratch();                             // the big node's scratch is not kept while the subtrees build
        std::vector<rt_bvh_node> L, R; uint32_t dl = 0, dr = 0;
        const unsigned tl = threads / 2, tr = threads - tl;
        std::thread left([&] { dl = build_par(first, split, L, tl); });
        dr = build_par(first + split, count - split, R, tr);
        left.join();
        const uint32_t nl = (uint32_t)L.size(), nr = (uint32_t)R.size();
        auto map_l = [](uint32_t i) { return i == 0 ? 1u : 3u + (i - 1); };
        auto map_r = [nl](uint32_t i) { return i == 0 ? 2u : 3u + (nl - 1) + (i - 1); };
        nodes.resize(1 + (size_t)nl + nr);
        nodes[0].is_leaf = 0; nodes[0].left = 1; nodes[0].right = 2;
        for (uint32_t i = 0; i < nl; ++i) { rt_bvh_node n = L[i]; if (!n.is_leaf) { n.left = map_l(n.left); n.right = map_l(n.right); } nodes[map_l(i)] = n; }
        for (uint32_t i = 0; i < nr; ++i) { rt_bvh_node n = R[i]; if (!n.is_leaf) { n.left = map_r(n.left); n.right = map_r(n.right); } nodes[map_r(i)] = n; }
        return 1 + (dl > dr ? dl : dr);
    }

    bool run(std::string& err) {
        const size_t n = sc.prims.size();
        if (n == 0) { err = "scene has no objects"; return false; }
        if (n > 0x3fffffffu) { err = "too many objects"; return false; }
        boxes.resize(n);
        for (size_t i = 0; i < n; ++i) boxes[i] = prim_box(sc.prims[i]);
        out.order.resize(n);
        for (size_t i = 0; i < n; ++i) out.order[i] = (uint32_t)i;
        unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 1; if (hw > 16) hw = 16;
        out.depth = build_par(0, n, out.nodes, hw);      // Bvh::new bvh.rs:22
        return true;
    }
};

} // namespace

bool build_bvh(const Scene& sc, BvhMode mode, Bvh& out, std::string& err) {
    out = Bvh();
    BvhBuilder b(sc, mode, out);
    return b.run(err);
}

// ------------------------------------------------------------------------------------------------
namespace {
size_t sat_usize(double x) { if (!(x > 0.0)) return 0; if (x >= 18446744073709551616.0) return std::numeric_limits<size_t>::max(); return (size_t)x; }
int32_t sat_i32(double x) { if (x != x) return 0; if (x <= -2147483648.0) return std::numeric_limits<int32_t>::min(); if (x >= 2147483647.0) return std::numeric_limits<int32_t>::max(); return (int32_t)x; }
size_t get_index(const double p[3], const double bs[3], const double total[3]) {     // grid.rs:64-73
    size_t x = sat_usize(p[0] / bs[0]), y = sat_usize(p[1] / bs[1]), z = sat_usize(p[2] / bs[2]);
    double size_x = total[0] / bs[0], size_y = total[1] / bs[1];
    return sat_usize((double)x + (double)y * size_x + (double)z * size_x * size_y);
}
} // namespace

bool build_grid(const Scene& sc, double s, Grid& out, std::string& err) {
    if (!(s >= 1.0)) { err = "grid size must be >= 1 (the reference's step_by panics on 0)"; return false; }
    out = Grid();
    rt_grid_desc& d = out.desc;                          // hittable_list.rs:43
    for (int c = 0; c < 3; ++c) { d.box_size[c] = s; d.origin[c] = -100.0; d.end[c] = 100.0; d.total_size[c] = 200.0; }
    const size_t n_cells = sat_usize((d.total_size[0] / s) * (d.total_size[1] / s) * (d.total_size[2] / s));   // grid.rs:27
    if (n_cells == 0) { err = "grid has no cells"; return false; }
    d.n_cells = n_cells;
    const size_t n = sc.prims.size();
    std::vector<Box> boxes(n);
    for (size_t i = 0; i < n; ++i) boxes[i] = prim_box(sc.prims[i]);

    // loop coordinates per axis (grid.rs:29-38): (start..end).step_by(box_size as usize)
    std::vector<double> coord[3];
    for (int c = 0; c < 3; ++c) {
        int64_t a = sat_i32(d.origin[c]), b = sat_i32(d.end[c]); size_t st = sat_usize(d.box_size[c]);
        for (int64_t x = a; x < b; x += (int64_t)st) coord[c].push_back((double)x);
    }
    const size_t nx = coord[0].size(), ny = coord[1].size(), nz = coord[2].size();
    // which loop cell ends up owning each table slot (later assignments overwrite earlier ones, grid.rs:55)
    const uint32_t NONE = 0xffffffffu;
    if (nx * ny * nz >= NONE) { err = "grid too fine"; return false; }
    std::vector<uint32_t> owner(n_cells, NONE);
    for (size_t ix = 0; ix < nx; ++ix) for (size_t iy = 0; iy < ny; ++iy) for (size_t iz = 0; iz < nz; ++iz) {
        double mn[3] = {coord[0][ix], coord[1][iy], coord[2][iz]};
        double dc[3] = {mn[0] - d.origin[0], mn[1] - d.origin[1], mn[2] - d.origin[2]};
        if (dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2] <= 1.0) continue;     // grid.rs:41-45
        double fin[3] = {(double)sat_i32(mn[0] - d.origin[0]), (double)sat_i32(mn[1] - d.origin[1]), (double)sat_i32(mn[2] - d.origin[2])};
        size_t index = get_index(fin, d.box_size, d.total_size);
        if (index >= n_cells) { err = "grid size does not tile the 200^3 world (the reference panics: index out of bounds)"; return false; }
        owner[index] = (uint32_t)((ix * ny + iy) * nz + iz);
    }
    // origin box (grid.rs:24-25): every unowned slot is a clone of it, AABB and list included (grid.rs:27)
    Box ob; for (int c = 0; c < 3; ++c) { ob.mn[c] = d.origin[c]; ob.mx[c] = d.origin[c] + d.box_size[c]; }
    auto overlaps = [](const Box& a, const Box& o) {     // aabb.rs:113-117 (symmetric, inclusive); grid.rs:267-271
        return a.mx[0] >= o.mn[0] && a.mn[0] <= o.mx[0] && a.mx[1] >= o.mn[1] && a.mn[1] <= o.mx[1] && a.mx[2] >= o.mn[2] && a.mn[2] <= o.mx[2];
    };
    std::vector<uint32_t> origin_list;
    for (size_t i = 0; i < n; ++i) if (overlaps(ob, boxes[i]) || overlaps(boxes[i], ob)) origin_list.push_back((uint32_t)i);

    // slot of each loop cell if it is the final owner
    std::vector<size_t> slot_of(nx * ny * nz, (size_t)-1);
    for (size_t idx = 0; idx < n_cells; ++idx) if (owner[idx] != NONE) slot_of[owner[idx]] = idx;

    // per-axis candidate cells of object i: exact 1-D predicate cell.max >= o.min && cell.min <= o.max
    auto axis_range = [&](int c, const Box& o, std::vector<uint32_t>& hits) {
        hits.clear();
        const std::vector<double>& co = coord[c];
        if (co.empty()) return;
        double st = (double)sat_usize(d.box_size[c]);
        double lo = std::floor((o.mn[c] - d.box_size[c] - co[0]) / st) - 1.0, hi = std::ceil((o.mx[c] - co[0]) / st) + 1.0;
        if (!(lo == lo) || !(hi == hi)) { lo = 0; hi = (double)co.size(); }     // NaN boxes: test everything
        size_t a = lo < 0 ? 0 : (lo > (double)co.size() ? co.size() : (size_t)lo);
        size_t b = hi < 0 ? 0 : (hi >= (double)co.size() ? co.size() : (size_t)hi + 1);
        for (size_t k = a; k < b && k < co.size(); ++k) {
            double cmin = co[k], cmax = co[k] + d.box_size[c];                 // GridBox::new grid.rs:261-264
            if (cmax >= o.mn[c] && cmin <= o.mx[c]) hits.push_back((uint32_t)k);
        }
    };
    out.cell_offset.assign(n_cells + 1, 0);
    std::vector<uint32_t> hx, hy, hz;
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<uint64_t> cursor;
        if (pass == 1) {
            for (size_t idx = 0; idx < n_cells; ++idx) if (owner[idx] == NONE) out.cell_offset[idx + 1] = origin_list.size();
            for (size_t idx = 0; idx < n_cells; ++idx) out.cell_offset[idx + 1] += out.cell_offset[idx];
            out.prim_list.assign(out.cell_offset[n_cells], 0);
            cursor.assign(out.cell_offset.begin(), out.cell_offset.end() - 1);
            for (size_t idx = 0; idx < n_cells; ++idx) if (owner[idx] == NONE)
                std::copy(origin_list.begin(), origin_list.end(), out.prim_list.begin() + (std::ptrdiff_t)out.cell_offset[idx]);
        }
        for (size_t i = 0; i < n; ++i) {                 // ascending object index == try_add_all order (grid.rs:274-278)
            axis_range(0, boxes[i], hx); axis_range(1, boxes[i], hy); axis_range(2, boxes[i], hz);
            for (uint32_t ix : hx) for (uint32_t iy : hy) for (uint32_t iz : hz) {
                size_t slot = slot_of[((size_t)ix * ny + iy) * nz + iz];
                if (slot == (size_t)-1) continue;
                if (pass == 0) out.cell_offset[slot + 1]++;
                else out.prim_list[cursor[slot]++] = (uint32_t)i;
            }
        }
    }
    out.cell_aabb.resize(n_cells * 6);
    for (size_t idx = 0; idx < n_cells; ++idx) {
        double* a = &out.cell_aabb[idx * 6];
        if (owner[idx] == NONE) { for (int c = 0; c < 3; ++c) { a[c] = ob.mn[c]; a[3 + c] = ob.mx[c]; } continue; }
        uint32_t lc = owner[idx]; size_t iz = lc % nz, iy = (lc / nz) % ny, ix = lc / (nz * ny);
        double mn[3] = {coord[0][ix], coord[1][iy], coord[2][iz]};
        for (int c = 0; c < 3; ++c) { a[c] = mn[c]; a[3 + c] = mn[c] + d.box_size[c]; }
    }
    return true;
}

} // namespace rth
