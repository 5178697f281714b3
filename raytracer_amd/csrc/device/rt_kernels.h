// rt_kernels.h — kernel argument structs shared by rt_kernels.hip and rt_api.hip
#pragma once
#include "rt_device.h"

#define RT_BLOCK 128          // extend kernels: two wave64 per workgroup
#define RT_LBLOCK 1024        // extend_bvh2 with the tree in LDS: one 16-wave workgroup per CU shares one copy
#define RT_LSTK 4             // ... and keeps four continuation-stack entries per lane in LDS instead of eight (64 KB of the 160)
#ifndef RT_SBLOCK
#define RT_SBLOCK 512
#endif                        // streaming kernels (generate, shade, accumulate): block-level queue compaction

// SoA ray queue in HBM: one entry per live path (origin, direction, throughput, path id)
struct RayQ { double *ox, *oy, *oz, *dx, *dy, *dz, *tr, *tg, *tb; uint32_t* id; };

// where extend writes its result; trace != 0 selects the parity-entry format (rt_trace)
// render format: prim = index of the last accepted primitive, -1 = miss, RT_HIT_EDGE = the ray touched a box edge (draw-boxes)
#define RT_HIT_EDGE (-2)
// tn (nullable) = t of the accepted primitive; only stored for scenes that mix spheres and triangles
struct ExtendOut { int32_t* prim; double* t; double* tp; double* tn; uint8_t* hit; int trace; };

// continuation of BvhNode::hit (bvh.rs:210-223): the pending far child and its entry distance
struct __align__(16) StackEntry { double far_d; uint32_t a, b; };

// root box of the BVH for the early sky test in k_generate (enabled = 0 for naive / grid)
struct DRootBox { int enabled; double mn[3], mx[3]; };

struct DCamera { uint32_t W, H; V3 center, pixel00, du, dv, ddu, ddv; double defocus_angle; };

struct DGrid {
    V3 box_size, origin, end, total_size; uint64_t n_cells; uint32_t max_depth;
    const double* cell_aabb; const uint64_t* cell_offset; const uint32_t* prim_list;
    const DCellTri* cell_tri;      // nullable: per list position, the triangle's plane record and index (all-triangle scenes)
    // cooperative cell test (k_extend_grid_coop; all-triangle scenes): cells whose list has at least coop_min entries keep it a
    // second time as 64-entry SoA chunks; cell_chunk[cell] = index of the cell's first chunk, RT_NO_CHUNK for the other cells
    const uint32_t* cell_chunk; const DCellChunk* chunks;
    // Two bits per cell (16 cells per word), nullable: bit 0 = the cell's list is not empty, bit 1 = the cell's stored box is NOT
    // origin + (ix, iy, iz) * size .. + size bit for bit (clone cells, grid.rs:24-27,42-45).  A fine grid is almost all code 0:
    // such a cell needs neither its 48-byte box nor its two list offsets from HBM (8 M cells at size 1: 450 MB of tables against
    // a 2 MB code table that stays in L2).  nx, ny, nz = cells per axis (only set with cell_code: integer size that tiles the world).
    const uint32_t* cell_code; uint32_t nx, ny, nz;
};
#define RT_NO_CHUNK 0xffffffffu

enum { ST_SCATTER = 0, ST_BOX, ST_PRIM, ST_STEPS, ST_OVERLAP, ST_GRIDBOX, ST_SEGMENTS, ST_N };

__global__ void k_generate(DCamera cam, DRootBox root, uint32_t s_base, uint32_t s_count, uint32_t lg_s, uint64_t seed, RayQ out, uint32_t* count_out,
                           double* rad, unsigned long long* stats);
__global__ void k_generate_list(DCamera cam, uint32_t sample, uint64_t seed, const uint32_t* pi, const uint32_t* pj,
                                uint64_t n, double* orig, double* dir);
// DRAW = Options.draw_boxes (rtweekend.rs:92): the sticky rec.hits_aabb_edge of aabb.rs:57-60 is tracked per ray
template <bool DRAW>
__global__ void k_extend_bvh(const DNode* nodes, const DPrim* prims, RayQ in, const uint32_t* count, uint64_t n_fixed,
                             double tmin0, double tmax0, ExtendOut out, StackEntry* stack, unsigned long long* stats);
// tree and primitives as the persistent extend kernel reads them (rt_device.h):
//   NODES 0 = DPair64 (f64 planes), 1 = DPair32 (lattice), 2 = DPairN (lattice, new planes only: two accesses per visit);  PRIMS 0 = DPrim (any mix), 1 = DPrim32 (lattice triangles), 2 = DSphere
struct BvhView { const DNode* nodes; const DPrim* prims; const DPair64* pairs64; const DSphere* spheres;
                 const DPair32* pairs32; const DPrim32* prims32; double scale; const DPairN* pairsn; uint32_t root_own[6];
                 uint32_t lds_pairs16, lds_prims16;
                 uint32_t pair64_stride; };              // bytes between generic pair records: 112 (packed) or 128 (one record per L1 line, big trees)    // LDS variant: 16-byte words of pairs64 and of spheres / prims to stage
template <int NODES, int PRIMS, bool TRACE, bool LDS = false>
__global__ void k_extend_bvh2(BvhView bvh, RayQ in, const uint32_t* count, uint64_t n_fixed,
                              double tmin0, double tmax0, ExtendOut out, StackEntry* spill, uint32_t* fetch_ctr, uint32_t tune, unsigned long long* stats, unsigned long long* dbg);
__global__ void k_tri_pre(const DPrim* prims, uint32_t n, DTriPre* pre);
__global__ void k_cell_tri(const uint32_t* prim_list, const DTriPre* pre, uint64_t n, DCellTri* out);
__global__ void k_extend_naive(const DPrim* prims, const DTriPre* pre, uint32_t n_prims, RayQ in, const uint32_t* count, uint64_t n_fixed,
                               double tmin0, double tmax0, ExtendOut out, unsigned long long* stats);
template <bool DRAW>
__global__ void k_extend_grid(DGrid g, const DPrim* prims, const DTriPre* pre, RayQ in, const uint32_t* count, uint64_t n_fixed,
                              double tmin0, double tmax0, ExtendOut out, uint32_t* fetch_ctr, unsigned long long* stats);
__global__ void k_grid_codes(DGrid g, uint32_t* codes);
__global__ void k_cell_chunks(const uint64_t* cell_offset, const uint32_t* prim_list, const uint32_t* big_cells, const uint32_t* big_first, uint32_t n_big,
                              const DPrim* prims, const DTriPre* pre, DCellChunk* chunks, uint32_t n_chunks);
__global__ void k_extend_grid_coop(DGrid g, const DPrim* prims, const DTriPre* pre, RayQ in, const uint32_t* count, uint64_t n_fixed,
                                   double tmin0, double tmax0, ExtendOut out, uint32_t* fetch_ctr, unsigned long long* stats);
template <int MODE>
__global__ void k_shade(const DPrim* prims, const DSphere* spheres, const DPrim32* tris32, double tri_scale, const DMat* mats, uint32_t n_mats, RayQ in, const uint32_t* count_in,
                        const int32_t* hit_prim_, const double* hit_tp, const double* hit_tn, RayQ out, uint32_t* count_out,
                        double* rad, uint32_t WH, uint32_t s_base, uint32_t s_count, uint32_t bounce, uint32_t max_depth, uint64_t seed,
                        unsigned long long* stats);
__global__ void k_accumulate(const double* rad, uint32_t WH, uint32_t s_count, double* fb, int first_batch);
__global__ void k_load_rays(const double* orig, const double* dir, uint64_t n, RayQ q);
