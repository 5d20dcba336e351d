// Kernel parameters of the eye pass, workgroup / tile constants, the block -> tile mapping.  Part of libcgrt.so (cgrt_hip.hip).
#ifndef CGRT_GRID_HPP
#define CGRT_GRID_HPP
#include "cgrt_device_math.hpp"

// =====================================================================================================
// kernel parameters
// =====================================================================================================
struct GridParams {
    int32_t W, H, rows, row_offset, stripe_rows, stripe_rank, stripe_nranks;
    int32_t spp, sample_offset, max_depth;
    int32_t accumulate;  // CGRT_GRID_ACCUMULATE: rgb += this pass (nhit is overwritten)
    int32_t xcd_tiles;   // block -> tile mapping: 1 = XCD-aware super-tiles, 0 = row-major (see tile_of_block)
    // CGRT_GRID_SPLIT_SAMPLES: chunks > 1 workgroups per tile, workgroup c of a tile takes samples [c*chunk_spp, ...) and
    // leaves its raw fp64 sums in partial[c][local pixel][3] (and its hit count in partial_nhit[c][local pixel])
    int32_t chunks, chunk_spp;
    double *partial;
    uint32_t *partial_nhit;
    // Cost-aware scheduling (cgrt_hip.hip, "classify -> probe -> plan -> render -> ordered sum").  The unit of
    // bookkeeping is a WAVE TILE of 16x4 pixels, numbered wy * ceil(W/16) + wx over the local rows.
    //   probe != 0 : trace this launch's first sample only to measure it -- nothing is stored except cost[wave tile] =
    //                shader-clock ticks the wave spent on it.
    //   render     : order[0..K) = the HEAVY wave tiles (plan_kernel), plan[0] = K, hidx[wave tile] = rank among them or -1.
    //                The first heavy_blocks workgroups of the render launch (the unit-form body; they loop until the queue
    //                is empty) serve the heavy tiles through a queue of ITEMS (plan[2] = next item; item = heavy tile rank *
    //                items_per_tile + part): an item is units_per_item (pixel, sample) UNITS of one heavy tile, which the
    //                lanes of the wave take one after another as they become free, so a heavy tile is spread over many waves
    //                on many CUs and no lane idles while units remain.  Every Hitpoint value of a unit goes to
    //                dvals[rank][sample][emission index][pixel] (dcnt = how many), and deferred_sum_kernel adds them per
    //                pixel in the reference's order -- sample by sample, emission order within a sample -- so the fp64 sum
    //                is bit for bit the sequential one.  The other tiles are rendered in the tile form (waves whose tile is
    //                heavy stand down): through the tile queue below, or one workgroup per tile.
    // Light tiles (classify_kernel): light[wave tile] != 0 -- no primary ray of the tile can come near a mesh, a Bezier
    // object or a mirror / glass sphere, so it is rendered by the kernel variant without tree, Bezier and pending-ray code
    // (fewer registers, more waves per SIMD; with bump-mapped diffuse planes: the tree-capable variant without Bezier and
    // pending-ray code), launched beside the full variant on a second stream.  light_mode: 0 = this
    // launch leaves the light tiles alone, 1 = this launch renders only them; light == nullptr: no split.
    const unsigned char *light;
    int32_t light_mode, pad_light_;
    const uint32_t *order;
    // Tile queue of the scheduled launch (chunks == 1): border[0..plan[3]) = the tiles (ty * tiles_x + tx) with at least one
    // wave tile that is neither heavy nor light, costliest first (plan_kernel); plan[4] = next entry.  The launch is then
    // heavy_blocks + a chip's worth of workgroups, each serving one queue until it is empty and then the other, so neither
    // empty tiles nor a late expensive tile cost anything at the end of the frame.  nullptr: one workgroup per tile.
    const uint32_t *border;
    uint32_t *cost;
    const int32_t *hidx;
    uint32_t *plan;
    double *dvals;
    unsigned char *dcnt;
    double *pconst;  // heavy tiles: per pixel {pdir(3), pof(3), bits of k_pix}, layout [rank][7][64], filled by pixel_const_kernel
    int32_t probe, heavy_blocks, items_per_tile, units_per_item, maxhp;
    // Primary-ray mesh hits of the heavy tiles' units, computed by primary_walk_kernel before the render launch
    // (cgrt_primwalk.hpp): [rank][sample][pixel] distance (kInf: none) and triangle (-1: none) in object prim_obj; nullptr: off
    const double *prim_len;
    const int32_t *prim_tri;
    int32_t prim_obj;
    int32_t prim_done;  // primary_walk_kernel also completes units (dcnt != 255: done there, the unit-queue body skips them)
    int32_t pw_refill, pw_rounds;  // primary_walk_kernel: idle lanes that trigger a refill; inner-node rounds between leaf phases
    // development aid (env CGRT_TIMELINE_FILE, cgrt_hip.hip): per workgroup {start, end (wall_clock64, 100 MHz), HW_ID | XCC_ID << 32,
    // tile_x | tile_y << 16 | rays << 32}; nullptr in normal operation
    unsigned long long *timeline;
    double inv_spp_total;
    uint64_t seed;
    double cam[3], half_width, focus_plane, lens_radius;
};

static constexpr int kTileW = 32, kTileH = 8, kThreads = 256;
static constexpr int kWaveTileW = 16, kWaveTileH = 4;  // one pixel per lane
static constexpr uint32_t kNoWaveTile = 0xffffffffu;
// Pending refracted rays (main.cpp:157) of a lane, newest last:
//   * a glass hit whose children are leaves of the recursion (depth_left == 2) keeps the refracted child in
//     REGISTERS (it is consumed right after the reflected child, before any other push) -- in a full glass tree
//     that is 8 of the 15 pushes;
//   * the first two other levels live in LDS: per level 9 doubles + one packed (depth, path) word per thread,
//     layout [level][field][thread] (conflict-free), 2 x 19 456 B = 38 912 B per workgroup;
//   * a third level (three nested glass hits with all siblings waiting) spills to scratch memory.
// The output needs no LDS (each wave transposes its 16x4 tile with lane shuffles), so stack + objs stays under 40 KiB and
// FOUR workgroups fit a CU's 160 KiB: occupancy 4 waves/SIMD instead of 3, worth ~10 % on C2.
static constexpr int kPendDoubles = 9;
static constexpr int kLdsLevels = 2;
static constexpr size_t kLevelBytes = (size_t)kThreads * (kPendDoubles * sizeof(double) + sizeof(uint32_t));
static constexpr size_t kStackBytes = (size_t)kLdsLevels * kLevelBytes;
static constexpr size_t kTileBytes = (size_t)8 * 32 * 3 * sizeof(float);

// local row -> global row (cgrt.h: block-cyclic stripes)
__device__ __forceinline__ int global_row(const GridParams &g, int j) {
    if (g.stripe_nranks > 1) {
        int S = g.stripe_rows;
        return ((j / S) * g.stripe_nranks + g.stripe_rank) * S + (j % S);
    }
    return g.row_offset + j;
}

// blockIdx -> tile, XCD-aware.  Workgroups are dealt round-robin to the 8 XCDs, each with a private 4 MiB L2, so the
// blocks b, b+8, b+16, ... share an L2.  Tiles are grouped in super-tiles of kSuperW x kSuperH tiles (128 x 32 pixels);
// the blocks of one XCD group walk one super-tile after another, so the tiles an L2 serves at any moment are neighbours in
// the image and want the same tree nodes, triangles and texels -- while successive super-tiles alternate between the XCD
// groups, which keeps the expensive part of a frame (a mesh in one corner) spread over all of them.  Only placement
// changes: every tile is still rendered exactly once by exactly one workgroup.
// Measured (MI355X): C3 (glass bunny) 50.3 -> 46.6 ms, C4 (dragon) 205.5 -> 197.2 ms; but C2 4.05 -> 4.47 ms and the
// Bezier vase 8.1 -> 9.0 ms -- scenes with no tree to share, whose expensive tiles (glass sphere, vase) then sit on
// one or two XCDs.  So the launch picks it for scenes with meshes and no Bezier object, row-major otherwise.
static constexpr int kXcds = 8, kSuperW = 4, kSuperH = 4, kSuperTiles = kSuperW * kSuperH;
// Workgroup shapes.  NT = 256: four waves on a 32x8-pixel tile (2x2 sub-tiles of 16x4).  NT = 64: ONE wave on a 16x4 tile --
// a workgroup's wave slots and LDS are only handed on when its LAST wave retires, so with four very unequal waves (a
// Bezier vase covering part of a tile: a wave over it works ~100x longer than its neighbours) slots idle; with one wave
// per workgroup every slot is reused the moment its wave ends.
template <int NT>
struct TileGeom {
    static_assert(NT == 256 || NT == 64, "workgroup = 4 waves or 1 wave");
    static constexpr int W = NT == 256 ? 32 : 16, H = NT == 256 ? 8 : 4;
    static constexpr size_t level_bytes = (size_t)NT * (kPendDoubles * sizeof(double) + sizeof(uint32_t));
    static constexpr size_t stack_bytes = (size_t)kLdsLevels * level_bytes;
    static constexpr size_t tile_bytes = (size_t)W * H * 3 * sizeof(float);
};
static_assert(TileGeom<256>::level_bytes == kLevelBytes && TileGeom<256>::tile_bytes == kTileBytes, "TileGeom<256>");

__host__ __device__ inline int tile_grid_blocks(int W, int rows, bool xcd_tiles, int tile_w = kTileW, int tile_h = kTileH) {
    const int tiles_x = (W + tile_w - 1) / tile_w, tiles_y = (rows + tile_h - 1) / tile_h;
    if (!xcd_tiles) return tiles_x * tiles_y;
    const int sx = (tiles_x + kSuperW - 1) / kSuperW, sy = (tiles_y + kSuperH - 1) / kSuperH;
    const int nsuper = sx * sy;
    return ((nsuper + kXcds - 1) / kXcds) * kXcds * kSuperTiles;
}
// false: this block has no tile (edge of the super-tile grid)
__device__ __forceinline__ bool tile_of_block(const GridParams &g, int b, int &tile_x, int &tile_y, int tile_w = kTileW,
                                              int tile_h = kTileH) {
    const int tiles_x = (g.W + tile_w - 1) / tile_w, tiles_y = (g.rows + tile_h - 1) / tile_h;
    if (!g.xcd_tiles) {
        tile_x = b % tiles_x;
        tile_y = b / tiles_x;
        return true;
    }
    const int sx = (tiles_x + kSuperW - 1) / kSuperW;
    const int group = b % kXcds, q = b / kXcds;
    const int super = (q / kSuperTiles) * kXcds + group, t = q % kSuperTiles;
    tile_x = (super % sx) * kSuperW + t % kSuperW;
    tile_y = (super / sx) * kSuperH + t / kSuperW;
    return tile_x < tiles_x && tile_y < tiles_y;
}

#endif
