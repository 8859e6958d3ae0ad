// srz_device.h — device-side data layout shared by the kernels (srz_kernels.hip) and the C-ABI host code
// (srz_api.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/srz.h"

namespace srz {

constexpr int TILE = 32;             // one wavefront owns one 32x32-pixel tile (its z + owner planes live in LDS)
constexpr int BAND = 32;             // band = one row of tiles; the unit of multi-GPU sharding and of binning
// k_raster: ONE wave (= one tile) per workgroup, so a long tile never pins the LDS / wave slots of finished neighbours and
// the dispatcher load-balances tile by tile.  Its tile state is one 64-bit key per pixel; rows are padded to 33 keys
// (264 B): lanes that hit the same column of consecutive rows land in different bank pairs, and the 33rd key of every
// row doubles as scratch for the item expansion.  8448 B per wave = 19 waves per CU.
constexpr int KEY_STRIDE = TILE + 1;
// LDS row stride (dwords) of the ORDERED rasteriser's z / owner planes (k_raster_slow)
constexpr int LDS_STRIDE = 32;
// ---- which rank owns which 32-row band (round 6) ----------------------------------------------------------------------------------
// Every group g = b / world of `world` consecutive bands hands ONE band to every rank, ROTATED by BAND_ROT steps per group:
//     rank(b) = (b + BAND_ROT * (b / world)) % world,   local band index = b / world,
//     band(lb, rank) = lb * world + (rank - BAND_ROT * lb) mod world.
// (Rounds 1-5: rank = b % world.  A scene whose period in bands is a multiple of `world` — config 4's four rows of cows, 16 bands each,
// on 8 ranks — then gave rank r the SAME two slices of every cow: per-rank render times 1.19-1.21 x their mean.  With a rotation a rank
// meets a different slice in every group.  Measured with one GPU playing every rank of 8 (bench.py emulate_shards, max / mean of the
// per-rank render times, configs 2 / 4 / 5): no rotation 1.05-1.06 / 1.19-1.21 / 1.02-1.04; 1 step 1.15 / 1.07-1.09 / 1.02-1.04; 3 steps
// 1.12 / 1.09 / 1.02; 5 steps 1.06 / 1.09 / 1.04 — five it is (one step where five is a multiple of the world).  Buffers, local indices
// and the exchange are unchanged: the gathered layout is still [rank][frame][plane][local band][32 rows].)
#ifndef SRZ_BAND_ROT
#define SRZ_BAND_ROT 5 /* (A/B builds: 0 = the plain b % world of rounds 1-5) */
#endif
__host__ __device__ inline int band_rot(int world) { return (SRZ_BAND_ROT % world) ? SRZ_BAND_ROT : (SRZ_BAND_ROT ? 1 : 0); }
__host__ __device__ inline int band_of(int lb, int rank, int world) {
  if (world == 1) return lb; // (the single-GPU case pays no division: a wave-uniform branch in the kernels)
  int j = (rank - band_rot(world) * lb) % world;
  return lb * world + (j < 0 ? j + world : j);
}
__host__ __device__ inline int rank_of_band(int b, int world) { return (b + band_rot(world) * (b / world)) % world; }
constexpr uint32_t NO_TRI = 0xffffffffu;
// per-frame atomic counters sit 128 bytes apart: atomics on the words of ONE cache line serialise (~10 ns each across the
// device) as if they were one address — with 16 frames of 4096^2 that was the whole of k_raster's time
constexpr uint32_t CNT_STRIDE = 32;
// FrameDesc::flags, internal (set by the host): 1..4 lights and an integer exponent 0..256 → the FAST build of k_shade for that count
constexpr uint32_t FD_FAST_SHADE = 0x100u, FD_NL_SHIFT = 9; // (+ the light count 1..4 in bits 9..11)
constexpr uint32_t FD_BUMPY = 0x1000u;                   // some batch of the frame is BUMP / DISPLACEMENT (FAST builds with those variants)
constexpr uint32_t FD_PACKED = 0x4000u;                  // fewer than 2^22 - 1 triangles and at most 1024 batches: a tile-list entry is
                                                         // triangle index | batch << 22 (k_shade stages the tile's triangles without
                                                         // a gather of their batch ids), and the depth keys' tie-breaks carry list positions
constexpr uint32_t PACK_IDX_BITS = 22, PACK_IDX_MASK = (1u << PACK_IDX_BITS) - 1u, PACK_MAX_BATCHES = 1024;
constexpr uint32_t FD_GREY = 0x8000u;                    // ka, ks and every light's intensity have three bit-equal channels (FrameK::grey in srz_kernels.hip)
constexpr uint32_t FD_GENPOW = 0x2000u;                  // a non-integer exponent in (0, 4096]: FAST builds whose power is pow_fast (exp2(p log2 x) in binary64 with a rounding-safety flag)
constexpr uint32_t SHADE_KIND_GENERIC = 12;              // k_shade builds: kinds 0..3 = FAST for 1..4 lights, 4..7 = the same + BUMPY,
                                                         // 8..11 = FAST for 1..4 lights with any exponent (GENPOW), 12 = generic
constexpr uint32_t N_WORK_LISTS = 8 * (SHADE_KIND_GENERIC + 1);
constexpr uint32_t UNLISTED = 0xffffffffu; // tile_off of a tile whose list did not fit the record pool
// Binning is O(triangles): k_setup / k_chunks sort every GROUP of GROUP_TRIS (512) consecutive triangles by the 32-row bands their
// bounding boxes reach (LDS count / scan / fill, no global atomics) into the group's own region of ENT_PER_GROUP 8-byte
// entries {triangle, tile-x range}, and write one descriptor word per (group, local band): first entry << 16 | entries.
// k_bin's (frame, band) workgroup then reads only its own entries.  A group whose entries do not fit the region (on average
// more than 4 bands per triangle: huge triangles) gets DESC_RAW in every band, and the band workgroups walk that group's
// bounding boxes themselves.
#ifndef SRZ_GROUP_K
#define SRZ_GROUP_K 2 // triangles per thread of a k_setup workgroup pass (measured 1 / 2 / 4: setup + bin 52 / 56 / 58 µs per 256 frames
                      // of config 2, 106 / 99 / 104 µs per 32 frames of config 4)
#endif
constexpr uint32_t GROUP_K = SRZ_GROUP_K, GROUP_TRIS = 256u * GROUP_K, ENT_PER_GROUP = 4u * GROUP_TRIS, DESC_RAW = 0xffffffffu,
                   MAX_LOCAL_BANDS = 1024;
constexpr int MAX_TEX = 64;
constexpr int MAX_MESH = 256;

// Screen-space bounding box of a surviving triangle (Triangle::calcBoundingBox, src/Triangle.cpp:243-257),
// 8 bytes so that a wave scans 64 of them with one coalesced 512-B load. Culled / non-finite: sx > ex.
struct __attribute__((aligned(8))) BBox {
  int16_t sx, sy, ex, ey;
};

// The triangle stream in HBM: the boundary's 96-byte srz_tri records (36 bytes of positions, 60 of normals + texture
// coordinates) as uploaded — what k_shade stages per tile — plus a DENSE COPY OF THE POSITIONS, 9 floats per triangle, made at
// upload (or written by k_vertex beside the record).  k_setup streams the dense copy (36 bytes per triangle instead of the 96
// its cache lines used to drag in) and k_raster gathers a tile's triangles from it through the 4-byte indices of the tile's list,
// the bounding box from bbox[] (8 bytes): there is no per-triangle record written by k_setup any more (it was 48 bytes written
// per kept triangle, every frame).  The kernels address triangle i's positions at tri_pos + i * pos_stride; srz_draw, which
// re-uploads one frame's records per call, makes no copy: tri_pos = the records themselves, pos_stride = 24.
// (the per-triangle constants of the coverage tests — 1 / fmsub(ABx,ACy,ACx*ABy) for the V columns, ABx*ACy - ABy*ACx for the
// S columns — are recomputed by k_raster from the positions, once per list entry and 64 entries at a time)
constexpr uint32_t TRI_POS_F = 9, TRI_AOS_F = 24; // floats

// What the Shader object bound to a batch holds (type + texture), resolved on the host at render time
struct __attribute__((aligned(8))) ShadeDescG {
  int32_t shader, tw, th, _pad;
  const uint32_t *tex;
};

// One mesh instance of one frame for the device vertex stage
struct DrawDesc {
  const srz_vertex *verts;
  const uint32_t *faces;
  uint32_t n_faces;
  uint32_t tri_off; // first output triangle (global index into tris[])
  uint32_t frame;   // the frame this draw belongs to (k_vertex's own triangle setup reads its size and eye)
  float zscale, zoffset;
  float ndc_mvp[16], normal_m[16];
};

struct FrameDesc {
  int32_t width, height;
  float eye[3];
  float ka[3];
  float ks[3];
  float p, kh, kn;
  uint32_t n_lights, light_off;  // into lights[]
  uint32_t n_tris, tri_off;      // into tris[] / bbox[] / tri_batch[]
  uint32_t n_batches, batch_off; // into batches[]
  uint32_t flags;
  uint32_t n_local_bands;        // bands of this frame owned by this ctx
  uint32_t chunk_off;            // into chunk_rows[]: first 64-triangle chunk of this frame
  uint32_t group_off;            // into band_desc[] / band_ent[]: first group (GROUP_TRIS triangles) of this frame
};

struct BatchDesc {
  int32_t shader, tex_id;
  uint32_t first, count; // triangle range inside the frame
};

struct TexDesc {
  const uint32_t *bgrx; // one dword per texel: B | G<<8 | R<<16
  int32_t w, h;
};

// counters accumulated by the STATS kernel variants (same order as srz_stats)
enum { ST_TRIS = 0, ST_CULLED, ST_PIXEL_TESTS, ST_FRAGMENTS, ST_SHADED, ST_VISIBLE, ST_VISIBLE_TEX,
       ST_DBG_CYC_A, ST_DBG_CYC_B, ST_DBG_CYC_C, ST_DBG_MAX_WAVE, ST_DBG_IEEE_TILES, ST_DBG_BLOCKS, ST_COUNT };

struct RenderArgs {
  const FrameDesc *frames;
  const srz_tri *tris;           // the records (k_shade)
  const float *tri_pos;          // [triangle * pos_stride]: ax ay z0 bx by z1 cx cy z2 (k_setup, k_raster)
  uint32_t pos_stride;           // floats: 9 (the dense copy) or 24 (the records themselves)
  const BBox *bbox;
  uint32_t *chunk_rows;          // per 64-triangle chunk: min sy | max ey << 16 of its kept triangles (k_setup → k_bands)
  uint32_t *band_desc;           // [group][n_local_bands]: first entry << 16 | entries of the group in that band, or DESC_RAW
  uint2 *band_ent;               // [group][ENT_PER_GROUP]: {triangle index in the frame, first tile x | last tile x << 16}
  const uint16_t *tri_batch;
  const BatchDesc *batches;
  const srz_light *lights;
  const TexDesc *tex;
  const ShadeDescG *sdesc;       // per batch (indexed like batches[])
  // per-tile triangle lists, UNORDERED (the rasteriser's result does not depend on list order): triangle indices in a pool
  // of n_sub equal sub-pools with one bump allocator each; a (frame, band) workgroup of k_bin takes its band's entries
  // from sub-pool (workgroup id & sub_mask) in one allocation.  A band that does not fit is left UNLISTED: its tiles are
  // rasterised by k_raster_slow straight from the frame's stream, and the host grows the pool before the next render.
  uint32_t *pool;
  uint32_t *pool_heads;          // [n_sub] records requested from each sub-pool by this render (zeroed by k_setup)
  uint32_t *pool_demand;         // pinned host memory laid out like pool_heads: what this render asked of each sub-pool (latency build of k_raster)
  uint32_t pool_sub_cap, pool_sub_mask;
  uint2 *tile_info;              // [frame][local band][tiles_x]: x = entries in the tile's list (0: k_clear's tile), y = its first
                                 // index in pool[], or UNLISTED — one 8-byte load tells a tile's wave both
  uint32_t *slow_list;           // tiles (frame * tiles_per_frame + tile) left to the ordered rasteriser
  uint32_t *slow_count;
  uint32_t other_streams;        // host hint: the previous render of this ctx went to another stream (lanes): k_shade's grid follows
  uint32_t clear_in_raster;      // small jobs: no k_clear is launched, k_raster's waves clear the tiles no bbox reaches
  uint32_t force_ordered;        // every touched tile goes to k_raster_slow (SRZ_ORDERED_RASTER, counting runs)
  uint32_t any_ordered;          // host hint: SRZ_ORDERED_RASTER is set on the render or on some frame (k_raster_slow gets a large grid)
  uint32_t force_generic;        // every frame is shaded by the generic build of k_shade (counting runs)
  uint32_t any_generic;          // some frame is not FD_FAST_SHADE (else the generic build only serves redo_list)
  uint4 *redo_list;              // work-list entries of the tiles the FAST builds of k_shade hand to the generic one
  uint32_t *redo_count;
  uint32_t *vis;                 // owner ids per tile [frame][local band][tile x][PIX_SLOT dwords]: 16 bits per pixel (position in
                                 // the tile's triangle list) or 32 (index in the frame) — written by the rasterisers' write-out for
                                 // tiles that have an owner, read by k_shade (srz_kernels.hip, PIX_SLOT)
  // k_shade's work: 8 lists [frame % 8] per build kind in use (FAST for 1..4 lights in 3 forms, generic) of the tiles that have an owner, as
  // {frame * tiles_per_frame + (lb*tiles_x + tx), flags, list entries, list offset} (srz_kernels.hip, work_append), in arrival
  // order; work_cap entries each
  uint4 *worklist;
  uint64_t kind_slots;           // storage slot of build kind k in worklist[]: (kind_slots >> 4k) & 15 — lists exist only for the kinds the
                                 // set's frames need (the counters in work_count[] are indexed by kind, all of them exist)
  uint32_t *work_count;          // [list * CNT_STRIDE]: word 0 = entries (zeroed by k_setup, bumped by k_raster), word 1 = k_shade's cursor
  uint32_t work_cap;
  uint32_t tiles_x, n_local_bands, n_frames;
  float *out;             // [frame][4][local_rows][width]
  uint64_t frame_stride;  // floats per frame in out = 4*local_rows*width
  uint32_t local_rows;    // rows per plane in out
  int32_t shard_rank, shard_world;
  uint32_t flags_or;      // OR-ed into every frame's flags (SRZ_FUSED_CLEAR / SRZ_UNIFIED)
  unsigned long long *stats;
  unsigned long long *timeline; // diagnostic (STATS variant only): per tile {start, end (wall clock 100 MHz), hw_id, blocks}
  const uint32_t *clear_wgs_dev; // k_clear: workgroups that take part, read on the device (ClearCtl::wgs; 0 = the first candidate); null: the whole grid
};

// The grid of the side-stream clear is MEASURED per frameset, on the device (srz_api.hip, srz_frameset::ClearTune; k_clear_tune): while a
// set measures, k_clear is launched with CLEAR_GRID_MAX workgroups of which the first ClearCtl::wgs take part, and a one-thread kernel at
// the end of every render reads the wall clock, files the time since the previous render's end under the grid in use and moves on —
// candidates a b c (ascending) in blocks of CLEAR_TUNE_BLOCK renders (the first sample of a block, the previous grid's tail, is dropped;
// a grid more than 5 % behind the best smaller one ends this pass: the larger ones would only be worse and are not tried); a grid tried
// after the best of them and more than 5 % behind it is dropped, the others go round once more in the mirrored order c b a (the first
// renders after idle run up to 10 % slower: the mirror takes a linear ramp out of the comparison, and the ramp is why an EARLIER grid is
// never dropped on the first pass), the smallest median of the four samples wins.  No host synchronisation, no event query: the
// decision takes effect with the next render whatever the host's run-ahead, and reaches the host through a word of mapped memory.
constexpr int CLEAR_CANDS = 3, CLEAR_TUNE_BLOCK = 3, CLEAR_TUNE_RENDERS = 2 * CLEAR_CANDS * CLEAR_TUNE_BLOCK;
constexpr uint32_t CLEAR_CAND[CLEAR_CANDS] = {96, 128, 256};
constexpr uint32_t CLEAR_GRID_MAX = 256, CLEAR_GRID_DEFAULT = 96;
struct ClearCtl {
  uint32_t wgs;                  // workgroups of the NEXT render's clear (0: CLEAR_CAND[0] — the state after a memset)
  uint32_t cur, pos, phase, done, alive;
  uint32_t n[CLEAR_CANDS];
  unsigned long long last;       // wall clock (100 MHz) at the end of the previous render
  float t[CLEAR_CANDS][4];       // samples, wall-clock ticks
  float score[CLEAR_CANDS];      // what the decision compared (ticks; 0: dropped after the first pass)
};
void launch_clear_tune(ClearCtl *ctl, uint32_t *h_wgs, hipStream_t s);

void launch_vertex(const DrawDesc *draws, uint32_t n_draws, uint32_t max_faces, srz_tri *tris, float *tri_pos, const FrameDesc *frames,
                   BBox *bbox_out, hipStream_t s);
void launch_chunks(const RenderArgs &a, int n_frames, uint32_t max_tris, hipStream_t s);
void launch_setup(const RenderArgs &a, int n_frames, uint32_t max_tris, bool stats, hipStream_t s);
void launch_bin(const RenderArgs &a, int n_frames, uint32_t max_tris, hipStream_t s);
void launch_raster(const RenderArgs &a, int n_frames, bool stats, hipStream_t s);
bool raster_four_waves(const RenderArgs &a); // the latency build of k_raster serves this job (it also reports the pool's demand)
void launch_clear(const RenderArgs &a, uint32_t max_tiles, bool beside_raster, hipStream_t s, uint32_t wgs);
void launch_shade(const RenderArgs &a, uint32_t max_tiles, bool stats, uint32_t fast_mask, bool any_generic, bool approx, hipStream_t s);
void launch_resolve8(const float *planes, uint8_t *out, uint32_t n_frames, uint32_t rows, uint32_t W, uint64_t frame_stride,
                     hipStream_t s);
void launch_deinterleave(const void *gathered, void *full, uint32_t world, uint32_t n_fp, uint32_t bands_per_rank, uint32_t row_bytes,
                         hipStream_t s);
void launch_verify_fastmath(unsigned long long *d_out4, hipStream_t s);
void launch_verify_fastdiv(unsigned long long *d_out3, hipStream_t s);
void launch_verify_fastpow(unsigned long long *d_out3, float p, hipStream_t s);
void launch_verify_fastlen(unsigned long long *d_out4, hipStream_t s);
void launch_tex_convert(const uint8_t *d_bgr, int w, int h, int row_stride, uint32_t *d_bgrx, hipStream_t s);

} // namespace srz
