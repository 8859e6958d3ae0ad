// srz_api.hip — the C ABI declared in include/srz.h (host side: contexts, framesets, uploads, launches).
// No CPU fallback exists: without a usable gfx950 device every compute entry point returns SRZ_E_NODEVICE.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "srz_device.h"

using namespace srz;

namespace {
std::string g_create_error;

struct EventPair {
  hipEvent_t t0, t1, t2, t3; // t0..t1 setup+binning, t1..t2 raster (visibility + clear), t2..t3 shade
  bool detailed;             // t1 / t2 were recorded (every recorded event is a barrier in the launch stream: ≈4 µs each)
};
} // namespace

struct srz_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  int shard_rank = 0, shard_world = 1;
  TexDesc h_tex[MAX_TEX];
  uint32_t *d_texmem[MAX_TEX];
  TexDesc *d_tex = nullptr;
  unsigned long long *d_stats = nullptr;
  struct MeshSlot {
    srz_vertex *d_verts = nullptr;
    uint32_t *d_faces = nullptr;
    uint32_t n_verts = 0, n_faces = 0;
  } mesh[MAX_MESH];
  int timing = 0; // 0 off, 1 whole launch set only (2 events per render), 2 per-kernel groups as well (4 events)
  std::vector<EventPair> ev_pool, ev_used;
  double acc_ms[4] = {0, 0, 0, 0}; // bin, raster, shade, total
  uint64_t tex_version = 1;
  int acc_launches = 0;
  std::vector<float> acc_samples; // whole-launch-set time of every timed render since the last reset (bounded)
  // span of the timed renders since the last reset: first t0 → latest t3 (renders on several streams overlap; the span is
  // what their launch sets took together)
  hipEvent_t span_t0 = nullptr;
  bool span_open = false;
  double span_ms = 0.0;
  unsigned long long dbg[ST_COUNT] = {};
  // srz_draw / srz_draw_scene keep their frameset and device framebuffer between calls: a call whose structure (size,
  // batch sizes, shader types, light count) equals the previous one only re-uploads the data
  srz_frameset *draw_fs = nullptr;
  float *draw_out = nullptr;
  std::vector<uint64_t> draw_sig;
  // diagnostic switches, read ONCE when the ctx is created (never per render): frames per sub-batch of a large set (0: the
  // default below), 32-bit owner ids even where 16 would do
  int env_sub_batch = 0;
  bool env_no_packed = false; // SRZ_NO_PACKED (tests): see srz_frameset::no_packed
  bool env_no_turns = false;  // SRZ_NO_TURNS (A/B): renders on different streams do not wait for each other's k_raster
  uint32_t env_clear_wgs = 0; // SRZ_CLEAR_WGS: fixed grid of the side-stream clear (else measured per set, srz_frameset::ClearTune)
  // what sets of this ctx have measured, by shape (clear_memo_key): a new set of a known shape starts with that grid instead of measuring
  std::vector<std::pair<uint64_t, uint32_t>> clear_memo;
  bool env_no_clear_tune = false, env_clear_trace = false; // SRZ_CLEAR_TUNE=0: the grid stays at 96; SRZ_CLEAR_TRACE=1: the measurement goes to stderr
  bool opt_approx_shade = false; // SRZ_OPT_APPROX_SHADE (srz_set_option): framesets created from now on shade in the tolerance mode
  bool opt_pool_lazy = false; // SRZ_OPT_POOL_LAZY (srz_set_option; initial value: the environment variable SRZ_POOL_LAZY, read in srz_create)
  hipStream_t stream2 = nullptr; // k_clear runs here, next to k_raster
  float *batch_out = nullptr;    // srz_draw_batch's device planes, kept between calls (grown on demand)
  size_t batch_out_bytes = 0;
  hipStream_t stream3 = nullptr; // srz_draw_batch: the read-back of one piece of the batch under the render of the next
  hipEvent_t ev_piece[8] = {};   // (EV_RING of them: "piece k has been rendered")
  static constexpr int EV_RING = 8;  // fork/join events are used round-robin: a render never re-records an event that
  hipEvent_t ev_fork[EV_RING] = {}, ev_join[EV_RING] = {}; // a wait of the previous few renders may still refer to
  unsigned ev_next = 0;
  // Renders submitted to DIFFERENT streams (LaneRenderer) take turns in the setup..raster phase: the next one's k_setup waits
  // for the previous one's k_raster.  Two k_rasters side by side slow each other (both LDS-bound) and then leave two k_shades
  // side by side (both VALU-bound); taking turns puts one stream's raster beside the other's shade, which is the overlap that
  // pays (MI355X, 2 x 128 frames of 1024^2: 1.13 → 1.09 ms per batch).  Renders on one stream are unaffected.
  hipEvent_t ev_raster[EV_RING] = {};
  unsigned raster_next = 0;
  hipStream_t raster_last_stream = nullptr;
  bool raster_valid = false;
};

struct srz_target {
  int width = 0, height = 0;
  float *d_planes = nullptr; // [z,c0,c1,c2][H][W]
  uint8_t *d_bgr8 = nullptr;
  bool pending_clear = false; // clear(Color|Depth) not yet materialised: the next draw runs with SRZ_FUSED_CLEAR
};

struct srz_frameset {
  int n_frames = 0, width = 0, height = 0;
  int shard_rank = 0, shard_world = 1;
  uint32_t n_bands = 0, n_local_bands = 0, bands_per_rank = 0, local_rows = 0;
  uint32_t max_tris = 0;
  uint64_t total_tris = 0, total_lights = 0;
  std::vector<FrameDesc> h_frames;
  std::vector<BatchDesc> h_batches;
  std::vector<ShadeDescG> h_sdesc;
  FrameDesc *d_frames = nullptr;
  srz_tri *d_tris = nullptr;     // the triangle stream as uploaded
  float *d_tri_pos = nullptr;    // dense copy of its positions (9 floats per triangle); null: srz_draw's one-frame set, re-uploaded per call
  bool tris_aos = false;
  BBox *d_bbox = nullptr;
  uint16_t *d_tri_batch = nullptr;
  BatchDesc *d_batches = nullptr;
  srz_light *d_lights = nullptr;
  // per-tile triangle lists: records in a pool of n_sub sub-pools (srz_device.h, RenderArgs); every render reports what
  // it asked of each sub-pool (h_pool_heads: mapped host memory the device stores into), and a render that finds the previous demand
  // above the capacity grows the pool first — so the memory is O(triangle-tile pairs), not O(bands x triangles)
  uint32_t *d_pool = nullptr; // tile lists: triangle indices
  uint32_t pool_sub_cap = 0, pool_n_sub = 1;
  bool pool_sized = false; // the first render has sized the pool by its own demand (render_impl)
  uint32_t *d_pool_heads = nullptr, *h_pool_heads = nullptr;
  static constexpr int DEMAND_PARTS = 8; // h_pool_heads holds one copy of the allocators' lines per sub-batch of a large render
  uint2 *d_tile_info = nullptr;
  uint32_t *d_slow_list = nullptr, *d_slow_count = nullptr;
  uint4 *d_redo_list = nullptr; // (its counter is d_slow_count[1])
  uint32_t fast_mask = 0;   // bit NL (+ 8 with BUMP / DISPLACEMENT batches, + 16 with a non-integer exponent): some frame is shaded by that FAST build of k_shade (classify_frames)
  bool any_generic = true;  // some frame needs the generic build
  uint32_t *d_vis = nullptr, *d_work_count = nullptr, *d_chunk_rows = nullptr; // d_vis: the per-tile pixel lists (srz_device.h)
  uint4 *d_worklist = nullptr;
  // k_shade's work lists are stored only for the build kinds some frame of the set needs (+ the generic one): slot of kind k =
  // (kind_slots >> 4k) & 15 (RenderArgs::kind_slots); a re-classification that brings a new kind in grows the storage
  uint64_t kind_slots = 0;
  uint32_t kind_mask = 0, n_kind_slots = 0;
  bool approx_shade = false; // SRZ_OPT_APPROX_SHADE at creation: frames of 1..4 lights without BUMP / DISPLACEMENT batches are shaded by the ApproxMath builds
  bool no_packed = false; // SRZ_NO_PACKED (tests): no frame is FD_PACKED — 32-bit owner ids by triangle index, no staged triangles
  uint32_t *d_band_desc = nullptr; // the band sort of k_setup / k_chunks (srz_device.h, GROUP_TRIS): descriptors [group][local band]
  uint2 *d_band_ent = nullptr;     // and entries [group][ENT_PER_GROUP]
  uint64_t total_groups = 0;
  ShadeDescG *d_sdesc = nullptr;
  DrawDesc *d_draws = nullptr; // device vertex stage (srz_sceneset_create), else null
  // scenesets keep everything srz_sceneset_update rewrites in ONE device block [FrameDesc | lights | DrawDesc] that is
  // refreshed by a single asynchronous copy from a small ring of pinned staging buffers (no stream sync per frame)
  static constexpr int STAGE_RING = 4;
  uint8_t *d_dyn = nullptr;
  size_t dyn_bytes = 0, dyn_lights_off = 0, dyn_draws_off = 0;
  uint8_t *h_stage[STAGE_RING] = {};
  hipEvent_t stage_ev[STAGE_RING] = {};
  bool stage_busy[STAGE_RING] = {};
  unsigned stage_next = 0;
  std::vector<DrawDesc> h_draws;
  std::vector<int> h_draw_mesh;
  uint32_t n_draws = 0, max_faces = 0;
  uint64_t sdesc_version = 0;
  uint32_t tiles_x = 0, max_tiles = 0;
  bool have_stats = false;
  bool update_failed = false; // an update re-classified the frames but could not get the work lists they need: renders are refused
  // Grid of the side-stream clear (launch_clear).  Its best size depends on what the clear runs beside — about 96 workgroups on configs 2
  // and 3, 256 on config 4, 160 on config 5, with 4 .. 8 % of a step between the best and the worst of them — so a set MEASURES it, on
  // the device (srz_device.h, ClearCtl / k_clear_tune): from render CLEAR_TUNE_SKIP on the clear is launched with CLEAR_GRID_MAX workgroups
  // of which the device-side state says how many take part, and a one-thread kernel ends each of the next <= 18 renders; the decision
  // arrives in a word of mapped host memory, and from the render that finds it there the host launches exactly that grid.  The pixels are
  // the same bits under every grid.  SRZ_CLEAR_WGS fixes the grid, SRZ_CLEAR_TUNE=0 leaves it at 96, SRZ_CLEAR_TRACE=1 prints the
  // measurement.  Every CLEAR_TUNE_AGAIN renders the set measures again (the scene of a sceneset changes under it).  A new set whose shape
  // another set of the ctx has measured (srz_ctx::clear_memo) starts with that set's grid and measures only then.
  static constexpr int CLEAR_TUNE_SKIP = 6, CLEAR_TUNE_AGAIN = 4096;
  struct ClearTune {
    uint32_t wgs = CLEAR_GRID_DEFAULT; // the grid in use outside the measurement
    bool done = false;  // the host has seen the decision of the current measurement
    int renders = 0;    // renders of the set enqueued so far (counted until the measurement's last render)
    int since = 0;      // renders since the last decision
    ClearCtl *d_ctl = nullptr;  // device-side state
    uint32_t *h_wgs = nullptr;  // mapped host memory: 0 until k_clear_tune has decided
  } clear_tune;
  srz_stats stats{};
};

namespace {

// `stream` arguments of the C ABI: NULL = the ctx's own (non-blocking) stream, SRZ_STREAM_NULL = HIP's null stream, else a
// hipStream_t
hipStream_t pick_stream(const srz_ctx *ctx, void *stream) {
  if (!stream) return ctx->stream;
  return stream == SRZ_STREAM_NULL ? (hipStream_t) nullptr : (hipStream_t)stream;
}

int fail(srz_ctx *ctx, int code, const std::string &msg) {
  if (ctx)
    ctx->err = msg;
  else
    g_create_error = msg;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                                             \
  do {                                                                                                                 \
    hipError_t e_ = (expr);                                                                                            \
    if (e_ != hipSuccess)                                                                                              \
      return fail(ctx, e_ == hipErrorOutOfMemory ? SRZ_E_NOMEM : SRZ_E_NODEVICE,                                      \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                                                  \
  } while (0)

void shard_layout(int height, int rank, int world, uint32_t &n_bands, uint32_t &n_local, uint32_t &per_rank,
                  uint32_t &local_rows) {
  n_bands = (uint32_t)((height + BAND - 1) / BAND);
  // (one band of every full group of `world` bands, and of the last, partial group if the rotation puts this rank inside it: band_of)
  n_local = n_bands / (uint32_t)world;
  if ((uint32_t)band_of((int)n_local, rank, world) < n_bands) ++n_local;
  per_rank = (n_bands + (uint32_t)world - 1) / (uint32_t)world;
  local_rows = world == 1 ? (uint32_t)height : per_rank * BAND;
}

// Which frames the FAST builds of k_shade can shade: 1..4 lights (Shader::p is 150 as the reference ships it, src/Shader.cpp:10;
// any light count and exponent are legal there, src/Shader.cpp:192-386) — every shader type qualifies.  An integer exponent
// 0 <= p <= 256 takes the builds with the exact multiplication chains, a non-integer exponent in (0, 4096] the builds whose power
// is pow_fast (FD_GENPOW; not combined with BUMP / DISPLACEMENT batches), every other exponent the generic build.  Sets FD_FAST_SHADE + the
// light count in the host copies of the descriptors.
// (lights: the host copy of the set's lights, indexed by FrameDesc::light_off — or, `one_frame`, that one frame's own array)
void classify_frames(srz_frameset *fs, const srz_light *lights, bool one_frame = false) {
  fs->fast_mask = 0, fs->any_generic = false;
  for (FrameDesc &d : fs->h_frames) {
    // FD_GREY: three bit-equal channels in ka, ks and every light's intensity (the shaders then compute a PHONG pixel's channel once)
    bool grey = lights != nullptr || d.n_lights == 0;
    grey = grey && std::memcmp(&d.ka[0], &d.ka[1], 4) == 0 && std::memcmp(&d.ka[1], &d.ka[2], 4) == 0 &&
           std::memcmp(&d.ks[0], &d.ks[1], 4) == 0 && std::memcmp(&d.ks[1], &d.ks[2], 4) == 0;
    for (uint32_t l = 0; grey && l < d.n_lights; ++l) {
      const srz_light &L = lights[(one_frame ? 0u : d.light_off) + l];
      grey = std::memcmp(&L.intensity[0], &L.intensity[1], 4) == 0 && std::memcmp(&L.intensity[1], &L.intensity[2], 4) == 0;
    }
    const bool intpow = d.p >= 0.0f && d.p <= 256.0f && d.p == std::trunc(d.p);
    bool bumpy = false;
    for (uint32_t b = 0; b < d.n_batches; ++b) {
      const int sh = fs->h_batches[d.batch_off + b].shader;
      bumpy = bumpy || sh == SRZ_SHADER_BUMP || sh == SRZ_SHADER_DISPLACEMENT;
    }
    const bool fracpow = d.p > 0.0f && d.p <= 4096.0f && d.p != std::trunc(d.p); // (pow_fast's domain)
    // the tolerance mode (SRZ_OPT_APPROX_SHADE): its builds take any finite exponent >= 0 (exp2(p log2 x)) through the plain kinds'
    // work lists; frames they do not cover keep the exact generic build
    const bool approx = fs->approx_shade && d.n_lights >= 1u && d.n_lights <= 4u && !bumpy && d.p >= 0.0f && std::isfinite(d.p);
    const bool fast = approx || (!fs->approx_shade && d.n_lights >= 1u && d.n_lights <= 4u && (intpow || (fracpow && !bumpy)));
    const bool plain = approx || intpow;
    d.flags = (d.flags & ~(FD_FAST_SHADE | FD_BUMPY | FD_GENPOW | FD_PACKED | FD_GREY | (7u << FD_NL_SHIFT))) | (grey ? FD_GREY : 0u) |
              (fast ? (FD_FAST_SHADE | (d.n_lights << FD_NL_SHIFT) | (bumpy ? FD_BUMPY : 0u) | (plain ? 0u : FD_GENPOW)) : 0u) |
              ((d.n_tris < PACK_IDX_MASK && d.n_batches <= PACK_MAX_BATCHES && !fs->no_packed) ? FD_PACKED : 0u);
    if (fast)
      fs->fast_mask |= 1u << (d.n_lights + (bumpy ? 8u : 0u) + (plain ? 0u : 16u));
    else
      fs->any_generic = true;
  }
}

// the build kinds (srz_device.h: SHADE_KIND_*) the classified frames send tiles to, as a bit mask; the generic kind always (counting
// runs and SRZ_ORDERED... send everything there)
uint32_t kinds_needed(const srz_frameset *fs) {
  uint32_t m = 1u << SHADE_KIND_GENERIC;
  for (uint32_t nl = 1; nl <= 4; ++nl) {
    if (fs->fast_mask & (1u << nl)) m |= 1u << (nl - 1u);
    if (fs->fast_mask & (1u << (8u + nl))) m |= 1u << (nl - 1u + 4u);
    if (fs->fast_mask & (1u << (16u + nl))) m |= 1u << (nl - 1u + 8u);
  }
  return m;
}
// (re)allocates the work-list storage when the set needs a kind that has no slot yet; hipSuccess when nothing had to change
hipError_t ensure_worklists(srz_frameset *fs) {
  const uint32_t need = kinds_needed(fs) | fs->kind_mask;
  if (need == fs->kind_mask && fs->d_worklist) return hipSuccess;
  uint64_t slots = 0;
  uint32_t n = 0;
  for (uint32_t k = 0; k <= SHADE_KIND_GENERIC; ++k)
    if (need & (1u << k)) slots |= (uint64_t)(n++) << (4u * k);
  const size_t cap = (size_t)(fs->n_frames < 8 ? fs->n_frames : (fs->n_frames + 7) / 8) * fs->n_local_bands * fs->tiles_x;
  // the new storage first, the swap only on success: a failed allocation leaves the set exactly as it was (old lists, old slots) —
  // callers that had already re-classified the frames for a kind without a slot roll that back (srz_sceneset_update, srz_draw)
  uint4 *nw = nullptr;
  const hipError_t e = hipMalloc((void **)&nw, std::max<size_t>(sizeof(uint4) * 8u * n * cap, 256));
  if (e != hipSuccess) return e;
  if (fs->d_worklist) { // renders in flight may still be walking the old lists
    (void)hipDeviceSynchronize();
    (void)hipFree(fs->d_worklist);
  }
  fs->d_worklist = nw, fs->kind_mask = need, fs->kind_slots = slots, fs->n_kind_slots = n;
  return hipSuccess;
}

void free_frameset_buffers(srz_frameset *fs) {
  for (int i = 0; i < srz_frameset::STAGE_RING; ++i) {
    if (fs->h_stage[i]) (void)hipHostFree(fs->h_stage[i]);
    if (fs->stage_ev[i]) (void)hipEventDestroy(fs->stage_ev[i]);
  }
  if (fs->d_dyn) { // d_frames / d_lights / d_draws point into the block
    (void)hipFree(fs->d_dyn);
    fs->d_frames = nullptr, fs->d_lights = nullptr, fs->d_draws = nullptr;
  }
  (void)hipFree(fs->d_frames);
  (void)hipFree(fs->d_tris);
  (void)hipFree(fs->d_tri_pos);
  (void)hipFree(fs->d_bbox);
  (void)hipFree(fs->d_tri_batch);
  (void)hipFree(fs->d_batches);
  (void)hipFree(fs->d_lights);
  (void)hipFree(fs->d_pool);
  (void)hipFree(fs->d_pool_heads);
  if (fs->h_pool_heads) (void)hipHostFree(fs->h_pool_heads);
  if (fs->clear_tune.h_wgs) (void)hipHostFree(fs->clear_tune.h_wgs), fs->clear_tune.h_wgs = nullptr;
  if (fs->clear_tune.d_ctl) (void)hipFree(fs->clear_tune.d_ctl), fs->clear_tune.d_ctl = nullptr;
  (void)hipFree(fs->d_tile_info);
  (void)hipFree(fs->d_slow_list);
  (void)hipFree(fs->d_slow_count);
  (void)hipFree(fs->d_redo_list);
  (void)hipFree(fs->d_vis);
  (void)hipFree(fs->d_worklist);
  (void)hipFree(fs->d_work_count);
  (void)hipFree(fs->d_chunk_rows);
  (void)hipFree(fs->d_band_desc);
  (void)hipFree(fs->d_band_ent);
  (void)hipFree(fs->d_sdesc);
  (void)hipFree(fs->d_draws);
}

RenderArgs make_args(const srz_ctx *ctx, const srz_frameset *fs, float *d_out, uint32_t flags_or) {
  RenderArgs a{};
  a.frames = fs->d_frames;
  a.tris = fs->d_tris;
  a.tri_pos = fs->d_tri_pos ? fs->d_tri_pos : reinterpret_cast<const float *>(fs->d_tris);
  a.pos_stride = fs->d_tri_pos ? TRI_POS_F : TRI_AOS_F;
  a.bbox = fs->d_bbox;
  a.chunk_rows = fs->d_chunk_rows;
  a.band_desc = fs->d_band_desc;
  a.band_ent = fs->d_band_ent;
  a.tri_batch = fs->d_tri_batch;
  a.batches = fs->d_batches;
  a.lights = fs->d_lights;
  a.tex = ctx->d_tex;
  a.pool = fs->d_pool;
  a.pool_heads = fs->d_pool_heads;
  a.pool_demand = fs->h_pool_heads; // (hipHostMallocMapped: the same address on the device)
  a.pool_sub_cap = fs->pool_sub_cap;
  a.pool_sub_mask = fs->pool_n_sub - 1u;
  a.tile_info = fs->d_tile_info;
  a.slow_list = fs->d_slow_list;
  a.slow_count = fs->d_slow_count;
  a.redo_list = fs->d_redo_list;
  a.redo_count = fs->d_slow_count + 1;
  a.force_ordered = 0, a.force_generic = 0, a.any_generic = fs->any_generic ? 1u : 0u;
  a.sdesc = fs->d_sdesc;
  a.vis = fs->d_vis;
  a.worklist = fs->d_worklist;
  a.kind_slots = fs->kind_slots;
  a.work_count = fs->d_work_count;
  // (a list holds the tiles of every 8th frame; of fewer than 8 frames: any of them)
  a.work_cap = (uint32_t)(fs->n_frames < 8 ? fs->n_frames : (fs->n_frames + 7) / 8) * fs->n_local_bands * fs->tiles_x;
  a.tiles_x = fs->tiles_x;
  a.n_local_bands = fs->n_local_bands;
  a.n_frames = (uint32_t)fs->n_frames;
  a.out = d_out;
  a.local_rows = fs->local_rows;
  a.frame_stride = 4ull * fs->local_rows * (uint64_t)fs->width;
  a.shard_rank = fs->shard_rank;
  a.shard_world = fs->shard_world;
  a.flags_or = flags_or;
  a.stats = ctx->d_stats;
  return a;
}

int get_events(srz_ctx *ctx, EventPair &ep) {
  if (!ctx->ev_pool.empty()) {
    ep = ctx->ev_pool.back();
    ctx->ev_pool.pop_back();
    return SRZ_OK;
  }
  HIP_TRY(ctx, hipEventCreate(&ep.t0));
  HIP_TRY(ctx, hipEventCreate(&ep.t1));
  HIP_TRY(ctx, hipEventCreate(&ep.t2));
  HIP_TRY(ctx, hipEventCreate(&ep.t3));
  return SRZ_OK;
}

int collect_events(srz_ctx *ctx) {
  for (auto &ep : ctx->ev_used) {
    HIP_TRY(ctx, hipEventSynchronize(ep.t3));
    float b = 0.f, r = 0.f, sh = 0.f, t = 0.f;
    if (ep.detailed) {
      HIP_TRY(ctx, hipEventElapsedTime(&b, ep.t0, ep.t1));
      HIP_TRY(ctx, hipEventElapsedTime(&r, ep.t1, ep.t2));
      HIP_TRY(ctx, hipEventElapsedTime(&sh, ep.t2, ep.t3));
    }
    HIP_TRY(ctx, hipEventElapsedTime(&t, ep.t0, ep.t3));
    ctx->acc_ms[0] += b, ctx->acc_ms[1] += r, ctx->acc_ms[2] += sh, ctx->acc_ms[3] += t;
    ctx->acc_launches++;
    if (ctx->acc_samples.size() < 65536) ctx->acc_samples.push_back(t);
    if (!ctx->span_open) { // keep the first render's start: swap it for the context's spare event
      if (!ctx->span_t0) HIP_TRY(ctx, hipEventCreate(&ctx->span_t0));
      std::swap(ctx->span_t0, ep.t0);
      ctx->span_open = true;
    }
    float sp = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&sp, ctx->span_t0, ep.t3));
    ctx->span_ms = std::max(ctx->span_ms, (double)sp);
    ctx->ev_pool.push_back(ep);
  }
  ctx->ev_used.clear();
  return SRZ_OK;
}

// setup → bands → raster for every frame of the set, asynchronously on `s`
// (one_frame_scratch: a counting run whose pixels nobody reads — every frame writes the SAME one-frame buffer)
// (size_only: the creation-time pass of srz_frameset_create / srz_sceneset_create — setup + binning of every sub-batch and the sizing
// of the tile-list pool by their demand, nothing rasterised, no texture needed yet)
// (f_begin, f_count: only frames [f_begin, f_begin + f_count) of the set — srz_draw_batch renders a set in pieces so that the
// read-back of one piece runs under the render of the next; d_out is the whole set's buffer either way)
int render_impl(srz_ctx *ctx, srz_frameset *fs, float *d_out, uint32_t flags_or, hipStream_t s, bool stats, bool one_frame_scratch = false,
                bool size_only = false, int f_begin = 0, int f_count = -1) {
  if (fs->shard_rank != ctx->shard_rank || fs->shard_world != ctx->shard_world)
    return fail(ctx, SRZ_E_INVALID, "frameset was created under a different shard (call srz_set_shard before srz_frameset_create)");
  if (fs->update_failed) return fail(ctx, SRZ_E_NOMEM, "the last update of this set failed (out of memory): update it again or destroy it");
  for (const BatchDesc &b : fs->h_batches) {
    if (size_only) break;
    bool needs = b.shader == SRZ_SHADER_TEXTURE || b.shader == SRZ_SHADER_DISPLACEMENT || b.shader == SRZ_SHADER_BUMP;
    if (needs && (b.tex_id < 0 || b.tex_id >= MAX_TEX || !ctx->h_tex[b.tex_id].bgrx))
      return fail(ctx, SRZ_E_TEXTURE, "batch uses texture slot " + std::to_string(b.tex_id) + " which was never uploaded");
  }
  if (!size_only && fs->sdesc_version != ctx->tex_version && !fs->h_batches.empty()) { // (re)resolve batch → shader/texture
    std::vector<ShadeDescG> &h = fs->h_sdesc; // (owned by the set: the asynchronous copy below may read it after we return)
    h.resize(fs->h_batches.size());
    for (size_t i = 0; i < h.size(); ++i) {
      const BatchDesc &b = fs->h_batches[i];
      bool needs = b.shader == SRZ_SHADER_TEXTURE || b.shader == SRZ_SHADER_DISPLACEMENT || b.shader == SRZ_SHADER_BUMP;
      h[i].shader = b.shader, h[i]._pad = 0;
      h[i].tw = needs ? ctx->h_tex[b.tex_id].w : 1, h[i].th = needs ? ctx->h_tex[b.tex_id].h : 1;
      h[i].tex = needs ? ctx->h_tex[b.tex_id].bgrx : nullptr;
    }
    // on the launch stream: ordered after the renders already submitted there, before this one
    HIP_TRY(ctx, hipMemcpyAsync(fs->d_sdesc, h.data(), sizeof(ShadeDescG) * h.size(), hipMemcpyHostToDevice, s));
    fs->sdesc_version = ctx->tex_version;
  }
  // the record pool follows what the previous renders asked for: growing is rare and the one place where a render waits
  // for the device
  { // (h_pool_heads is pinned host memory the rasteriser's first workgroup stores into: whatever it holds is a demand some
    // finished or running render of this set really had — a stale value only delays the growth by a render)
    uint32_t need = 0;
    for (uint32_t i = 0; i < fs->pool_n_sub * (uint32_t)srz_frameset::DEMAND_PARTS; ++i) { // (every sub-batch's region)
      const uint32_t v = static_cast<volatile uint32_t *>(fs->h_pool_heads)[((i / fs->pool_n_sub) * 64u + i % fs->pool_n_sub) * CNT_STRIDE];
      if (v > need) need = v;
    }
    if (need > fs->pool_sub_cap) {
      HIP_TRY(ctx, hipDeviceSynchronize());
      const uint64_t cap = (uint64_t)need + need / 4u + 64u;
      if (cap * fs->pool_n_sub >= 0xffffffffull) return fail(ctx, SRZ_E_NOMEM, "tile lists exceed 2^32 records; split the batch");
      uint32_t *p = nullptr;
      HIP_TRY(ctx, hipMalloc(&p, sizeof(uint32_t) * cap * fs->pool_n_sub));
      (void)hipFree(fs->d_pool);
      fs->d_pool = p, fs->pool_sub_cap = (uint32_t)cap;
    }
  }
  RenderArgs a = make_args(ctx, fs, d_out, flags_or);
  if (one_frame_scratch) a.frame_stride = 0;
  a.force_ordered = a.force_generic = stats ? 1u : 0u; // the counters are those of the reference's ordered walk
  a.any_ordered = ((flags_or & SRZ_ORDERED_RASTER) != 0 ||
                   std::any_of(fs->h_frames.begin(), fs->h_frames.end(), [](const FrameDesc &f) { return (f.flags & SRZ_ORDERED_RASTER) != 0; }))
                      ? 1u : 0u;
  EventPair ep{};
  const bool timed = ctx->timing != 0 && !stats && !size_only && ctx->ev_used.size() < 65536, detailed = timed && ctx->timing >= 2;
  if (timed) {
    int rc = get_events(ctx, ep);
    if (rc) return rc;
    ep.detailed = detailed;
    HIP_TRY(ctx, hipEventRecord(ep.t0, s));
  }
  if (stats) HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, ST_COUNT * sizeof(unsigned long long), s));
  if (fs->max_tris == 0) { // (else: k_setup resets the per-render counters)
    HIP_TRY(ctx, hipMemsetAsync(fs->d_work_count, 0, sizeof(uint32_t) * CNT_STRIDE * N_WORK_LISTS, s));
    HIP_TRY(ctx, hipMemsetAsync(fs->d_pool_heads, 0, sizeof(uint32_t) * CNT_STRIDE * fs->pool_n_sub, s));
    HIP_TRY(ctx, hipMemsetAsync(fs->d_slow_count, 0, 2 * sizeof(uint32_t), s));
  }
  const bool turns = !stats && !size_only && fs->max_tiles >= 8192 && !ctx->env_no_turns; // (batches; small jobs are launch-bound and gain nothing)
  if (turns) {
    if (!ctx->ev_raster[0])
      for (int i = 0; i < srz_ctx::EV_RING; ++i) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_raster[i], hipEventDisableTiming));
    if (ctx->raster_valid && ctx->raster_last_stream != s) {
      HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_raster[(ctx->raster_next + srz_ctx::EV_RING - 1) % srz_ctx::EV_RING], 0));
      a.other_streams = 1u;
    }
  }
  // vertex stage on the device; outside counting runs it does the triangles' setup too (cull + bounding box from the registers
  // that hold the transformed triangle), and k_chunks replaces k_setup below
  const bool vertex_setup = fs->d_draws != nullptr && !stats;
  if (fs->d_draws) launch_vertex(fs->d_draws, fs->n_draws, fs->max_faces, fs->d_tris, fs->d_tri_pos, fs->d_frames, vertex_setup ? fs->d_bbox : nullptr, s);
  // fused clear of the tiles no bbox reaches: beside k_raster on a second stream (batches), or in the rasteriser (small jobs)
  const bool any_fused = (flags_or & SRZ_FUSED_CLEAR) != 0 ||
                         std::any_of(fs->h_frames.begin(), fs->h_frames.end(), [](const FrameDesc &f) { return (f.flags & SRZ_FUSED_CLEAR) != 0; });
  const bool side = any_fused && fs->max_tiles >= 8192 && !size_only;
  if (side && !ctx->stream2) {
    // The clear must run BESIDE the launch stream, so it may not share a hardware queue with it: HIP deals its streams
    // round-robin onto a few hardware queues (seen: the caller's stream and this one on the same queue — the clear then ran
    // in front of k_raster instead of beside it, +20 % per render).  Streams of another priority live on queues of their own;
    // the clear is throttled by its grid size, not by priority, so the highest one costs the rasteriser nothing.
    int prio_least = 0, prio_greatest = 0;
    HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, prio_greatest));
    for (int i = 0; i < srz_ctx::EV_RING; ++i) {
      HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fork[i], hipEventDisableTiming));
      HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming));
    }
  }
  if (!side && any_fused) a.clear_in_raster = 1u; // (small job: the rasteriser's own waves clear the tiles no bbox reaches)
  // A LARGE set is rendered as sub-batches of whole frames one after the other on the same stream: what k_raster leaves for
  // k_shade (depth, owner ids) and k_bin for k_raster (records) stays in the 256 MiB Infinity Cache only while the frames in
  // flight are few — config 2 at 128 / 256 / 384 / 512 frames per launch set: 0.539 / 0.545 / 0.499 / 0.490 of the roofline (round 3; round 6,
  // beside the per-plane clear, 1024 frames on one stream in pieces of 176 / 208 / 256 / 344 / 512: 0.633 / 0.644 / 0.656 / 0.653 / 0.632).
  // A sub-batch is a view: every per-frame array from its first frame on; counters, record pool and work lists are shared
  // (k_setup resets them, and stream order keeps one sub-batch's kernels behind the previous one's).  Counting runs and the
  // per-kernel timing mode render in one piece.
  const int n_all = f_count < 0 ? fs->n_frames : f_count;
  int chunk = n_all;
  // The sub-batch size is 256 frames' worth of 1024^2 (≈ 262 k tiles: the measured sweet spot; 192 until round 6), in frames of THIS
  // set — a rank of an 8-GPU job holds an eighth of every frame and takes 2048 of them at a time; frames so large that fewer than 96
  // make a sub-batch (2048^2 and up) are left alone: nothing of theirs fits the cache either way (config 4, 256 frames, in pieces of
  // 64 / 128 / 256: 0.425 / 0.422 / 0.422).
  const int sub_env = ctx->env_sub_batch; // (tuning / tests: frames per sub-batch)
  const size_t tiles_per_frame = std::max<size_t>((size_t)fs->n_local_bands * fs->tiles_x, 1);
  const int sub = sub_env > 0 ? sub_env : (int)std::min<size_t>((256u * 1024u / tiles_per_frame + 7u) / 8u * 8u, 1u << 20);
  if (sub >= 96 && n_all >= sub + sub / 4 && !stats && !detailed) {
    const int parts = (n_all + sub - 1) / sub;
    chunk = ((n_all + parts - 1) / parts + 7) / 8 * 8;
  }
  const size_t tpf = (size_t)fs->n_local_bands * fs->tiles_x;
  // the side clear's grid: measured per set, on the device (srz_frameset::ClearTune)
  uint32_t clear_wgs = ctx->env_clear_wgs ? ctx->env_clear_wgs : fs->clear_tune.wgs;
  bool tune_stamp = false;
  if (side && !ctx->env_clear_wgs && !ctx->env_no_clear_tune && !stats && f_count < 0 && fs->clear_tune.d_ctl) {
    srz_frameset::ClearTune &ct = fs->clear_tune;
    if (ct.done && !detailed && ++ct.since >= srz_frameset::CLEAR_TUNE_AGAIN) {
      // what the clear runs beside may have changed (srz_sceneset_update): measure again (<= 18 of 4096 renders)
      ct.done = false, ct.since = 0, ct.renders = srz_frameset::CLEAR_TUNE_SKIP;
      *static_cast<volatile uint32_t *>(ct.h_wgs) = 0u; // (no k_clear_tune is in flight: the host stopped launching them when it saw the decision)
      HIP_TRY(ctx, hipMemsetAsync(ct.d_ctl, 0, sizeof(ClearCtl), s));
    }
    // the shape of the set, as far as the clear's best grid depends on it: frame size, frames, bands, triangles, the shading builds in use
    auto memo_key = [&]() {
      uint64_t k = 0xcbf29ce484222325ull;
      for (uint64_t v : {(uint64_t)fs->width, (uint64_t)fs->height, (uint64_t)fs->n_frames, (uint64_t)fs->n_local_bands, fs->total_tris,
                         (uint64_t)fs->fast_mask, (uint64_t)fs->approx_shade})
        k = (k ^ v) * 0x100000001b3ull;
      return k;
    };
    if (!ct.done && ct.renders == 0) // (the set's first render: has a set of this shape measured before?)
      for (const auto &m : ctx->clear_memo)
        if (m.first == memo_key()) ct.wgs = clear_wgs = m.second, ct.done = true;
    if (!ct.done) {
      const uint32_t h = *static_cast<volatile uint32_t *>(ct.h_wgs);
      const int j = ct.renders - srz_frameset::CLEAR_TUNE_SKIP;
      if (h != 0u) { // decided: launch that grid from now on
        ct.wgs = clear_wgs = h, ct.done = true;
        {
          const uint64_t key = memo_key();
          auto it = std::find_if(ctx->clear_memo.begin(), ctx->clear_memo.end(), [&](const std::pair<uint64_t, uint32_t> &m) { return m.first == key; });
          if (it != ctx->clear_memo.end()) it->second = h;
          else if (ctx->clear_memo.size() < 256) ctx->clear_memo.emplace_back(key, h);
        }
        if (ctx->env_clear_trace) {
          ClearCtl c;
          if (hipMemcpy(&c, ct.d_ctl, sizeof c, hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "srz: clear grid of set %p:", (void *)fs);
            for (int i = 0; i < CLEAR_CANDS; ++i) fprintf(stderr, " %u:%.3f", CLEAR_CAND[i], c.score[i] * 1e-5f);
            fprintf(stderr, " ms (0: dropped after the first pass) -> %u\n", h);
          }
        }
      } else if (j < 0) {
        if (!detailed) ++ct.renders;
      } else if (!detailed) { // measuring (or the decision has not reached the host yet): the device says how many of the grid take part
        clear_wgs = CLEAR_GRID_MAX, a.clear_wgs_dev = &ct.d_ctl->wgs;
        if (j < CLEAR_TUNE_RENDERS) ++ct.renders, tune_stamp = true;
      } // (the per-kernel timing mode puts barriers into the stream: not a sample, not counted, and rendered with the grid in use before)
    }
  }
  int part = 0;
  for (int f0 = 0; f0 < n_all; f0 += chunk, ++part) {
    RenderArgs v = a;
    const int n = std::min(chunk, n_all - f0);
    if (n != fs->n_frames) {
      const int fa = f_begin + f0; // (first frame of this piece in the set)
      v.frames += fa, v.n_frames = (uint32_t)n;
      v.vis += (size_t)fa * tpf * ((size_t)TILE * TILE);
      v.tile_info += (size_t)fa * tpf;
      v.out += (size_t)fa * a.frame_stride;
      v.work_cap = (uint32_t)((size_t)(n < 8 ? n : (n + 7) / 8) * tpf);
    }
    const uint32_t tiles = (uint32_t)((size_t)n * tpf);
    if (vertex_setup)
      launch_chunks(v, n, fs->max_tris, s);
    else
      launch_setup(v, n, fs->max_tris, stats, s);
    launch_bin(v, n, fs->max_tris, s);
    // the record pool's demand of this (sub-)render → host (word 0 of every allocator's line; one region per sub-batch).  No
    // event: the next render reads whatever has arrived (see the growth check above).  The latency build of k_raster stores it itself.
    auto copy_demand = [&](hipStream_t cs) {
      uint32_t *dst = fs->h_pool_heads + (size_t)std::min(part, srz_frameset::DEMAND_PARTS - 1) * CNT_STRIDE * 64;
      HIP_TRY(ctx, hipMemcpyAsync(dst, fs->d_pool_heads, sizeof(uint32_t) * CNT_STRIDE * fs->pool_n_sub, hipMemcpyDeviceToHost, cs));
      return (int)SRZ_OK;
    };
    // The FIRST render of a set sizes the pool by what it needs itself: it waits for k_bin's count, grows the pool if a band
    // did not fit and bins again — a one-shot set (srz_draw_batch, the host layer's per-draw sets) has no second render that
    // could profit from the lazy growth, and would otherwise leave its overflowing bands to the ordered rasteriser.
    if (!fs->pool_sized && !stats) {
      if (int rc = copy_demand(s)) return rc;
      HIP_TRY(ctx, hipStreamSynchronize(s));
      const uint32_t *dem = fs->h_pool_heads + (size_t)std::min(part, srz_frameset::DEMAND_PARTS - 1) * CNT_STRIDE * 64;
      uint32_t need = 0;
      for (uint32_t i = 0; i < fs->pool_n_sub; ++i) {
        const uint32_t v = static_cast<const volatile uint32_t *>(dem)[i * CNT_STRIDE];
        if (v > need) need = v;
      }
      if (need > fs->pool_sub_cap) {
        HIP_TRY(ctx, hipDeviceSynchronize()); // (earlier sub-batches of this render may still be reading the old pool)
        const uint64_t cap = (uint64_t)need + need / 8u + 64u;
        if (cap * fs->pool_n_sub >= 0xffffffffull) return fail(ctx, SRZ_E_NOMEM, "tile lists exceed 2^32 records; split the batch");
        uint32_t *p = nullptr;
        HIP_TRY(ctx, hipMalloc(&p, sizeof(uint32_t) * cap * fs->pool_n_sub));
        (void)hipFree(fs->d_pool);
        fs->d_pool = p, fs->pool_sub_cap = (uint32_t)cap;
        a.pool = v.pool = p, a.pool_sub_cap = v.pool_sub_cap = (uint32_t)cap;
        HIP_TRY(ctx, hipMemsetAsync(fs->d_pool_heads, 0, sizeof(uint32_t) * CNT_STRIDE * fs->pool_n_sub, s));
        launch_bin(v, n, fs->max_tris, s);
      }
    }
    if (size_only) continue; // (the creation-time pass ends with the binning)
    if (detailed) HIP_TRY(ctx, hipEventRecord(ep.t1, s));
    unsigned ev = 0;
    // SRZ_CLEAR_AT=1 (diagnostic, A/B): the clear starts beside k_shade instead of beside k_raster
    static const bool clear_late = getenv("SRZ_CLEAR_AT") && atoi(getenv("SRZ_CLEAR_AT")) == 1;
    if (side && clear_late) launch_raster(v, n, stats, s);
    if (side) {
      ev = ctx->ev_next++ % srz_ctx::EV_RING;
      HIP_TRY(ctx, hipEventRecord(ctx->ev_fork[ev], s));
      hipStream_t side_s = ctx->stream2;
      HIP_TRY(ctx, hipStreamWaitEvent(side_s, ctx->ev_fork[ev], 0));
      launch_clear(v, tiles, true, side_s, clear_wgs);
      // (in front of the join: whatever follows on the launch stream — the next sub-batch's or render's k_setup zeroes the
      // allocators — is ordered behind this copy; it is 16 words behind a kernel that outlasts k_raster)
      if (int rc = copy_demand(side_s)) return rc;
      HIP_TRY(ctx, hipEventRecord(ctx->ev_join[ev], side_s));
    }
    if (!(side && clear_late)) launch_raster(v, n, stats, s);
    if (turns) {
      HIP_TRY(ctx, hipEventRecord(ctx->ev_raster[ctx->raster_next++ % srz_ctx::EV_RING], s));
      ctx->raster_last_stream = s, ctx->raster_valid = true;
    }
    if (detailed) HIP_TRY(ctx, hipEventRecord(ep.t2, s));
    launch_shade(v, tiles, stats, fs->fast_mask, fs->any_generic, fs->approx_shade, s);
    if (side) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_join[ev], 0));
    else if (!raster_four_waves(v))
      if (int rc = copy_demand(s)) return rc;
  }
  if (!stats) fs->pool_sized = true;
  if (tune_stamp) launch_clear_tune(fs->clear_tune.d_ctl, fs->clear_tune.h_wgs, s);
  if (timed) {
    HIP_TRY(ctx, hipEventRecord(ep.t3, s));
    ctx->ev_used.push_back(ep);
  }
  HIP_TRY(ctx, hipGetLastError());
  return SRZ_OK;
}

int read_stats(srz_ctx *ctx, hipStream_t s, srz_stats *st);

// The counters of srz_stats are those of the reference's ORDERED walk ("shaded" = fragments that pass the z-test when
// their triangle is drawn), which the order-independent rasteriser does not produce.  A draw that asks for them runs the
// counting kernels (ordered rasteriser) once more on a scratch copy of the framebuffer the draw starts from.
int stats_pass(srz_ctx *ctx, srz_frameset *fs, const float *d_start, uint32_t flags_or, hipStream_t s, srz_stats *st) {
  const size_t bytes = (size_t)fs->n_frames * 4u * fs->local_rows * (size_t)fs->width * sizeof(float);
  float *d_tmp = nullptr;
  HIP_TRY(ctx, hipMalloc(&d_tmp, bytes));
  hipError_t e = d_start ? hipMemcpyAsync(d_tmp, d_start, bytes, hipMemcpyDeviceToDevice, s) : hipSuccess;
  int rc = e == hipSuccess ? render_impl(ctx, fs, d_tmp, flags_or, s, true) : fail(ctx, SRZ_E_NODEVICE, hipGetErrorString(e));
  if (rc == SRZ_OK) rc = read_stats(ctx, s, st); // (synchronises s)
  else (void)hipStreamSynchronize(s);
  (void)hipFree(d_tmp);
  return rc;
}

int read_stats(srz_ctx *ctx, hipStream_t s, srz_stats *st) {
  unsigned long long h[ST_COUNT];
  HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  st->n_tris = h[ST_TRIS], st->n_culled = h[ST_CULLED], st->pixel_tests = h[ST_PIXEL_TESTS];
  st->fragments = h[ST_FRAGMENTS], st->shaded = h[ST_SHADED], st->visible = h[ST_VISIBLE];
  st->visible_textured = h[ST_VISIBLE_TEX];
  std::memcpy(ctx->dbg, h, sizeof h);
  return SRZ_OK;
}

} // namespace

extern "C" {

int srz_abi_version(void) { return SRZ_ABI_VERSION; }

int srz_create(srz_ctx **out, int device_id) {
  if (!out) return fail(nullptr, SRZ_E_INVALID, "srz_create: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, SRZ_E_NODEVICE,
                std::string("srz_create: no HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "count=0") +
                    "); this library has no CPU fallback");
  if (device_id < 0 || device_id >= n) return fail(nullptr, SRZ_E_INVALID, "srz_create: device_id out of range");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess)
    return fail(nullptr, SRZ_E_NODEVICE, "srz_create: hipGetDeviceProperties failed");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, SRZ_E_NODEVICE, std::string("srz_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
  srz_ctx *ctx = new (std::nothrow) srz_ctx();
  if (!ctx) return fail(nullptr, SRZ_E_NOMEM, "srz_create: out of host memory");
  ctx->device = device_id;
  ctx->env_sub_batch = getenv("SRZ_SUB_BATCH") ? atoi(getenv("SRZ_SUB_BATCH")) : 0;
  ctx->env_no_packed = getenv("SRZ_NO_PACKED") != nullptr;
  ctx->env_no_turns = getenv("SRZ_NO_TURNS") != nullptr;
  ctx->env_clear_wgs = getenv("SRZ_CLEAR_WGS") ? (uint32_t)std::max(atoi(getenv("SRZ_CLEAR_WGS")), 0) : 0u;
  ctx->env_no_clear_tune = getenv("SRZ_CLEAR_TUNE") && atoi(getenv("SRZ_CLEAR_TUNE")) == 0;
  ctx->env_clear_trace = getenv("SRZ_CLEAR_TRACE") && atoi(getenv("SRZ_CLEAR_TRACE")) != 0;
  ctx->opt_pool_lazy = getenv("SRZ_POOL_LAZY") != nullptr;
  for (int i = 0; i < MAX_TEX; ++i) ctx->h_tex[i] = TexDesc{nullptr, 0, 0}, ctx->d_texmem[i] = nullptr;
  auto bail = [&](const char *what, hipError_t err) {
    g_create_error = std::string(what) + ": " + hipGetErrorString(err);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    (void)hipFree(ctx->d_tex);
    (void)hipFree(ctx->d_stats);
    delete ctx;
    return SRZ_E_NODEVICE;
  };
  if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  if ((e = hipMalloc(&ctx->d_tex, sizeof(TexDesc) * MAX_TEX)) != hipSuccess) return bail("hipMalloc(tex table)", e);
  if ((e = hipMemset(ctx->d_tex, 0, sizeof(TexDesc) * MAX_TEX)) != hipSuccess) return bail("hipMemset", e);
  if ((e = hipMalloc(&ctx->d_stats, sizeof(unsigned long long) * ST_COUNT)) != hipSuccess) return bail("hipMalloc(stats)", e);
  *out = ctx;
  return SRZ_OK;
}

void srz_destroy(srz_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->draw_fs) srz_frameset_destroy(ctx, ctx->draw_fs);
  (void)hipFree(ctx->draw_out);
  (void)hipFree(ctx->batch_out);
  if (ctx->span_t0) (void)hipEventDestroy(ctx->span_t0);
  for (auto &ep : ctx->ev_used) ctx->ev_pool.push_back(ep);
  for (auto &ep : ctx->ev_pool) (void)hipEventDestroy(ep.t0), (void)hipEventDestroy(ep.t1), (void)hipEventDestroy(ep.t2), (void)hipEventDestroy(ep.t3);
  for (int i = 0; i < MAX_TEX; ++i) (void)hipFree(ctx->d_texmem[i]);
  for (int i = 0; i < MAX_MESH; ++i) (void)hipFree(ctx->mesh[i].d_verts), (void)hipFree(ctx->mesh[i].d_faces);
  (void)hipFree(ctx->d_tex);
  (void)hipFree(ctx->d_stats);
  if (ctx->stream2) {
    (void)hipStreamSynchronize(ctx->stream2);
    (void)hipStreamDestroy(ctx->stream2);
    for (int i = 0; i < srz_ctx::EV_RING; ++i) (void)hipEventDestroy(ctx->ev_fork[i]), (void)hipEventDestroy(ctx->ev_join[i]);
  }
  if (ctx->stream3) {
    (void)hipStreamSynchronize(ctx->stream3);
    (void)hipStreamDestroy(ctx->stream3);
    for (int i = 0; i < srz_ctx::EV_RING; ++i) (void)hipEventDestroy(ctx->ev_piece[i]);
  }
  if (ctx->ev_raster[0]) {
    for (int i = 0; i < srz_ctx::EV_RING; ++i) (void)hipEventDestroy(ctx->ev_raster[i]);
  }
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *srz_last_error(const srz_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int srz_set_option(srz_ctx *ctx, int option, int value) {
  if (!ctx) return SRZ_E_INVALID;
  if (option == SRZ_OPT_POOL_LAZY)
    ctx->opt_pool_lazy = value != 0;
  else if (option == SRZ_OPT_APPROX_SHADE)
    ctx->opt_approx_shade = value != 0;
  else
    return fail(ctx, SRZ_E_INVALID, "srz_set_option: unknown option " + std::to_string(option));
  return SRZ_OK;
}

int srz_set_shard(srz_ctx *ctx, int rank, int world) {
  if (!ctx) return SRZ_E_INVALID;
  if (world < 1 || rank < 0 || rank >= world) return fail(ctx, SRZ_E_INVALID, "srz_set_shard: need 0 <= rank < world");
  ctx->shard_rank = rank, ctx->shard_world = world;
  return SRZ_OK;
}

int srz_texture_upload(srz_ctx *ctx, int tex_id, const uint8_t *bgr, int w, int h, int row_stride) {
  if (!ctx) return SRZ_E_INVALID;
  if (tex_id < 0 || tex_id >= MAX_TEX || !bgr || w <= 0 || h <= 0 || row_stride < 3 * w || w > 32768 || h > 32768)
    return fail(ctx, SRZ_E_INVALID, "srz_texture_upload: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint8_t *d_raw = nullptr;
  uint32_t *d_px = nullptr;
  size_t raw = (size_t)row_stride * h;
  HIP_TRY(ctx, hipMalloc(&d_raw, raw));
  hipError_t e = hipMalloc(&d_px, (size_t)w * h * 4);
  if (e != hipSuccess) {
    (void)hipFree(d_raw);
    return fail(ctx, SRZ_E_NOMEM, "srz_texture_upload: hipMalloc failed");
  }
  e = hipMemcpyAsync(d_raw, bgr, raw, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    launch_tex_convert(d_raw, w, h, row_stride, d_px, ctx->stream);
    e = hipStreamSynchronize(ctx->stream);
  }
  (void)hipFree(d_raw);
  if (e != hipSuccess) {
    (void)hipFree(d_px);
    return fail(ctx, SRZ_E_NODEVICE, std::string("srz_texture_upload: ") + hipGetErrorString(e));
  }
  (void)hipFree(ctx->d_texmem[tex_id]);
  ctx->d_texmem[tex_id] = d_px;
  ctx->h_tex[tex_id] = TexDesc{d_px, w, h};
  ctx->tex_version++;
  HIP_TRY(ctx, hipMemcpy(ctx->d_tex + tex_id, &ctx->h_tex[tex_id], sizeof(TexDesc), hipMemcpyHostToDevice));
  return SRZ_OK;
}

static int build_frameset(srz_ctx *ctx, const srz_frame *frames, int n_frames, srz_frameset **out, bool copy_tris, bool tris_aos = false) {
  if (!ctx) return SRZ_E_INVALID;
  if (!out) return fail(ctx, SRZ_E_INVALID, "srz_frameset_create: out is NULL");
  *out = nullptr;
  if (!frames || n_frames <= 0) return fail(ctx, SRZ_E_INVALID, "srz_frameset_create: no frames");
  const int W = frames[0].width, H = frames[0].height;
  if (W <= 0 || H <= 0 || W > 32767 || H > 32767) return fail(ctx, SRZ_E_INVALID, "srz_frameset_create: width/height must be in 1..32767");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  srz_frameset *fs = new (std::nothrow) srz_frameset();
  if (!fs) return fail(ctx, SRZ_E_NOMEM, "srz_frameset_create: out of host memory");
  fs->n_frames = n_frames, fs->width = W, fs->height = H;
  fs->shard_rank = ctx->shard_rank, fs->shard_world = ctx->shard_world;
  shard_layout(H, fs->shard_rank, fs->shard_world, fs->n_bands, fs->n_local_bands, fs->bands_per_rank, fs->local_rows);
  fs->tiles_x = (uint32_t)((W + TILE - 1) / TILE);
  if ((uint64_t)((n_frames + 7) / 8 * 8) * fs->n_local_bands * fs->tiles_x >= 0x7fffffffull) {
    delete fs;
    return fail(ctx, SRZ_E_INVALID, "srz_frameset_create: frames x tiles exceeds the launch grid limit; split the batch");
  }
  uint64_t tri_off = 0, light_off = 0, batch_off = 0, group_off = 0;
  for (int f = 0; f < n_frames; ++f) {
    const srz_frame &fr = frames[f];
    auto bad = [&](const char *m) {
      delete fs;
      return fail(ctx, SRZ_E_INVALID, std::string("srz_frameset_create: frame ") + std::to_string(f) + ": " + m);
    };
    if (fr.width != W || fr.height != H) return bad("all frames of a set must share width/height");
    if ((fr.n_lights && !fr.lights) || (fr.n_batches && !fr.batches)) return bad("null lights/batches");
    if (fr.n_batches > 65535) return bad("more than 65535 batches");
    FrameDesc d{};
    d.width = W, d.height = H;
    std::memcpy(d.eye, fr.eye, sizeof d.eye);
    std::memcpy(d.ka, fr.ka, sizeof d.ka);
    std::memcpy(d.ks, fr.ks, sizeof d.ks);
    d.p = fr.p, d.kh = fr.kh, d.kn = fr.kn;
    d.n_lights = fr.n_lights, d.light_off = (uint32_t)light_off;
    d.tri_off = (uint32_t)tri_off, d.n_batches = fr.n_batches, d.batch_off = (uint32_t)batch_off;
    d.flags = fr.flags & (SRZ_UNIFIED | SRZ_FUSED_CLEAR | SRZ_ORDERED_RASTER);
    uint64_t nt = 0;
    for (uint32_t b = 0; b < fr.n_batches; ++b) {
      const srz_batch &sb = fr.batches[b];
      if (copy_tris && sb.n_tris && !sb.tris) return bad("batch with null triangle pointer");
      if (sb.shader < SRZ_SHADER_NORMAL || sb.shader > SRZ_SHADER_BUMP) return bad("unknown shader type");
      fs->h_batches.push_back(BatchDesc{sb.shader, sb.tex_id, (uint32_t)nt, sb.n_tris});
      nt += sb.n_tris;
    }
    if (tri_off + nt > 0xfffffff0ull || nt >= 0x7fffffffull) return bad("too many triangles");
    d.n_tris = (uint32_t)nt;
    d.n_local_bands = fs->n_local_bands;
    d.chunk_off = (uint32_t)(tri_off / 64u) + (uint32_t)f; // (every frame's chunk words start on a word of their own)
    d.group_off = (uint32_t)group_off;
    group_off += (nt + GROUP_TRIS - 1u) / GROUP_TRIS;
    fs->h_frames.push_back(d);
    fs->max_tris = std::max(fs->max_tris, d.n_tris);
    tri_off += nt, light_off += fr.n_lights, batch_off += fr.n_batches;
  }
  fs->total_tris = tri_off, fs->total_lights = light_off, fs->total_groups = group_off;
  fs->no_packed = ctx->env_no_packed;
  fs->approx_shade = ctx->opt_approx_shade;
  fs->pool_sized = ctx->opt_pool_lazy; // (SRZ_OPT_POOL_LAZY: no first-render sizing — the pool only follows the previous renders' demand)

  // stage host copies (pinned not needed: one-time upload)
  fs->tris_aos = tris_aos && copy_tris;
  std::vector<srz_tri> h_tris(copy_tris ? (size_t)tri_off : 0);
  std::vector<float> h_pos(copy_tris && !fs->tris_aos ? (size_t)tri_off * TRI_POS_F : 0); // the dense copy of the positions
  std::vector<uint16_t> h_tb((size_t)tri_off);
  std::vector<srz_light> h_lights((size_t)light_off);
  for (int f = 0; f < n_frames; ++f) {
    const srz_frame &fr = frames[f];
    const FrameDesc &d = fs->h_frames[f];
    size_t o = d.tri_off;
    for (uint32_t b = 0; b < fr.n_batches; ++b) {
      const srz_batch &sb = fr.batches[b];
      if (copy_tris && sb.n_tris) {
        std::memcpy(&h_tris[o], sb.tris, sizeof(srz_tri) * sb.n_tris);
        if (!h_pos.empty())
          for (size_t t = 0; t < sb.n_tris; ++t) std::memcpy(&h_pos[(o + t) * TRI_POS_F], &sb.tris[t].pos[0][0], sizeof(float) * TRI_POS_F);
      }
      std::fill(h_tb.begin() + o, h_tb.begin() + o + sb.n_tris, (uint16_t)b);
      o += sb.n_tris;
    }
    if (fr.n_lights) std::memcpy(&h_lights[d.light_off], fr.lights, sizeof(srz_light) * fr.n_lights);
  }
  classify_frames(fs, h_lights.data());
  auto dev_alloc = [&](void **p, size_t bytes) { return hipMalloc(p, std::max<size_t>(bytes, 256)); };
  hipError_t e = hipSuccess;
#define FS_TRY(expr)                                                                                                   \
  if (e == hipSuccess) e = (expr)
  FS_TRY(dev_alloc((void **)&fs->d_frames, sizeof(FrameDesc) * n_frames));
  FS_TRY(dev_alloc((void **)&fs->d_tris, sizeof(srz_tri) * tri_off));
  if (!fs->tris_aos) FS_TRY(dev_alloc((void **)&fs->d_tri_pos, sizeof(float) * TRI_POS_F * tri_off + 16)); // (+16: slack behind the last triangle's 9 floats, which are read as three 12-byte pieces)
  FS_TRY(dev_alloc((void **)&fs->d_bbox, sizeof(BBox) * tri_off));
  FS_TRY(dev_alloc((void **)&fs->d_chunk_rows, sizeof(uint32_t) * (tri_off / 64 + (size_t)n_frames + 1)));
  FS_TRY(dev_alloc((void **)&fs->d_band_desc, sizeof(uint32_t) * group_off * fs->n_local_bands));
  FS_TRY(dev_alloc((void **)&fs->d_band_ent, sizeof(uint2) * group_off * ENT_PER_GROUP));
  FS_TRY(dev_alloc((void **)&fs->d_tri_batch, sizeof(uint16_t) * tri_off));
  FS_TRY(dev_alloc((void **)&fs->d_batches, sizeof(BatchDesc) * fs->h_batches.size()));
  FS_TRY(dev_alloc((void **)&fs->d_lights, sizeof(srz_light) * light_off));
  fs->max_tiles = (uint32_t)n_frames * fs->n_local_bands * fs->tiles_x;
  {  // record pool: one sub-pool per ~128 binning workgroups (their bump allocators are single addresses); first guess
     // 4 records per triangle — the renders' own demand corrects it (render_impl)
    const uint64_t n_wgs = (uint64_t)((n_frames + 7) / 8 * 8) * fs->n_local_bands;
    uint32_t n_sub = 1;
    while (n_sub < 64u && (uint64_t)n_sub * 256u <= n_wgs) n_sub *= 2u;
    fs->pool_n_sub = n_sub;
    const uint64_t cap = std::min<uint64_t>((4ull * tri_off + 4096u) / n_sub + 64u, 0xfffffff0ull / n_sub);
    fs->pool_sub_cap = (uint32_t)cap;
    FS_TRY(dev_alloc((void **)&fs->d_pool, sizeof(uint32_t) * cap * n_sub));
    FS_TRY(dev_alloc((void **)&fs->d_pool_heads, sizeof(uint32_t) * CNT_STRIDE * 64)); // (one cache line per allocator)
    // what a render asked of each sub-pool comes back through pinned, device-mapped host memory: small jobs store it from
    // k_raster's first workgroup (no copy, no event, no query on the launch path), batches copy it on the clear's side stream
    FS_TRY(hipHostMalloc((void **)&fs->h_pool_heads, sizeof(uint32_t) * CNT_STRIDE * 64 * srz_frameset::DEMAND_PARTS, hipHostMallocMapped | hipHostMallocCoherent));
    if (e == hipSuccess) std::memset(fs->h_pool_heads, 0, sizeof(uint32_t) * CNT_STRIDE * 64 * srz_frameset::DEMAND_PARTS);
    // the measurement of the side clear's grid (srz_frameset::ClearTune): device-side state, and the word its decision arrives in
    FS_TRY(dev_alloc((void **)&fs->clear_tune.d_ctl, sizeof(ClearCtl)));
    if (e == hipSuccess) FS_TRY(hipMemset(fs->clear_tune.d_ctl, 0, sizeof(ClearCtl)));
    FS_TRY(hipHostMalloc((void **)&fs->clear_tune.h_wgs, 64, hipHostMallocMapped | hipHostMallocCoherent));
    if (e == hipSuccess) *fs->clear_tune.h_wgs = 0u;
    FS_TRY(dev_alloc((void **)&fs->d_tile_info, sizeof(uint2) * fs->max_tiles));
    FS_TRY(dev_alloc((void **)&fs->d_slow_list, sizeof(uint32_t) * fs->max_tiles));
    FS_TRY(dev_alloc((void **)&fs->d_slow_count, 2 * sizeof(uint32_t)));
    FS_TRY(dev_alloc((void **)&fs->d_redo_list, sizeof(uint4) * fs->max_tiles));
  }
  FS_TRY(dev_alloc((void **)&fs->d_vis, sizeof(uint32_t) * (size_t)fs->max_tiles * ((size_t)TILE * TILE)));
  FS_TRY(ensure_worklists(fs)); // (8 lists per build kind the classified frames need: not all 104)
  FS_TRY(dev_alloc((void **)&fs->d_work_count, sizeof(uint32_t) * CNT_STRIDE * N_WORK_LISTS));
  FS_TRY(dev_alloc((void **)&fs->d_sdesc, sizeof(ShadeDescG) * fs->h_batches.size()));
  FS_TRY(hipMemcpy(fs->d_frames, fs->h_frames.data(), sizeof(FrameDesc) * n_frames, hipMemcpyHostToDevice));
  if (tri_off) {
    if (copy_tris) FS_TRY(hipMemcpy(fs->d_tris, h_tris.data(), sizeof(srz_tri) * tri_off, hipMemcpyHostToDevice));
    if (!h_pos.empty()) FS_TRY(hipMemcpy(fs->d_tri_pos, h_pos.data(), sizeof(float) * h_pos.size(), hipMemcpyHostToDevice));
    FS_TRY(hipMemcpy(fs->d_tri_batch, h_tb.data(), sizeof(uint16_t) * tri_off, hipMemcpyHostToDevice));
  }
  if (!fs->h_batches.empty())
    FS_TRY(hipMemcpy(fs->d_batches, fs->h_batches.data(), sizeof(BatchDesc) * fs->h_batches.size(), hipMemcpyHostToDevice));
  if (light_off) FS_TRY(hipMemcpy(fs->d_lights, h_lights.data(), sizeof(srz_light) * light_off, hipMemcpyHostToDevice));
#undef FS_TRY
  if (e != hipSuccess) {
    free_frameset_buffers(fs);
    delete fs;
    return fail(ctx, e == hipErrorOutOfMemory ? SRZ_E_NOMEM : SRZ_E_NODEVICE,
                std::string("srz_frameset_create: ") + hipGetErrorString(e));
  }
  *out = fs;
  return SRZ_OK;
}

// A set made through the public entry points sizes its tile-list pool NOW (creation is synchronous anyway: it uploads), by a binning
// pass of its own, so that every srz_frameset_render — the first included — is asynchronous on its stream.  (The ctx's internal
// one-frame sets of srz_draw / srz_draw_scene are rendered at once: their first render does the sizing.)
static int size_pool_at_create(srz_ctx *ctx, srz_frameset **out) {
  srz_frameset *fs = *out;
  if (fs->pool_sized) return SRZ_OK; // SRZ_OPT_POOL_LAZY
  int rc = render_impl(ctx, fs, nullptr, 0, ctx->stream, false, false, /*size_only=*/true);
  if (rc == SRZ_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, SRZ_E_NODEVICE, "srz_frameset_create: the binning pass failed");
  if (rc != SRZ_OK) {
    srz_frameset_destroy(ctx, fs);
    *out = nullptr;
  }
  return rc;
}

int srz_frameset_create(srz_ctx *ctx, const srz_frame *frames, int n_frames, srz_frameset **out) {
  int rc = build_frameset(ctx, frames, n_frames, out, true);
  return rc == SRZ_OK ? size_pool_at_create(ctx, out) : rc;
}

int srz_mesh_upload(srz_ctx *ctx, int mesh_id, const srz_vertex *verts, uint32_t n_verts, const uint32_t *faces, uint32_t n_faces) {
  if (!ctx) return SRZ_E_INVALID;
  if (mesh_id < 0 || mesh_id >= MAX_MESH || !verts || !faces || n_verts == 0) return fail(ctx, SRZ_E_INVALID, "srz_mesh_upload: bad arguments");
  for (uint32_t i = 0; i < 3u * n_faces; ++i)
    if (faces[i] >= n_verts) return fail(ctx, SRZ_E_INVALID, "srz_mesh_upload: face index out of range");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  srz_ctx::MeshSlot m;
  HIP_TRY(ctx, hipMalloc(&m.d_verts, sizeof(srz_vertex) * n_verts));
  hipError_t e = hipMalloc(&m.d_faces, sizeof(uint32_t) * 3 * (n_faces ? n_faces : 1));
  if (e == hipSuccess) e = hipMemcpy(m.d_verts, verts, sizeof(srz_vertex) * n_verts, hipMemcpyHostToDevice);
  if (e == hipSuccess && n_faces) e = hipMemcpy(m.d_faces, faces, sizeof(uint32_t) * 3 * n_faces, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(m.d_verts), (void)hipFree(m.d_faces);
    return fail(ctx, SRZ_E_NOMEM, std::string("srz_mesh_upload: ") + hipGetErrorString(e));
  }
  m.n_verts = n_verts, m.n_faces = n_faces;
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(ctx->mesh[mesh_id].d_verts), (void)hipFree(ctx->mesh[mesh_id].d_faces);
  ctx->mesh[mesh_id] = m;
  return SRZ_OK;
}

static int sceneset_create_impl(srz_ctx *ctx, const srz_scene_frame *frames, int n_frames, srz_frameset **out, bool size_pool);
int srz_sceneset_create(srz_ctx *ctx, const srz_scene_frame *frames, int n_frames, srz_frameset **out) {
  return sceneset_create_impl(ctx, frames, n_frames, out, /*size_pool=*/true);
}
// (size_pool = false: srz_draw_scene's internal one-frame set — rendered at once, its first render sizes the pool: no second binning
// pass, no extra synchronisation per signature change)
static int sceneset_create_impl(srz_ctx *ctx, const srz_scene_frame *frames, int n_frames, srz_frameset **out, bool size_pool) {
  if (!ctx) return SRZ_E_INVALID;
  if (!out) return fail(ctx, SRZ_E_INVALID, "srz_sceneset_create: out is NULL");
  *out = nullptr;
  if (!frames || n_frames <= 0) return fail(ctx, SRZ_E_INVALID, "srz_sceneset_create: no frames");
  // describe every frame as an srz_frame whose batches carry sizes only; the triangles are produced by k_vertex
  std::vector<srz_frame> fr((size_t)n_frames);
  std::vector<std::vector<srz_batch>> batches((size_t)n_frames);
  for (int f = 0; f < n_frames; ++f) {
    const srz_scene_frame &sf = frames[f];
    if (sf.n_draws && !sf.draws) return fail(ctx, SRZ_E_INVALID, "srz_sceneset_create: null draws");
    for (uint32_t d = 0; d < sf.n_draws; ++d) {
      const srz_mesh_draw &dr = sf.draws[d];
      if (dr.mesh_id < 0 || dr.mesh_id >= MAX_MESH || !ctx->mesh[dr.mesh_id].d_verts)
        return fail(ctx, SRZ_E_INVALID, "srz_sceneset_create: draw names a mesh slot that was never uploaded");
      srz_batch b{};
      b.shader = dr.shader, b.tex_id = dr.tex_id, b.n_tris = ctx->mesh[dr.mesh_id].n_faces, b.tris = nullptr;
      batches[f].push_back(b);
    }
    srz_frame &o = fr[f];
    o = srz_frame{};
    o.width = sf.width, o.height = sf.height;
    std::memcpy(o.eye, sf.eye, sizeof o.eye), std::memcpy(o.ka, sf.ka, sizeof o.ka), std::memcpy(o.ks, sf.ks, sizeof o.ks);
    o.p = sf.p, o.kh = sf.kh, o.kn = sf.kn;
    o.n_lights = sf.n_lights, o.lights = sf.lights;
    o.n_batches = sf.n_draws, o.batches = batches[f].data();
    o.flags = sf.flags;
  }
  srz_frameset *fs = nullptr;
  int rc = build_frameset(ctx, fr.data(), n_frames, &fs, false);
  if (rc) return rc;
  std::vector<DrawDesc> h;
  for (int f = 0; f < n_frames; ++f) {
    uint32_t first = fs->h_frames[f].tri_off;
    for (uint32_t d = 0; d < frames[f].n_draws; ++d) {
      const srz_mesh_draw &dr = frames[f].draws[d];
      const srz_ctx::MeshSlot &m = ctx->mesh[dr.mesh_id];
      DrawDesc dd{};
      dd.verts = m.d_verts, dd.faces = m.d_faces, dd.n_faces = m.n_faces, dd.tri_off = first, dd.frame = (uint32_t)f;
      dd.zscale = frames[f].zscale, dd.zoffset = frames[f].zoffset;
      std::memcpy(dd.ndc_mvp, dr.ndc_mvp, sizeof dd.ndc_mvp), std::memcpy(dd.normal_m, dr.normal_m, sizeof dd.normal_m);
      h.push_back(dd);
      fs->h_draw_mesh.push_back(dr.mesh_id);
      first += m.n_faces;
      fs->max_faces = std::max(fs->max_faces, m.n_faces);
    }
  }
  fs->n_draws = (uint32_t)h.size();
  fs->h_draws = h;
  // one block for what srz_sceneset_update rewrites: [FrameDesc x n | lights | DrawDesc x draws], 16-byte aligned parts
  auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  fs->dyn_lights_off = up16(sizeof(FrameDesc) * (size_t)n_frames);
  fs->dyn_draws_off = up16(fs->dyn_lights_off + sizeof(srz_light) * (size_t)fs->total_lights);
  fs->dyn_bytes = up16(fs->dyn_draws_off + sizeof(DrawDesc) * std::max<size_t>(h.size(), 1));
  uint8_t *blk = nullptr;
  hipError_t e = hipMalloc(&blk, fs->dyn_bytes);
  if (e == hipSuccess) e = hipMemcpy(blk, fs->d_frames, sizeof(FrameDesc) * (size_t)n_frames, hipMemcpyDeviceToDevice);
  if (e == hipSuccess && fs->total_lights)
    e = hipMemcpy(blk + fs->dyn_lights_off, fs->d_lights, sizeof(srz_light) * (size_t)fs->total_lights, hipMemcpyDeviceToDevice);
  if (e == hipSuccess && !h.empty()) e = hipMemcpy(blk + fs->dyn_draws_off, h.data(), sizeof(DrawDesc) * h.size(), hipMemcpyHostToDevice);
  for (int i = 0; e == hipSuccess && i < srz_frameset::STAGE_RING; ++i) {
    e = hipHostMalloc((void **)&fs->h_stage[i], fs->dyn_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&fs->stage_ev[i], hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    (void)hipFree(blk);
    srz_frameset_destroy(ctx, fs);
    return fail(ctx, SRZ_E_NOMEM, std::string("srz_sceneset_create: ") + hipGetErrorString(e));
  }
  (void)hipFree(fs->d_frames);
  (void)hipFree(fs->d_lights);
  fs->d_dyn = blk;
  fs->d_frames = reinterpret_cast<FrameDesc *>(blk);
  fs->d_lights = reinterpret_cast<srz_light *>(blk + fs->dyn_lights_off);
  fs->d_draws = reinterpret_cast<DrawDesc *>(blk + fs->dyn_draws_off);
  *out = fs;
  return size_pool ? size_pool_at_create(ctx, out) : SRZ_OK; // (with the matrices of creation: a later srz_sceneset_update is followed by the lazy growth)
}

int srz_sceneset_update(srz_ctx *ctx, srz_frameset *fs, const srz_scene_frame *frames, int n_frames) {
  if (!ctx) return SRZ_E_INVALID;
  if (!fs || !frames || !fs->d_draws) return fail(ctx, SRZ_E_INVALID, "srz_sceneset_update: not a sceneset");
  if (n_frames != fs->n_frames) return fail(ctx, SRZ_E_INVALID, "srz_sceneset_update: frame count changed");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::vector<srz_light> h_lights((size_t)fs->total_lights);
  size_t di = 0, bi = 0;
  bool batches_changed = false;
  for (int f = 0; f < n_frames; ++f) {
    const srz_scene_frame &sf = frames[f];
    FrameDesc &d = fs->h_frames[f];
    if (sf.width != fs->width || sf.height != fs->height || sf.n_lights != d.n_lights || sf.n_draws != d.n_batches ||
        (sf.n_lights && !sf.lights) || (sf.n_draws && !sf.draws))
      return fail(ctx, SRZ_E_INVALID, "srz_sceneset_update: structure changed");
    std::memcpy(d.eye, sf.eye, sizeof d.eye), std::memcpy(d.ka, sf.ka, sizeof d.ka), std::memcpy(d.ks, sf.ks, sizeof d.ks);
    d.p = sf.p, d.kh = sf.kh, d.kn = sf.kn, d.flags = sf.flags & (SRZ_UNIFIED | SRZ_FUSED_CLEAR | SRZ_ORDERED_RASTER);
    if (sf.n_lights) std::memcpy(&h_lights[d.light_off], sf.lights, sizeof(srz_light) * sf.n_lights);
    for (uint32_t k = 0; k < sf.n_draws; ++k, ++di, ++bi) {
      const srz_mesh_draw &dr = sf.draws[k];
      if (dr.mesh_id != fs->h_draw_mesh[di] || dr.mesh_id < 0 || dr.mesh_id >= MAX_MESH ||
          ctx->mesh[dr.mesh_id].n_faces != fs->h_draws[di].n_faces || ctx->mesh[dr.mesh_id].d_verts != fs->h_draws[di].verts)
        return fail(ctx, SRZ_E_INVALID, "srz_sceneset_update: mesh binding changed");
      if (dr.shader < SRZ_SHADER_NORMAL || dr.shader > SRZ_SHADER_BUMP) return fail(ctx, SRZ_E_INVALID, "srz_sceneset_update: unknown shader type");
      DrawDesc &dd = fs->h_draws[di];
      dd.zscale = sf.zscale, dd.zoffset = sf.zoffset;
      std::memcpy(dd.ndc_mvp, dr.ndc_mvp, sizeof dd.ndc_mvp), std::memcpy(dd.normal_m, dr.normal_m, sizeof dd.normal_m);
      BatchDesc &b = fs->h_batches[bi];
      if (b.shader != dr.shader || b.tex_id != dr.tex_id) b.shader = dr.shader, b.tex_id = dr.tex_id, batches_changed = true;
    }
  }
  classify_frames(fs, h_lights.data());
  // (the host copies of the descriptors are re-classified by now: if the lists for a new build kind cannot be had, the set stays
  // refused by render_impl until an update succeeds — its old lists are intact, but its frames no longer match them)
  fs->update_failed = ensure_worklists(fs) != hipSuccess;
  if (fs->update_failed) return fail(ctx, SRZ_E_NOMEM, "srz_sceneset_update: hipMalloc of the work lists failed");
  // one asynchronous copy on the context's stream: ordered after every render already submitted there (which may still
  // be reading the descriptors) and before the next one.  Renders submitted on OTHER streams are the caller's to order.
  const unsigned slot = fs->stage_next++ % srz_frameset::STAGE_RING;
  if (fs->stage_busy[slot]) HIP_TRY(ctx, hipEventSynchronize(fs->stage_ev[slot])); // its copy of 4 updates ago
  uint8_t *st = fs->h_stage[slot];
  std::memcpy(st, fs->h_frames.data(), sizeof(FrameDesc) * (size_t)n_frames);
  if (!h_lights.empty()) std::memcpy(st + fs->dyn_lights_off, h_lights.data(), sizeof(srz_light) * h_lights.size());
  if (!fs->h_draws.empty()) std::memcpy(st + fs->dyn_draws_off, fs->h_draws.data(), sizeof(DrawDesc) * fs->h_draws.size());
  HIP_TRY(ctx, hipMemcpyAsync(fs->d_dyn, st, fs->dyn_bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(fs->stage_ev[slot], ctx->stream));
  fs->stage_busy[slot] = true;
  if (batches_changed) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // (rare: a draw switched shader or texture)
    HIP_TRY(ctx, hipMemcpy(fs->d_batches, fs->h_batches.data(), sizeof(BatchDesc) * fs->h_batches.size(), hipMemcpyHostToDevice));
    fs->sdesc_version = 0; // re-resolve batch → shader / texture at the next render
  }
  fs->have_stats = false;
  return SRZ_OK;
}

int srz_target_create(srz_ctx *ctx, int width, int height, srz_target **out) {
  if (!ctx) return SRZ_E_INVALID;
  if (!out || width <= 0 || height <= 0 || width > 32767 || height > 32767) return fail(ctx, SRZ_E_INVALID, "srz_target_create: bad size");
  *out = nullptr;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  srz_target *t = new (std::nothrow) srz_target();
  if (!t) return fail(ctx, SRZ_E_NOMEM, "srz_target_create: out of host memory");
  t->width = width, t->height = height;
  const size_t plane = (size_t)width * height;
  hipError_t e = hipMalloc(&t->d_planes, plane * 16);
  if (e == hipSuccess) e = hipMalloc(&t->d_bgr8, plane * 3 + 16);
  if (e != hipSuccess) {
    (void)hipFree(t->d_planes), (void)hipFree(t->d_bgr8);
    delete t;
    return fail(ctx, SRZ_E_NOMEM, "srz_target_create: hipMalloc failed");
  }
  *out = t;
  return srz_target_clear(ctx, t, 1, 1);
}

void srz_target_destroy(srz_ctx *ctx, srz_target *t) {
  if (!t) return;
  if (ctx) (void)hipSetDevice(ctx->device), (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(t->d_planes), (void)hipFree(t->d_bgr8);
  delete t;
}

static int target_materialize_clear(srz_ctx *ctx, srz_target *t) {
  if (!t->pending_clear) return SRZ_OK;
  const size_t plane = (size_t)t->width * t->height;
  HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)t->d_planes, 0x7f800000, plane, ctx->stream)); // +inf
  HIP_TRY(ctx, hipMemsetAsync(t->d_planes + plane, 0, plane * 12, ctx->stream));
  t->pending_clear = false;
  return SRZ_OK;
}

int srz_target_clear(srz_ctx *ctx, srz_target *t, int color, int depth) {
  if (!ctx || !t) return SRZ_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t plane = (size_t)t->width * t->height;
  if (color && depth) {
    t->pending_clear = true; // free if a draw follows; materialised by the next read / partial clear otherwise
    return SRZ_OK;
  }
  if (!color && !depth) return SRZ_OK;
  int rc = target_materialize_clear(ctx, t);
  if (rc) return rc;
  if (depth) HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)t->d_planes, 0x7f800000, plane, ctx->stream));
  if (color) HIP_TRY(ctx, hipMemsetAsync(t->d_planes + plane, 0, plane * 12, ctx->stream));
  return SRZ_OK;
}

int srz_target_draw(srz_ctx *ctx, srz_target *t, int primitive, srz_frameset *fs, srz_stats *stats) {
  if (!ctx) return SRZ_E_INVALID;
  if (primitive != SRZ_PRIMITIVE_LINES && primitive != SRZ_PRIMITIVE_TRIANGLES)
    return fail(ctx, SRZ_E_PRIMITIVE, "Primitive Type is not supported!");
  if (!t || !fs) return fail(ctx, SRZ_E_INVALID, "srz_target_draw: null argument");
  if (fs->n_frames != 1 || fs->width != t->width || fs->height != t->height || fs->shard_world != 1)
    return fail(ctx, SRZ_E_INVALID, "srz_target_draw: needs an unsharded 1-frame set of the target's size");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint32_t flags = t->pending_clear ? SRZ_FUSED_CLEAR : 0u;
  int rc = SRZ_OK;
  if (stats) rc = stats_pass(ctx, fs, t->pending_clear ? nullptr : t->d_planes, flags, ctx->stream, stats);
  if (rc == SRZ_OK) rc = render_impl(ctx, fs, t->d_planes, flags, ctx->stream, false);
  if (rc == SRZ_OK) t->pending_clear = false; // (a failed draw leaves the pending clear(Color|Depth) in place)
  return rc;
}

int srz_target_read(srz_ctx *ctx, srz_target *t, float *z, float *c0, float *c1, float *c2) {
  if (!ctx || !t) return SRZ_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = target_materialize_clear(ctx, t);
  if (rc) return rc;
  const size_t plane = (size_t)t->width * t->height;
  float *host[4] = {z, c0, c1, c2};
  for (int p = 0; p < 4; ++p)
    if (host[p]) HIP_TRY(ctx, hipMemcpyAsync(host[p], t->d_planes + p * plane, plane * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return SRZ_OK;
}

int srz_target_read_bgr8(srz_ctx *ctx, srz_target *t, uint8_t *bgr8) {
  if (!ctx || !t || !bgr8) return SRZ_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = target_materialize_clear(ctx, t);
  if (rc) return rc;
  const size_t plane = (size_t)t->width * t->height;
  launch_resolve8(t->d_planes, t->d_bgr8, 1, (uint32_t)t->height, (uint32_t)t->width, 4ull * plane, ctx->stream);
  HIP_TRY(ctx, hipMemcpyAsync(bgr8, t->d_bgr8, plane * 3, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return SRZ_OK;
}

void srz_frameset_destroy(srz_ctx *ctx, srz_frameset *fs) {
  if (!fs) return;
  if (ctx) {
    // renders of this set may still be running on the ctx's stream, on the clear's side stream or on a stream the caller
    // passed to srz_frameset_render: wait for the whole device before its buffers (and the pinned demand words) go
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
  }
  free_frameset_buffers(fs);
  delete fs;
}

int srz_frameset_local_rows(const srz_ctx *ctx, const srz_frameset *fs) { return fs ? (int)fs->local_rows : SRZ_E_INVALID; }

size_t srz_frameset_out_bytes(const srz_ctx *ctx, const srz_frameset *fs) {
  return fs ? (size_t)fs->n_frames * 4u * fs->local_rows * (size_t)fs->width * sizeof(float) : 0;
}

int srz_frameset_render(srz_ctx *ctx, srz_frameset *fs, void *d_out, size_t out_bytes, uint32_t flags, void *stream) {
  if (!ctx) return SRZ_E_INVALID;
  if (!fs || !d_out) return fail(ctx, SRZ_E_INVALID, "srz_frameset_render: null frameset / output");
  if (out_bytes < srz_frameset_out_bytes(ctx, fs)) return fail(ctx, SRZ_E_INVALID, "srz_frameset_render: output buffer too small");
  if (((uintptr_t)d_out & 15u) != 0) return fail(ctx, SRZ_E_INVALID, "srz_frameset_render: output must be 16-byte aligned");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = pick_stream(ctx, stream);
  flags &= SRZ_UNIFIED | SRZ_FUSED_CLEAR | SRZ_ORDERED_RASTER;
  return render_impl(ctx, fs, (float *)d_out, flags, s, false);
}

int srz_frameset_resolve8(srz_ctx *ctx, const srz_frameset *fs, const void *d_planes, void *d_bgr8, size_t bgr8_bytes, void *stream) {
  if (!ctx) return SRZ_E_INVALID;
  if (!fs || !d_planes || !d_bgr8) return fail(ctx, SRZ_E_INVALID, "srz_frameset_resolve8: null argument");
  if (bgr8_bytes < (size_t)fs->n_frames * fs->local_rows * (size_t)fs->width * 3) return fail(ctx, SRZ_E_INVALID, "srz_frameset_resolve8: output too small");
  if ((uintptr_t)d_planes & 3u) return fail(ctx, SRZ_E_INVALID, "srz_frameset_resolve8: misaligned planes");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = pick_stream(ctx, stream);
  launch_resolve8((const float *)d_planes, (uint8_t *)d_bgr8, (uint32_t)fs->n_frames, fs->local_rows, (uint32_t)fs->width,
                  4ull * fs->local_rows * (uint64_t)fs->width, s);
  HIP_TRY(ctx, hipGetLastError());
  return SRZ_OK;
}

int srz_frameset_stats(srz_ctx *ctx, srz_frameset *fs, srz_stats *stats) {
  if (!ctx) return SRZ_E_INVALID;
  if (!fs || !stats) return fail(ctx, SRZ_E_INVALID, "srz_frameset_stats: null argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // the counters come from the kernels' own state, and with the fused clear nothing reads the incoming framebuffer: the
  // counting run's pixels go to a scratch buffer of ONE frame that every frame overwrites (16.8 MB instead of 4.3 GB at 256
  // frames of 1024^2)
  float *d_out = nullptr;
  HIP_TRY(ctx, hipMalloc(&d_out, srz_frameset_out_bytes(ctx, fs) / (size_t)std::max(fs->n_frames, 1)));
  int rc = render_impl(ctx, fs, d_out, SRZ_FUSED_CLEAR, ctx->stream, true, true);
  if (rc == SRZ_OK) rc = read_stats(ctx, ctx->stream, stats);
  (void)hipFree(d_out);
  if (rc == SRZ_OK) fs->stats = *stats, fs->have_stats = true;
  return rc;
}

uint64_t srz_frameset_algorithmic_bytes(const srz_ctx *ctx, const srz_frameset *fs) {
  if (!ctx || !fs) return 0;
  // rows actually owned (not the all-gather padding)
  uint64_t rows = 0;
  for (uint32_t lb = 0; lb < fs->n_local_bands; ++lb) {
    int band = band_of((int)lb, fs->shard_rank, fs->shard_world);
    rows += (uint64_t)std::min(BAND, fs->height - band * BAND);
  }
  uint64_t fb = 16ull * (uint64_t)fs->width * rows * (uint64_t)fs->n_frames;
  uint64_t stream = 96ull * fs->total_tris + 24ull * fs->total_lights;
  uint64_t tex = 0;
  if (fs->have_stats) {
    // B_tex = min(3*texW*texH per frame, 3 bytes per texture-shaded pixel)
    uint64_t cap = 0;
    for (const BatchDesc &b : fs->h_batches)
      if (b.tex_id >= 0 && b.tex_id < MAX_TEX && ctx->h_tex[b.tex_id].bgrx &&
          (b.shader == SRZ_SHADER_TEXTURE || b.shader == SRZ_SHADER_DISPLACEMENT || b.shader == SRZ_SHADER_BUMP))
        cap = std::max<uint64_t>(cap, 3ull * ctx->h_tex[b.tex_id].w * ctx->h_tex[b.tex_id].h);
    tex = std::min<uint64_t>(cap * (uint64_t)fs->n_frames, 3ull * fs->stats.visible_textured);
  }
  return fb + stream + tex;
}

int srz_set_kernel_timing(srz_ctx *ctx, int enabled) {
  if (!ctx) return SRZ_E_INVALID;
  ctx->timing = enabled < 0 ? 0 : (enabled > 2 ? 2 : enabled);
  return SRZ_OK;
}

int srz_kernel_time_ms(srz_ctx *ctx, int reset, double *ms4, int *launches) {
  if (!ctx) return SRZ_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = collect_events(ctx);
  if (rc) return rc;
  for (int i = 0; i < 4; ++i)
    if (ms4) ms4[i] = ctx->acc_launches ? ctx->acc_ms[i] / ctx->acc_launches : 0.0;
  if (launches) *launches = ctx->acc_launches;
  if (reset) {
    ctx->acc_ms[0] = ctx->acc_ms[1] = ctx->acc_ms[2] = ctx->acc_ms[3] = 0.0, ctx->acc_launches = 0, ctx->acc_samples.clear();
    ctx->span_open = false, ctx->span_ms = 0.0;
  }
  return SRZ_OK;
}

/* the whole-launch-set time (ms) of each timed render since the last reset, in submission order, and the span from the
 * first one's start to the latest end (renders submitted to different streams overlap: the span is what they took
 * together).  Call before the resetting srz_kernel_time_ms.  *n = values written (at most cap) */
int srz_kernel_time_samples(srz_ctx *ctx, float *out, int cap, int *n, double *span_ms) {
  if (!ctx || !n || (cap > 0 && !out)) return SRZ_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = collect_events(ctx);
  if (rc) return rc;
  int m = (int)std::min<size_t>(ctx->acc_samples.size(), (size_t)std::max(cap, 0));
  for (int i = 0; i < m; ++i) out[i] = ctx->acc_samples[i];
  *n = m;
  if (span_ms) *span_ms = ctx->span_ms;
  return SRZ_OK;
}

/* self-check: exhaustive (2^32 operands) comparison of the kernels' short exact rcp / sqrt sequences with the IEEE
 * expansions. out4 = {fast-path operands, rcp mismatches, sqrt mismatches, 1/sqrt mismatches} */
int srz_verify_fastmath(srz_ctx *ctx, uint64_t *out4) {
  if (!ctx || !out4) return SRZ_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, 4 * sizeof(unsigned long long), ctx->stream));
  launch_verify_fastmath(ctx->d_stats, ctx->stream);
  unsigned long long h[4];
  HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 4; ++i) out4[i] = h[i];
  return SRZ_OK;
}

int srz_verify_fastdiv(srz_ctx *ctx, uint64_t *out3) {
  if (!ctx || !out3) return SRZ_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, 4 * sizeof(unsigned long long), ctx->stream));
  launch_verify_fastdiv(ctx->d_stats, ctx->stream);
  unsigned long long h[3];
  HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 3; ++i) out3[i] = h[i];
  return SRZ_OK;
}

int srz_verify_fastpow(srz_ctx *ctx, float p, uint64_t *out4) {
  if (!ctx || !out4) return SRZ_E_INVALID;
  if (!(p > 0.0f && p <= 4096.0f) || p == std::trunc(p)) return fail(ctx, SRZ_E_INVALID, "srz_verify_fastpow: p must be a non-integer in (0, 4096]");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, 4 * sizeof(unsigned long long), ctx->stream));
  launch_verify_fastpow(ctx->d_stats, p, ctx->stream);
  unsigned long long h[4];
  HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 4; ++i) out4[i] = h[i];
  return SRZ_OK;
}

/* diagnostic (tests): what the LAST render of the set left in its counters — out4 = { tiles the ordered rasteriser (k_raster_slow)
 * took, tiles the FAST shading builds handed to the generic one, the tile-list pool's capacity per sub-pool, the largest demand a
 * sub-pool has reported }.  Waits for the device. */
int srz_frameset_debug_counters(srz_ctx *ctx, srz_frameset *fs, uint32_t *out6) {
  if (!ctx || !fs || !out6) return ctx ? fail(ctx, SRZ_E_INVALID, "srz_frameset_debug_counters: null argument") : SRZ_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  uint32_t h[2] = {0, 0};
  HIP_TRY(ctx, hipMemcpy(h, fs->d_slow_count, sizeof h, hipMemcpyDeviceToHost));
  uint32_t need = 0;
  for (uint32_t i = 0; i < fs->pool_n_sub * (uint32_t)srz_frameset::DEMAND_PARTS; ++i) {
    const uint32_t v = static_cast<volatile uint32_t *>(fs->h_pool_heads)[((i / fs->pool_n_sub) * 64u + i % fs->pool_n_sub) * CNT_STRIDE];
    if (v > need) need = v;
  }
  out6[0] = h[0], out6[1] = h[1], out6[2] = fs->pool_sub_cap, out6[3] = need;
  const uint32_t decided = fs->clear_tune.h_wgs ? *static_cast<volatile uint32_t *>(fs->clear_tune.h_wgs) : 0u;
  out6[4] = ctx->env_clear_wgs ? ctx->env_clear_wgs : (decided ? decided : fs->clear_tune.wgs), out6[5] = (fs->clear_tune.done || decided) ? 1u : 0u;
  return SRZ_OK;
}

int srz_verify_fastlen(srz_ctx *ctx, uint64_t *out5) {
  if (!ctx || !out5) return SRZ_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, 5 * sizeof(unsigned long long), ctx->stream));
  launch_verify_fastlen(ctx->d_stats, ctx->stream);
  unsigned long long h[5];
  HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 5; ++i) out5[i] = h[i];
  return SRZ_OK;
}

/* diagnostic: raw counters of the last STATS run (incl. per-phase cycle sums); not part of the stable ABI */
int srz_debug_counters(srz_ctx *ctx, uint64_t *out, int n) {
  if (!ctx || !out) return SRZ_E_INVALID;
#ifdef SRZ_PHASE_PROBE /* dev build: the device counters as they are now (k_shade's phase clocks), then zeroed */
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(ctx->dbg, ctx->d_stats, sizeof ctx->dbg, hipMemcpyDeviceToHost);
  (void)hipMemset(ctx->d_stats, 0, sizeof ctx->dbg);
#endif
  for (int i = 0; i < n && i < ST_COUNT; ++i) out[i] = ctx->dbg[i];
  return ST_COUNT;
}

/* ---- multi-GPU: the band exchange on RCCL ---------------------------------------------------------------------------
 * librccl is loaded on first use (a single-GPU program never needs it); if the process already holds a copy (PyTorch
 * ships its own), that one is used. */
} // extern "C"
namespace {
struct RcclApi {
  void *handle = nullptr;
  struct UniqueId { char internal[128]; };
  int (*GetUniqueId)(UniqueId *) = nullptr;
  int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string error;
};
RcclApi &rccl() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api;
  tried = true;
  const char *names[] = {"librccl.so", "librccl.so.1"};
  for (const char *n : names)
    if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); // a copy the process already loaded
  for (const char *n : names)
    if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!api.handle) api.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!api.handle) {
    api.error = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?");
    return api;
  }
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
  api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.handle, "ncclAllGather"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
  if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy) api.error = "librccl lacks the nccl* entry points";
  return api;
}
std::string rccl_err(int rc) {
  RcclApi &r = rccl();
  return r.GetErrorString ? r.GetErrorString(rc) : ("rccl error " + std::to_string(rc));
}
} // namespace
extern "C" {

struct srz_comm {
  void *comm = nullptr;
  int rank = 0, world = 1;
};

int srz_comm_unique_id(uint8_t *out128) {
  if (!out128) return fail(nullptr, SRZ_E_INVALID, "srz_comm_unique_id: out is NULL");
  RcclApi &r = rccl();
  if (!r.error.empty()) return fail(nullptr, SRZ_E_NODEVICE, "srz_comm_unique_id: " + r.error);
  RcclApi::UniqueId id;
  const int rc = r.GetUniqueId(&id);
  if (rc != 0) return fail(nullptr, SRZ_E_NODEVICE, "ncclGetUniqueId: " + rccl_err(rc));
  std::memcpy(out128, id.internal, 128);
  return SRZ_OK;
}

int srz_comm_create(srz_ctx *ctx, const uint8_t *id128, int rank, int world, srz_comm **out) {
  if (!ctx) return SRZ_E_INVALID;
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return fail(ctx, SRZ_E_INVALID, "srz_comm_create: bad arguments");
  *out = nullptr;
  RcclApi &r = rccl();
  if (!r.error.empty()) return fail(ctx, SRZ_E_NODEVICE, "srz_comm_create: " + r.error);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  srz_comm *c = new (std::nothrow) srz_comm();
  if (!c) return fail(ctx, SRZ_E_NOMEM, "srz_comm_create: out of host memory");
  RcclApi::UniqueId id;
  std::memcpy(id.internal, id128, 128);
  const int rc = r.CommInitRank(&c->comm, world, id, rank);
  if (rc != 0) {
    delete c;
    return fail(ctx, SRZ_E_NODEVICE, "ncclCommInitRank: " + rccl_err(rc));
  }
  c->rank = rank, c->world = world;
  ctx->shard_rank = rank, ctx->shard_world = world; // = srz_set_shard: framesets created from now on are this rank's bands
  *out = c;
  return SRZ_OK;
}

void srz_comm_destroy(srz_ctx *ctx, srz_comm *c) {
  if (!c) return;
  if (ctx) (void)hipSetDevice(ctx->device), (void)hipDeviceSynchronize();
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  delete c;
}

size_t srz_frameset_exchange_bytes(const srz_ctx *ctx, const srz_frameset *fs, int what) {
  if (!fs) return 0;
  const size_t row = what == SRZ_EXCHANGE_BGR8 ? (size_t)fs->width * 3u : (size_t)fs->width * sizeof(float);
  const size_t planes = what == SRZ_EXCHANGE_BGR8 ? 1u : 4u;
  return (size_t)fs->n_frames * planes * fs->local_rows * row;
}

int srz_frameset_deinterleave(srz_ctx *ctx, const srz_frameset *fs, const void *d_gathered, void *d_full, int what, void *stream) {
  if (!ctx) return SRZ_E_INVALID;
  if (!fs || !d_gathered || !d_full || (what != SRZ_EXCHANGE_PLANES && what != SRZ_EXCHANGE_BGR8))
    return fail(ctx, SRZ_E_INVALID, "srz_frameset_deinterleave: bad arguments");
  if (fs->shard_world == 1) return fail(ctx, SRZ_E_INVALID, "srz_frameset_deinterleave: the frameset is not sharded");
  const uint32_t row_bytes = what == SRZ_EXCHANGE_BGR8 ? (uint32_t)fs->width * 3u : (uint32_t)fs->width * 4u;
  if (what == SRZ_EXCHANGE_PLANES && (((uintptr_t)d_gathered | (uintptr_t)d_full) & 3u))
    return fail(ctx, SRZ_E_INVALID, "srz_frameset_deinterleave: misaligned buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = pick_stream(ctx, stream);
  launch_deinterleave(d_gathered, d_full, (uint32_t)fs->shard_world, (uint32_t)fs->n_frames * (what == SRZ_EXCHANGE_BGR8 ? 1u : 4u),
                      fs->bands_per_rank, row_bytes, s);
  HIP_TRY(ctx, hipGetLastError());
  return SRZ_OK;
}

int srz_frameset_allgather(srz_ctx *ctx, srz_comm *c, const srz_frameset *fs, const void *d_shard, void *d_gathered, void *d_full,
                           int what, void *stream) {
  if (!ctx) return SRZ_E_INVALID;
  if (!c || !fs || !d_shard || !d_gathered || !d_full) return fail(ctx, SRZ_E_INVALID, "srz_frameset_allgather: null argument");
  if (fs->shard_world != c->world || fs->shard_rank != c->rank)
    return fail(ctx, SRZ_E_INVALID, "srz_frameset_allgather: the frameset was not created under this communicator's shard");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = pick_stream(ctx, stream);
  const size_t bytes = srz_frameset_exchange_bytes(ctx, fs, what);
  if (bytes == 0) return fail(ctx, SRZ_E_INVALID, "srz_frameset_allgather: bad exchange kind");
  if (c->world == 1) { // one rank: the shard is the frame
    HIP_TRY(ctx, hipMemcpyAsync(d_full, d_shard, bytes, hipMemcpyDeviceToDevice, s));
    return SRZ_OK;
  }
  const int rc = rccl().AllGather(d_shard, d_gathered, bytes, /* ncclUint8 */ 1, c->comm, s);
  if (rc != 0) return fail(ctx, SRZ_E_NODEVICE, "ncclAllGather: " + rccl_err(rc));
  return srz_frameset_deinterleave(ctx, fs, d_gathered, d_full, what, stream);
}

int srz_frameset_allgather_inplace(srz_ctx *ctx, srz_comm *c, const srz_frameset *fs, void *d_gathered, int what, void *stream) {
  if (!ctx) return SRZ_E_INVALID;
  if (!c || !fs || !d_gathered) return fail(ctx, SRZ_E_INVALID, "srz_frameset_allgather_inplace: null argument");
  if (fs->shard_world != c->world || fs->shard_rank != c->rank)
    return fail(ctx, SRZ_E_INVALID, "srz_frameset_allgather_inplace: the frameset was not created under this communicator's shard");
  const size_t bytes = srz_frameset_exchange_bytes(ctx, fs, what);
  if (bytes == 0 || (what != SRZ_EXCHANGE_PLANES && what != SRZ_EXCHANGE_BGR8)) return fail(ctx, SRZ_E_INVALID, "srz_frameset_allgather_inplace: bad exchange kind");
  if (c->world == 1) return SRZ_OK; // one rank: its shard is the whole buffer
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = pick_stream(ctx, stream);
  // in place: sendbuff == recvbuff + rank * count (the form NCCL / RCCL document for an all-gather without a copy of the own part)
  const int rc = rccl().AllGather(static_cast<const uint8_t *>(d_gathered) + (size_t)c->rank * bytes, d_gathered, bytes, /* ncclUint8 */ 1, c->comm, s);
  if (rc != 0) return fail(ctx, SRZ_E_NODEVICE, "ncclAllGather (in place): " + rccl_err(rc));
  return SRZ_OK;
}

size_t srz_frameset_gathered_row_offset(const srz_ctx *ctx, const srz_frameset *fs, int what, int frame, int plane, int row) {
  if (!fs || frame < 0 || frame >= fs->n_frames || row < 0 || row >= fs->height) return (size_t)-1;
  if (what != SRZ_EXCHANGE_PLANES && what != SRZ_EXCHANGE_BGR8) return (size_t)-1;
  const size_t planes = what == SRZ_EXCHANGE_BGR8 ? 1u : 4u;
  if (plane < 0 || (size_t)plane >= planes) return (size_t)-1;
  const size_t row_bytes = what == SRZ_EXCHANGE_BGR8 ? (size_t)fs->width * 3u : (size_t)fs->width * sizeof(float);
  const size_t band = (size_t)row / BAND, world = (size_t)fs->shard_world;
  const size_t rank = (size_t)rank_of_band((int)band, (int)world), local_row = (band / world) * BAND + (size_t)row % BAND;
  const size_t shard_rows = fs->shard_world == 1 ? (size_t)fs->height : (size_t)fs->bands_per_rank * BAND;
  return (((rank * (size_t)fs->n_frames + (size_t)frame) * planes + (size_t)plane) * shard_rows + local_row) * row_bytes;
}

int srz_frameset_read_gathered_frame(srz_ctx *ctx, const srz_frameset *fs, const void *d_gathered, int what, int frame, void *host_out,
                                     void *stream) {
  if (!ctx) return SRZ_E_INVALID;
  if (!fs || !d_gathered || !host_out || frame < 0 || frame >= fs->n_frames || (what != SRZ_EXCHANGE_PLANES && what != SRZ_EXCHANGE_BGR8))
    return fail(ctx, SRZ_E_INVALID, "srz_frameset_read_gathered_frame: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = pick_stream(ctx, stream);
  const size_t planes = what == SRZ_EXCHANGE_BGR8 ? 1u : 4u;
  const size_t row_bytes = what == SRZ_EXCHANGE_BGR8 ? (size_t)fs->width * 3u : (size_t)fs->width * sizeof(float);
  const size_t band_bytes = row_bytes * BAND, world = (size_t)fs->shard_world;
  const size_t n_bands = ((size_t)fs->height + BAND - 1) / BAND;
  // band by band (a rank's bands are consecutive in its shard but, with the rotated band → rank map, not equidistant in the image)
  (void)band_bytes, (void)world;
  for (size_t p = 0; p < planes; ++p)
    for (size_t b = 0; b < n_bands; ++b) {
      const size_t rows = std::min<size_t>(BAND, (size_t)fs->height - b * BAND);
      const uint8_t *src = static_cast<const uint8_t *>(d_gathered) + srz_frameset_gathered_row_offset(ctx, fs, what, frame, (int)p, (int)(b * BAND));
      uint8_t *dst = static_cast<uint8_t *>(host_out) + (p * (size_t)fs->height + b * BAND) * row_bytes;
      HIP_TRY(ctx, hipMemcpyAsync(dst, src, rows * row_bytes, hipMemcpyDeviceToHost, s));
    }
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return SRZ_OK;
}

/* Page-locks `bytes` of the caller's memory at `ptr` (hipHostRegister): planes inside a registered range move between host and
 * device by DMA at the link's rate in srz_draw / srz_draw_scene / srz_draw_batch instead of through the runtime's staging copies. */
int srz_host_register(srz_ctx *ctx, void *ptr, size_t bytes) {
  if (!ctx || !ptr || !bytes) return ctx ? fail(ctx, SRZ_E_INVALID, "srz_host_register: null argument") : SRZ_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  return SRZ_OK;
}
int srz_host_unregister(srz_ctx *ctx, void *ptr) {
  if (!ctx || !ptr) return ctx ? fail(ctx, SRZ_E_INVALID, "srz_host_unregister: null argument") : SRZ_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // (nothing of ours may still be copying from / to the range)
  HIP_TRY(ctx, hipHostUnregister(ptr));
  return SRZ_OK;
}

int srz_sync(srz_ctx *ctx) {
  if (!ctx) return SRZ_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return SRZ_OK;
}

// re-upload the data of a 1-frame set made by srz_frameset_create (same structure: checked by the caller's signature)
static int refresh_plain_frame(srz_ctx *ctx, srz_frameset *fs, const srz_frame &fr, hipStream_t s) {
  FrameDesc &d = fs->h_frames[0];
  std::memcpy(d.eye, fr.eye, sizeof d.eye), std::memcpy(d.ka, fr.ka, sizeof d.ka), std::memcpy(d.ks, fr.ks, sizeof d.ks);
  d.p = fr.p, d.kh = fr.kh, d.kn = fr.kn;
  d.flags = fr.flags & (SRZ_UNIFIED | SRZ_FUSED_CLEAR | SRZ_ORDERED_RASTER);
  if (fr.n_lights != d.n_lights || (fr.n_lights && !fr.lights)) return fail(ctx, SRZ_E_INVALID, "srz_draw: light count changed");
  classify_frames(fs, fr.lights, /*one_frame=*/true);
  fs->update_failed = ensure_worklists(fs) != hipSuccess; // (see srz_sceneset_update)
  if (fs->update_failed) return fail(ctx, SRZ_E_NOMEM, "srz_draw: hipMalloc of the work lists failed");
  HIP_TRY(ctx, hipMemcpyAsync(fs->d_frames, fs->h_frames.data(), sizeof(FrameDesc), hipMemcpyHostToDevice, s));
  if (fr.n_lights) HIP_TRY(ctx, hipMemcpyAsync(fs->d_lights, fr.lights, sizeof(srz_light) * fr.n_lights, hipMemcpyHostToDevice, s));
  size_t o = 0;
  for (uint32_t b = 0; b < fr.n_batches; ++b) {
    const srz_batch &sb = fr.batches[b];
    if (sb.n_tris && !sb.tris) return fail(ctx, SRZ_E_INVALID, "srz_draw: batch with null triangle pointer");
    if (sb.n_tris) HIP_TRY(ctx, hipMemcpyAsync(fs->d_tris + o, sb.tris, sizeof(srz_tri) * sb.n_tris, hipMemcpyHostToDevice, s));
    o += sb.n_tris;
  }
  fs->have_stats = false;
  return SRZ_OK;
}

static int draw_impl(srz_ctx *ctx, int primitive, const srz_frame *frame, const srz_scene_frame *scene, float *z, float *c0,
                     float *c1, float *c2, srz_stats *stats) {
  if (!ctx) return SRZ_E_INVALID;
  if (primitive != SRZ_PRIMITIVE_LINES && primitive != SRZ_PRIMITIVE_TRIANGLES)
    return fail(ctx, SRZ_E_PRIMITIVE, "Primitive Type is not supported!");
  if ((!frame && !scene) || !z || !c0 || !c1 || !c2) return fail(ctx, SRZ_E_INVALID, "srz_draw: null argument");
  if (ctx->shard_world != 1) return fail(ctx, SRZ_E_INVALID, "srz_draw: whole-frame draw needs an unsharded ctx (srz_set_shard(ctx,0,1))");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int W = frame ? frame->width : scene->width, H = frame ? frame->height : scene->height;
  // ---- the structure of this call; equal to the previous call's → its frameset and framebuffer are reused ----------------
  std::vector<uint64_t> sig;
  if (frame) {
    if ((frame->n_lights && !frame->lights) || (frame->n_batches && !frame->batches)) return fail(ctx, SRZ_E_INVALID, "srz_draw: null lights/batches");
    sig = {1u, (uint64_t)(uint32_t)W << 32 | (uint32_t)H, frame->n_lights, frame->n_batches};
    for (uint32_t b = 0; b < frame->n_batches; ++b)
      sig.push_back((uint64_t)frame->batches[b].n_tris << 32 | (uint64_t)(uint8_t)frame->batches[b].shader << 16 | (uint16_t)frame->batches[b].tex_id);
  } else {
    sig = {2u, (uint64_t)(uint32_t)W << 32 | (uint32_t)H, scene->n_lights, scene->n_draws};
  }
  int rc = SRZ_OK;
  bool reuse = ctx->draw_fs && sig == ctx->draw_sig;
  if (reuse) {
    ctx->draw_fs->approx_shade = ctx->opt_approx_shade; // (the ctx's own one-frame set follows the option call by call)
    rc = frame ? refresh_plain_frame(ctx, ctx->draw_fs, *frame, s) : srz_sceneset_update(ctx, ctx->draw_fs, scene, 1);
    if (rc != SRZ_OK && !frame) reuse = false, rc = SRZ_OK; // (a scene whose mesh bindings changed: rebuild)
    if (rc != SRZ_OK) return rc;
  }
  if (!reuse) {
    if (ctx->draw_fs) srz_frameset_destroy(ctx, ctx->draw_fs);
    (void)hipFree(ctx->draw_out);
    ctx->draw_fs = nullptr, ctx->draw_out = nullptr, ctx->draw_sig.clear();
    rc = frame ? build_frameset(ctx, frame, 1, &ctx->draw_fs, true, /*tris_aos=*/true) : sceneset_create_impl(ctx, scene, 1, &ctx->draw_fs, false);
    if (rc) return rc;
    if (hipMalloc(&ctx->draw_out, 4 * (size_t)W * H * sizeof(float)) != hipSuccess) {
      srz_frameset_destroy(ctx, ctx->draw_fs);
      ctx->draw_fs = nullptr;
      return fail(ctx, SRZ_E_NOMEM, "srz_draw: hipMalloc failed");
    }
    ctx->draw_sig = sig;
  }
  srz_frameset *fs = ctx->draw_fs;
  float *d_out = ctx->draw_out;
  const uint32_t fflags = frame ? frame->flags : scene->flags;
  const size_t plane = (size_t)W * H, pb = plane * sizeof(float);
  hipError_t e = hipSuccess;
  const bool fused = (fflags & SRZ_FUSED_CLEAR) != 0;
  float *host[4] = {z, c0, c1, c2};
  if (!fused)
    for (int p = 0; p < 4 && e == hipSuccess; ++p) e = hipMemcpyAsync(d_out + p * plane, host[p], pb, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    if (stats) rc = stats_pass(ctx, fs, fused ? nullptr : d_out, 0, s, stats);
    if (rc == SRZ_OK) rc = render_impl(ctx, fs, d_out, 0, s, false);
  }
  // (planes the caller page-locked with srz_host_register move by DMA straight from / to his memory; pageable ones through the
  // runtime's staging buffers.  SRZ_NO_Z_READBACK: the depth plane stays on the device)
  for (int p = (fflags & SRZ_NO_Z_READBACK) ? 1 : 0; p < 4 && e == hipSuccess && rc == SRZ_OK; ++p)
    e = hipMemcpyAsync(host[p], d_out + p * plane, pb, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return fail(ctx, SRZ_E_NODEVICE, std::string("srz_draw: ") + hipGetErrorString(e));
  return rc;
}

int srz_draw(srz_ctx *ctx, int primitive, const srz_frame *frame, float *z, float *c0, float *c1, float *c2,
             srz_stats *stats) {
  if (ctx && !frame) return fail(ctx, SRZ_E_INVALID, "srz_draw: null argument");
  return draw_impl(ctx, primitive, frame, nullptr, z, c0, c1, c2, stats);
}

int srz_draw_scene(srz_ctx *ctx, int primitive, const srz_scene_frame *frame, float *z, float *c0, float *c1, float *c2,
                   srz_stats *stats) {
  if (ctx && !frame) return fail(ctx, SRZ_E_INVALID, "srz_draw_scene: null argument");
  return draw_impl(ctx, primitive, nullptr, frame, z, c0, c1, c2, stats);
}

int srz_draw_batch(srz_ctx *ctx, int primitive, const srz_frame *frames, int n_frames, float *const *planes, srz_stats *stats) {
  if (!ctx) return SRZ_E_INVALID;
  if (primitive != SRZ_PRIMITIVE_LINES && primitive != SRZ_PRIMITIVE_TRIANGLES)
    return fail(ctx, SRZ_E_PRIMITIVE, "Primitive Type is not supported!");
  if (!frames || n_frames <= 0 || !planes) return fail(ctx, SRZ_E_INVALID, "srz_draw_batch: null argument");
  for (int f = 0; f < n_frames; ++f)
    if (!planes[f]) return fail(ctx, SRZ_E_INVALID, "srz_draw_batch: null plane pointer");
  if (ctx->shard_world != 1) return fail(ctx, SRZ_E_INVALID, "srz_draw_batch: whole-frame draw needs an unsharded ctx");
  srz_frameset *fs = nullptr;
  int rc = srz_frameset_create(ctx, frames, n_frames, &fs);
  if (rc) return rc;
  const size_t fb = 4ull * (size_t)fs->width * (size_t)fs->height * sizeof(float); // one frame: z,c0,c1,c2 planes
  // (the device planes are kept between calls: allocating and freeing half a gigabyte costs as much as moving it)
  hipError_t e = hipSuccess;
  if (ctx->batch_out_bytes < fb * (size_t)n_frames) {
    (void)hipFree(ctx->batch_out);
    ctx->batch_out = nullptr, ctx->batch_out_bytes = 0;
    e = hipMalloc(&ctx->batch_out, fb * (size_t)n_frames);
    if (e != hipSuccess) {
      srz_frameset_destroy(ctx, fs);
      return fail(ctx, SRZ_E_NOMEM, "srz_draw_batch: hipMalloc failed");
    }
    ctx->batch_out_bytes = fb * (size_t)n_frames;
  }
  float *d_out = ctx->batch_out;
  hipStream_t s = ctx->stream;
  // The set is rendered in pieces of whole frames (~128 MB of planes each), all enqueued at once; a second stream brings piece k
  // back to the caller's planes while piece k + 1 renders (and, for accumulate-mode frames, while its planes go up) — with planes
  // the caller page-locked (srz_host_register) both directions are DMA at the link's rate, pageable ones go through the runtime's
  // staging copies.  A counting run (stats) renders in one piece first.
  if (!ctx->stream3) {
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking));
    for (int i = 0; i < srz_ctx::EV_RING; ++i) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_piece[i], hipEventDisableTiming));
  }
  const int per = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_frames, ((size_t)128 << 20) / fb));
  const int n_pieces = (n_frames + per - 1) / per;
  const size_t plane_b = fb / 4;
  if (stats) { // (frames with SRZ_FUSED_CLEAR ignore what the scratch copy holds)
    for (int f = 0; f < n_frames && e == hipSuccess; ++f)
      if (!(frames[f].flags & SRZ_FUSED_CLEAR))
        e = hipMemcpyAsync(reinterpret_cast<uint8_t *>(d_out) + fb * f, planes[f], fb, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) rc = stats_pass(ctx, fs, d_out, 0, s, stats);
  }
  // (piece k's read-back is issued AFTER piece k + 1's render has been enqueued: a copy to pageable memory holds the calling thread
  // until it is done, and the device must have its next piece by then)
  auto read_back = [&](int k) {
    const int f0 = k * per, n = std::min(per, n_frames - f0);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream3, ctx->ev_piece[k % srz_ctx::EV_RING], 0); // "piece k has been rendered"
    for (int f = f0; f < f0 + n && e == hipSuccess; ++f) {
      const size_t skip = (frames[f].flags & SRZ_NO_Z_READBACK) ? plane_b : 0; // (the depth plane is the first of a frame's four)
      e = hipMemcpyAsync(reinterpret_cast<uint8_t *>(planes[f]) + skip, reinterpret_cast<uint8_t *>(d_out) + fb * f + skip, fb - skip,
                         hipMemcpyDeviceToHost, ctx->stream3);
    }
  };
  for (int k = 0; k < n_pieces && e == hipSuccess && rc == SRZ_OK; ++k) {
    const int f0 = k * per, n = std::min(per, n_frames - f0);
    if (!stats) // accumulate-mode frames start from the caller's planes (a counting run has uploaded them already)
      for (int f = f0; f < f0 + n && e == hipSuccess; ++f)
        if (!(frames[f].flags & SRZ_FUSED_CLEAR))
          e = hipMemcpyAsync(reinterpret_cast<uint8_t *>(d_out) + fb * f, planes[f], fb, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) break;
    rc = render_impl(ctx, fs, d_out, 0, s, false, false, false, f0, n);
    if (rc != SRZ_OK) break;
    e = hipEventRecord(ctx->ev_piece[k % srz_ctx::EV_RING], s); // (slot k % 8 was last waited for by piece k - 8's read-back, issued long ago)
    if (k > 0) read_back(k - 1);
  }
  if (e == hipSuccess && rc == SRZ_OK) read_back(n_pieces - 1);
  { // both streams drain before the buffers go, whatever happened above
    const hipError_t e1 = hipStreamSynchronize(s), e2 = hipStreamSynchronize(ctx->stream3);
    if (e == hipSuccess) e = e1 != hipSuccess ? e1 : e2;
  }
  srz_frameset_destroy(ctx, fs);
  if (e != hipSuccess) return fail(ctx, SRZ_E_NODEVICE, std::string("srz_draw_batch: ") + hipGetErrorString(e));
  return rc;
}

} // extern "C"
