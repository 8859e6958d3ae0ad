/*
 * srz.h — C ABI of the MI355X-native triangle rasterization + fragment-shading stage.
 *
 * This is the drop-in boundary for ONE path of Liupeter01/Software-Rasterizer:
 *     virtual void RenderingPipeline::draw(Primitive) = 0        (include/base/Render.hpp:84)
 *     void TraditionalRasterizer::draw(Primitive)                (src/Rasterizer.cpp:183-240)
 * The reference has no FFI; the seam is that C++ virtual.  Everything `draw` pulls from the
 * scene (post-MVP triangle stream, lights, eye, shader type, texture, Blinn-Phong constants)
 * is passed here as plain pointers and sizes; everything it writes (z-buffer + 3 planar float
 * colour planes, include/base/Render.hpp:250-257) comes back through plain pointers.
 *
 * Conventions (all entry points):
 *   - return 0 on success, negative SRZ_E_* on error; text via srz_last_error().
 *   - no exception crosses this boundary; the C++ host shim (software-rasterizer_amd/host)
 *     rethrows as std::runtime_error for draw() to keep the reference's convention
 *     (src/Rasterizer.cpp:185-189).
 *   - the caller owns every host pointer; the library copies in/out and keeps only device state.
 *   - one ctx per host thread, one GPU per ctx, one process per GPU (multi-GPU = one ctx per rank,
 *     bands of 32 rows dealt round-robin (each round of N rotated by five ranks) with srz_set_shard, reassembled by an RCCL all-gather).
 *   - there is NO CPU fallback: every compute entry point fails with SRZ_E_NODEVICE when no
 *     gfx950 device is usable.
 */
#ifndef SRZ_H_
#define SRZ_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an entry point is added, a struct changes or an argument changes its meaning; every binding compares it
 * with srz_abi_version() when it loads the library and refuses a mismatch (srz/__init__.py: ImportError; libsrz_host.so:
 * std::runtime_error from the TraditionalRasterizer constructor).
 *   1  round 1
 *   2  srz_draw_batch, scenesets, targets
 *   3  `stream` arguments: NULL = the ctx's own stream, SRZ_STREAM_NULL = HIP's null stream (was: NULL = null stream);
 *      srz_comm_*, srz_frameset_allgather / _deinterleave / _exchange_bytes, srz_kernel_time_samples
 *   4  srz_frameset_allgather_inplace, srz_frameset_gathered_row_offset, srz_frameset_read_gathered_frame
 *   5  srz_set_option, srz_verify_fastpow; srz_frameset_gathered_row_offset returns (size_t)-1 for an unknown `what` too
 *   6  SRZ_OPT_APPROX_SHADE (the tolerance mode of the shaders); srz_frameset_resolve8 / _deinterleave / the bgr8 exchange take any width;
 *      the tile-list pool is sized by srz_frameset_create / srz_sceneset_create, the first srz_frameset_render no longer blocks
 *   7  srz_host_register / srz_host_unregister, SRZ_NO_Z_READBACK, srz_verify_fastlen, srz_frameset_debug_counters; srz_draw_batch reads one piece of the batch
 *      back while the next one renders
 */
#define SRZ_ABI_VERSION 7

/* error codes */
#define SRZ_OK 0
#define SRZ_E_INVALID (-1)  /* bad argument (null pointer, bad size, unknown shader, ...) */
#define SRZ_E_NODEVICE (-2) /* no usable HIP device / HIP runtime error */
#define SRZ_E_NOMEM (-3)    /* host or device allocation failed */
#define SRZ_E_TEXTURE (-4)  /* a TEXTURE/BUMP/DISPLACEMENT batch names a tex_id that was not uploaded */
#define SRZ_E_PRIMITIVE (-5) /* primitive type not supported (src/Rasterizer.cpp:185-189) */

/* = SoftRasterizer::SHADERS_TYPE (include/shader/Shader.hpp:32-38) */
#define SRZ_SHADER_NORMAL 0
#define SRZ_SHADER_TEXTURE 1
#define SRZ_SHADER_PHONG 2
#define SRZ_SHADER_DISPLACEMENT 3
#define SRZ_SHADER_BUMP 4

/* = SoftRasterizer::Primitive (include/base/Render.hpp:74) */
#define SRZ_PRIMITIVE_LINES 0
#define SRZ_PRIMITIVE_TRIANGLES 1

/* frame flags */
#define SRZ_EXACT_SPLIT 0u /* default: reproduce the reference's AVX-columns / scalar-tail split per (triangle,pixel) */
#define SRZ_UNIFIED 1u     /* every pixel uses the 8-wide ("AVX") semantics; NOT reference-exact, for A/B only */
#define SRZ_FUSED_CLEAR 2u /* treat z/colour as just cleared (clear(Color|Depth), src/Render.cpp:46-55): write-only framebuffer */
#define SRZ_NO_Z_READBACK 8u /* srz_draw / srz_draw_scene / srz_draw_batch only: the depth plane is not copied back to the host (the caller
                              * reads colour only: a quarter of the PCIe bytes less).  The caller's z buffer is then stale: use it
                              * with SRZ_FUSED_CLEAR frames, i.e. where the next draw does not start from it */
#define SRZ_ORDERED_RASTER 4u /* rasterise every tile with the reference's ordered triangle walk (src/Rasterizer.cpp:199-236)
                               * instead of the order-independent depth keys; same result bit for bit, slower; for A/B only */

/* Post-MVP triangle = payload of SoftRasterizer::Triangle that draw() consumes
 * (m_vertex/m_normal/m_texCoords, include/object/Triangle.hpp:88-93).  Screen-space x,y in pixels,
 * z remapped to [near,far] (src/Scene.cpp:937-947).  bbox, cull and binning are recomputed on device. */
typedef struct srz_tri {
  float pos[3][3];
  float nrm[3][3];
  float uv[3][2];
} srz_tri; /* 96 B */

/* = light_struct {position,intensity} (include/light/Light.hpp:8-45) */
typedef struct srz_light {
  float pos[3];
  float intensity[3];
} srz_light; /* 24 B */

/* One mesh's triangles with the shader bound to that mesh = one ObjTuple of
 * Scene::loadTriangleStream (include/scene/Scene.hpp:30-31). Batches are drawn in array order,
 * triangles in array order (src/Rasterizer.cpp:199-200). */
typedef struct srz_batch {
  int32_t shader;  /* SRZ_SHADER_* */
  int32_t tex_id;  /* texture slot (srz_texture_upload); ignored by NORMAL / PHONG */
  uint32_t n_tris;
  uint32_t _pad;
  const srz_tri *tris; /* host pointer, n_tris entries */
} srz_batch;

/* Everything one draw() of one scene pulls (src/Rasterizer.cpp:191-196) */
typedef struct srz_frame {
  int32_t width, height; /* RenderingPipeline::m_width/m_height */
  float eye[3];          /* Scene::loadEyeVec() (camera POSITION) */
  float ka[3], ks[3];    /* Shader::ka / ks statics (src/Shader.cpp:7-8) */
  float p;               /* Shader::p (src/Shader.cpp:10) */
  float kh, kn;          /* Shader::kh / kn (src/Shader.cpp:11-12), BUMP / DISPLACEMENT only */
  uint32_t n_lights;
  uint32_t n_batches;
  const srz_light *lights;
  const srz_batch *batches;
  uint32_t flags; /* SRZ_EXACT_SPLIT | SRZ_UNIFIED | SRZ_FUSED_CLEAR | SRZ_ORDERED_RASTER */
  uint32_t _pad;
} srz_frame;

/* Per-call counters (summed over the frames of the call). */
typedef struct srz_stats {
  uint64_t n_tris;      /* triangles submitted */
  uint64_t n_culled;    /* rejected by the backface test (src/Rasterizer.cpp:203-205) */
  uint64_t pixel_tests; /* sum of bbox areas of the surviving triangles */
  uint64_t fragments;   /* (triangle,pixel) pairs that pass the coverage test */
  uint64_t shaded;      /* of those, pairs that also pass the z-test in submission order (the reference's ordered walk;
                         * a call that asks for stats runs the ordered rasteriser once more on a scratch copy) */
  uint64_t visible;     /* pixels whose final owner is a triangle of this call */
  uint64_t visible_textured; /* of those, pixels whose owner's shader fetches the texture (B_tex of the roofline) */
} srz_stats;

/* ---- device vertex stage (= Scene::loadTriangleStream, src/Scene.cpp:903-964, run on the GPU) ------------------
 * Instead of post-MVP triangles the caller hands over meshes (uploaded once) and, per frame, one srz_mesh_draw per mesh:
 * the two matrices loadTriangleStream builds (:922-923) and the depth remap (:279-280).  Matrices are column-major
 * (glm layout, m[col*4+row]). */
typedef struct srz_vertex { /* = SoftRasterizer::Vertex {position, normal, texCoord} (include/object/Object.hpp:17-31) */
  float pos[3];
  float nrm[3];
  float uv[2];
} srz_vertex; /* 32 B */

typedef struct srz_mesh_draw {
  int32_t mesh_id;     /* srz_mesh_upload slot */
  int32_t shader;      /* SRZ_SHADER_* of the Shader bound to the mesh */
  int32_t tex_id;      /* texture slot, ignored by NORMAL / PHONG */
  int32_t _pad;
  float ndc_mvp[16];   /* m_ndcToScreenMatrix * m_projection * m_view * modelMatrix */
  float normal_m[16];  /* transpose(inverse(modelMatrix)) */
} srz_mesh_draw;

typedef struct srz_scene_frame {
  int32_t width, height;
  float eye[3], ka[3], ks[3];
  float p, kh, kn;
  float zscale, zoffset; /* (far-near)/2, (far+near)/2 */
  uint32_t n_lights, n_draws;
  const srz_light *lights;
  const srz_mesh_draw *draws; /* in mesh registration order = batch order */
  uint32_t flags, _pad;
} srz_scene_frame;

typedef struct srz_ctx srz_ctx;
typedef struct srz_frameset srz_frameset;

/* ---- lifetime ------------------------------------------------------------------------- */
int srz_abi_version(void);
/* device_id: HIP ordinal. Replaces the TraditionalRasterizer(w,h) construction of device state. */
int srz_create(srz_ctx **out, int device_id);
void srz_destroy(srz_ctx *ctx);
const char *srz_last_error(const srz_ctx *ctx); /* ctx may be NULL: last error of srz_create */

/* Per-ctx switches (value 0 / 1), applied to framesets created AFTERWARDS:
 *   SRZ_OPT_POOL_LAZY  1 = creating a set does NOT size its tile-list pool by a binning pass of its own (see srz_frameset_render:
 *                      bands whose lists do not fit take the ordered rasteriser until a later render has grown the pool).
 *                      Default 0, or 1 when the environment variable SRZ_POOL_LAZY is set — read ONCE, in srz_create. */
#define SRZ_OPT_POOL_LAZY 1
/*   SRZ_OPT_APPROX_SHADE  1 = TOLERANCE MODE of the fragment shaders (default 0 = exact: bit-identical to the CPU oracle).  The
 *                      reference's x86 path shades with approximate instructions — _mm256_rcp_ps (include/shader/Shader.hpp:131,
 *                      src/Tools.cpp:19, include/loader/TextureLoader.hpp:99, src/Rasterizer.cpp:111) and SVML _mm256_pow_ps
 *                      (include/shader/Shader.hpp:195); this switch gives the colour arithmetic the same class on gfx950 (v_rcp_f32 /
 *                      v_rsq_f32 / v_sqrt_f32 at 1 ulp, x^p = exp2(p log2 x)).  Depth, coverage and ownership stay bit-exact (the
 *                      rasteriser does not change); colours: 8-wide ("V") columns within 0.5 of the exact value on the 0..255 scale,
 *                      scalar-tail ("S") columns equal except where the value in front of the floor lies within 1e-3 of an integer
 *                      (tests/test_gpu_approx.py).  Frames with 1..4 lights and no BUMP / DISPLACEMENT batch take it; others stay exact. */
#define SRZ_OPT_APPROX_SHADE 2
int srz_set_option(srz_ctx *ctx, int option, int value);

/* Multi-GPU: this ctx owns the 32-row bands b with b % world == rank (local band b / world).
 * Default (0,1) = whole frame. Affects srz_frameset_* only. */
int srz_set_shard(srz_ctx *ctx, int rank, int world);

/* ---- texture = TextureLoader's cv::Mat (BGR u8, row-major, top row first;
 *      src/TextureLoader.cpp:3-12).  row_stride in bytes. Slots 0..63. ------------------- */
int srz_texture_upload(srz_ctx *ctx, int tex_id, const uint8_t *bgr, int w, int h, int row_stride);

/* ---- mesh = Mesh::vertices + Mesh::faces (include/object/Mesh.hpp:52-57), slots 0..255; faces = 3 indices each ---- */
int srz_mesh_upload(srz_ctx *ctx, int mesh_id, const srz_vertex *verts, uint32_t n_verts, const uint32_t *faces,
                    uint32_t n_faces);

/* ---- draw = TraditionalRasterizer::draw(TRIANGLES) for one scene ----------------------
 * z/c0/c1/c2: W*H floats each, in/out (m_zBuffer, m_channels[0..2]); draw never clears unless
 * SRZ_FUSED_CLEAR.  primitive: LINES is accepted and rasterised as triangles exactly like the
 * reference (src/Rasterizer.cpp:185-189; rasterizeWireframe is never called); anything else →
 * SRZ_E_PRIMITIVE.  stats may be NULL. */
int srz_draw(srz_ctx *ctx, int primitive, const srz_frame *frame, float *z, float *c0, float *c1,
             float *c2, srz_stats *stats);

/* Batch form of srz_draw (host buffers): n frames of one size in one launch set.  planes[f] points to 4*W*H floats of
 * frame f laid out [z | c0 | c1 | c2], in/out like srz_draw's four pointers.  stats (optional) = sums over the batch.
 * The batch is rendered in pieces of ~128 MB of planes: piece k is read back on a second stream while piece k + 1 renders (and
 * uploads its in/out planes); planes registered with srz_host_register move by DMA.  SRZ_NO_Z_READBACK (per frame) skips the depth
 * plane's download.  For data that stays on the device use the frameset calls below. */
int srz_draw_batch(srz_ctx *ctx, int primitive, const srz_frame *frames, int n_frames, float *const *planes,
                   srz_stats *stats);

/* ---- throughput mode: frames resident in HBM ------------------------------------------
 * A frameset copies n frames (same width/height) to the device once.  srz_frameset_render
 * rasterises + shades all of them into ONE device buffer laid out
 *     [frame][plane: z,c0,c1,c2][local_rows][width]   (float32)
 * where local_rows = srz_frameset_local_rows() (= height for an unsharded ctx, else
 * bands_per_rank*32, zero-padded).  d_out is a DEVICE pointer (e.g. a torch tensor's data_ptr),
 * stream a hipStream_t (NULL = the ctx's own non-blocking stream; pass SRZ_STREAM_NULL for HIP's null stream:
 * work on the ctx's stream is NOT ordered against the null stream).  The call is asynchronous on that stream, the first render of
 * a set included: srz_frameset_create / srz_sceneset_create (synchronous anyway: they upload) run a binning pass of their own and
 * size the tile-list pool by its count of (triangle, tile) pairs, so a set is rendered by the fast path as a whole from its first
 * render on.  Afterwards the pool follows the previous renders'
 * demand (a sceneset's geometry may change with srz_sceneset_update); growth — rare — is the one place a render waits for the
 * device (hipDeviceSynchronize + hipMalloc), and the bands that did not fit take the ordered rasteriser in the render that found out.
 * srz_set_option(ctx, SRZ_OPT_POOL_LAZY, 1) before creating the set skips the creation-time pass (the pool then only follows the
 * renders' demand). */
#define SRZ_STREAM_NULL ((void *)(intptr_t)-1)
int srz_frameset_create(srz_ctx *ctx, const srz_frame *frames, int n_frames, srz_frameset **out);
/* Same, but the frames are given as meshes + matrices: every srz_frameset_render first runs the vertex stage on the
 * device (k_vertex) to produce the post-MVP stream, i.e. it times the reference's whole draw(). */
int srz_sceneset_create(srz_ctx *ctx, const srz_scene_frame *frames, int n_frames, srz_frameset **out);
/* draw() for one scene with the vertex stage on the device (host planes in/out, like srz_draw) */
int srz_draw_scene(srz_ctx *ctx, int primitive, const srz_scene_frame *frame, float *z, float *c0, float *c1, float *c2,
                   srz_stats *stats);
void srz_frameset_destroy(srz_ctx *ctx, srz_frameset *fs);
int srz_frameset_local_rows(const srz_ctx *ctx, const srz_frameset *fs);
size_t srz_frameset_out_bytes(const srz_ctx *ctx, const srz_frameset *fs);
int srz_frameset_render(srz_ctx *ctx, srz_frameset *fs, void *d_out, size_t out_bytes,
                        uint32_t flags, void *stream);
/* display()'s resolve on the device (cv::merge + convertTo(CV_8UC3), src/Render.cpp:61-62): the three colour planes of
 * a rendered buffer (layout of srz_frameset_render) → interleaved 8-bit [frame][local_rows][width][3], round half to even,
 * saturate.  Asynchronous on `stream`.  Any width (sizes whose plane is not a multiple of 4 pixels take a one-pixel-per-thread kernel). */
int srz_frameset_resolve8(srz_ctx *ctx, const srz_frameset *fs, const void *d_planes, void *d_bgr8, size_t bgr8_bytes,
                          void *stream);
/* Re-upload the per-frame data of a sceneset (matrices, eye, lights, shader constants, flags, shader/texture per draw)
 * without re-allocating anything.  The structure must be unchanged: same frame count and size, same mesh slots and face
 * counts per draw, same light counts; otherwise SRZ_E_INVALID (create a new set).  The upload is ONE asynchronous copy
 * on the context's own stream (ordered against srz_target_draw / renders submitted there, no host synchronisation);
 * renders of this set submitted on a caller-provided stream must be ordered against it by the caller. */
int srz_sceneset_update(srz_ctx *ctx, srz_frameset *fs, const srz_scene_frame *frames, int n_frames);

/* ---- multi-GPU exchange: the band shards of every rank → full row-major frames on every rank ------------------------
 * One process per GPU.  Rank 0 makes an id (srz_comm_unique_id), the host program hands the 128 bytes to every rank
 * (MPI, torch.distributed, a file), and every rank calls srz_comm_create, which builds the RCCL communicator (librccl is
 * loaded on first use) and sets the ctx's shard like srz_set_shard(ctx, rank, world).
 * srz_frameset_allgather = ncclAllGather over xGMI of this rank's shard (what srz_frameset_render / _resolve8 wrote:
 * [frame][4 planes | 1][local_rows][W x 4 | W x 3 bytes]) into d_gathered (world x shard bytes), then one HIP pass that
 * de-interleaves the round-robin bands into d_full = [frame][planes][bands_per_rank*world*32 rows][row] (rows >= height
 * are padding).  Asynchronous on `stream`; to overlap the exchange of step k with the render of step k+1 give the two
 * different streams and shard buffers and order them with events (bench.py does).  srz_frameset_deinterleave is the second
 * half alone (for a host program that brings its own collective). */
#define SRZ_EXCHANGE_PLANES 0 /* the 4 float planes, 16 bytes per pixel */
#define SRZ_EXCHANGE_BGR8 1   /* display()'s resolved image, 3 bytes per pixel */
typedef struct srz_comm srz_comm;
int srz_comm_unique_id(uint8_t *out128);
int srz_comm_create(srz_ctx *ctx, const uint8_t *id128, int rank, int world, srz_comm **out);
void srz_comm_destroy(srz_ctx *ctx, srz_comm *comm);
size_t srz_frameset_exchange_bytes(const srz_ctx *ctx, const srz_frameset *fs, int what); /* bytes of this rank's shard */
int srz_frameset_allgather(srz_ctx *ctx, srz_comm *comm, const srz_frameset *fs, const void *d_shard, void *d_gathered,
                           void *d_full, int what, void *stream);
int srz_frameset_deinterleave(srz_ctx *ctx, const srz_frameset *fs, const void *d_gathered, void *d_full, int what,
                              void *stream);
/* The exchange WITHOUT a second pass.  The rank renders (srz_frameset_render) or resolves (srz_frameset_resolve8) its shard
 * directly at  d_gathered + rank * srz_frameset_exchange_bytes()  and this call is one IN-PLACE ncclAllGather that fills in the
 * other ranks' shards around it: no staging copy, no de-interleave kernel, nothing but the xGMI transfers.  The result stays in
 * the all-gather's own order, "rank-major shards":
 *     [rank][frame][plane: z,c0,c1,c2 | 1][bands_per_rank * 32 rows][row bytes]
 * row y of a frame lives in the shard of rank (y / 32) % world at local row (y / 32 / world) * 32 + y % 32:
 * srz_frameset_gathered_row_offset() returns that row's byte offset in d_gathered, and srz_frameset_read_gathered_frame() brings
 * one frame to the host as row-major planes ([4][H][W] float, or [H][W][3] bytes for SRZ_EXCHANGE_BGR8), de-interleaving in the
 * device→host copy itself (one strided copy per rank and plane).  A device-side consumer that needs row-major planes uses
 * srz_frameset_allgather (all-gather + one HIP pass) instead.  With world = 1 the call does nothing. */
int srz_frameset_allgather_inplace(srz_ctx *ctx, srz_comm *comm, const srz_frameset *fs, void *d_gathered, int what, void *stream);
size_t srz_frameset_gathered_row_offset(const srz_ctx *ctx, const srz_frameset *fs, int what, int frame, int plane, int row);
int srz_frameset_read_gathered_frame(srz_ctx *ctx, const srz_frameset *fs, const void *d_gathered, int what, int frame,
                                     void *host_out, void *stream);

/* ---- device-resident framebuffer = RenderingPipeline's m_zBuffer + m_channels kept in HBM between calls ----------
 * clear(Color|Depth) immediately followed by a draw costs nothing (fused into the raster kernel); planes come back to
 * the host only when asked for, and display() only needs the 3-byte resolved image. */
typedef struct srz_target srz_target;
int srz_target_create(srz_ctx *ctx, int width, int height, srz_target **out); /* starts cleared (z=+inf, colour 0) */
void srz_target_destroy(srz_ctx *ctx, srz_target *t);
int srz_target_clear(srz_ctx *ctx, srz_target *t, int color, int depth);      /* = RenderingPipeline::clear(Buffers) */
/* draw frame 0 of a 1-frame set (srz_frameset_create / srz_sceneset_create of the target's size) into the target */
int srz_target_draw(srz_ctx *ctx, srz_target *t, int primitive, srz_frameset *fs, srz_stats *stats);
int srz_target_read(srz_ctx *ctx, srz_target *t, float *z, float *c0, float *c1, float *c2); /* NULL planes are skipped */
int srz_target_read_bgr8(srz_ctx *ctx, srz_target *t, uint8_t *bgr8);         /* display()'s resolve, W*H*3 bytes */
/* synchronous: runs the counting variant of the kernels once and returns the counters */
int srz_frameset_stats(srz_ctx *ctx, srz_frameset *fs, srz_stats *stats);
/* Algorithmic bytes of one render of the frameset on this ctx (SURVEY §8d / DESIGN.md):
 * 16*W*local_rows + 96*N_tri + 24*N_lights + B_tex per frame. n_shaded_tex = texture-shaded pixels. */
uint64_t srz_frameset_algorithmic_bytes(const srz_ctx *ctx, const srz_frameset *fs);
/* Average device time (ms) per srz_frameset_render since the last call with reset!=0, measured with hipEvents on
 * the launch stream: ms4[0] = setup+binning kernels, ms4[1] = raster kernel (visibility), ms4[2] = shade kernel up to
 * the join with the clear kernel that runs beside both on a second stream, ms4[3] = whole pipeline. */
int srz_kernel_time_ms(srz_ctx *ctx, int reset, double *ms4, int *launches);
/* ms4[3] of each timed render since the last reset, in submission order (at most cap values, *n = how many): the
 * per-step distribution (p10 / median / p90) bench.py prints; *span_ms (may be NULL) = from the first of them starting
 * to the last of them ending — renders submitted to different streams overlap, the span is what they took together.
 * Call it before the resetting srz_kernel_time_ms. */
int srz_kernel_time_samples(srz_ctx *ctx, float *out, int cap, int *n, double *span_ms);
/* enabled: 0 off; 1 the whole launch set only (ms4[3]; two events per render); 2 also the three groups (four events per
 * render — every event is a barrier in the launch stream, ≈4 µs each, so throughput is measured with 1 and broken down with 2) */
int srz_set_kernel_timing(srz_ctx *ctx, int enabled);
int srz_sync(srz_ctx *ctx);
/* Page-lock `bytes` of host memory at `ptr` / release it again (hipHostRegister / hipHostUnregister behind the C ABI, so that a
 * binding needs no HIP header).  The host-buffer entry points — srz_draw, srz_draw_scene, srz_draw_batch — move planes that lie in a
 * registered range by DMA at the link's rate instead of through the runtime's pageable staging copies (MI355X, 1024^2: see
 * DESIGN.md §5 "PCIe-inclusive rate").  Register a buffer once after allocating it (the reference's m_zBuffer / m_channels live as
 * long as the pipeline object), unregister it before freeing it.  What these calls replace on the reference's side: nothing — its
 * framebuffer never leaves host memory (src/Render.cpp:46-64); this is the price of the seam at src/Rasterizer.cpp:183-240. */
int srz_host_register(srz_ctx *ctx, void *ptr, size_t bytes);
int srz_host_unregister(srz_ctx *ctx, void *ptr);
/* self-check of the device arithmetic: compares the kernels' short exact reciprocal / square-root sequences with the
 * IEEE expansions on all 2^32 binary32 operands. out4 = {operands on the fast path, rcp, sqrt, 1/sqrt mismatches} */
int srz_verify_fastmath(srz_ctx *ctx, uint64_t *out4);
/* Same for the division-by-reciprocal sequence of the scalar-tail shaders (texel / 255, intensity / distance):
 * out3 = { pairs tested, mismatches vs a / b on pseudo-random in-range pairs, mismatches of texel / 255 for texel 0..255 }. */
int srz_verify_fastdiv(srz_ctx *ctx, uint64_t *out3);
/* Same for the optimistic x^p of the shading builds for non-integer exponents (exp2(p log2 x) in binary64 with a rounding-safety
 * flag): every binary32 x in [2^-40, 1] at exponent p (a non-integer in (0, 4096]) against the correctly rounded path.
 * out4 = { operands, results that differ although the flag was clear (must be 0), flagged operands with a normal result >= 2^-120
 * (ambiguous roundings), flagged operands with a smaller result } — the tiles of flagged pixels are shaded by the generic build. */
int srz_verify_fastpow(srz_ctx *ctx, float p, uint64_t *out4);
/* Same for the attenuation distance of the scalar Blinn-Phong, sqrt(dx^2 + dy^2) evaluated in binary64 and rounded once
 * (src/Shader.cpp:516-523 of the reference): the FAST builds take two binary64 Heron steps from the binary32 root instead of the
 * binary64 square root, with a flag when the result lies within 8 binary64 ulps of a binary32 rounding boundary.  4.3e9 pseudo-random
 * pairs (random / close exponents / few significant bits / scaled Pythagorean pairs whose root IS a rounding boundary):
 * out5 = { pairs, results that differ although the flag was clear (must be 0), flagged random pairs, flagged Pythagorean pairs,
 * flagged few-significant-bit pairs }. */
int srz_verify_fastlen(srz_ctx *ctx, uint64_t *out5);
/* diagnostic only (tests): counters the LAST render of the set left — out6 = { tiles taken by the ordered rasteriser, tiles the FAST
 * shading builds handed to the generic build, capacity of a tile-list sub-pool, largest demand a sub-pool reported, workgroups of the
 * side-stream clear (a batch-sized set measures them on the device within its first 24 renders, and again every 4096), 1 once that measurement has been taken }; waits for the device */
int srz_frameset_debug_counters(srz_ctx *ctx, srz_frameset *fs, uint32_t *out6);
/* diagnostic only: raw device counters of the last stats run (layout = csrc/srz_device.h ST_*); returns their count */
int srz_debug_counters(srz_ctx *ctx, uint64_t *out, int n);

#ifdef __cplusplus
}
#endif
#endif /* SRZ_H_ */
