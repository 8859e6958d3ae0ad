/* CPU-only: the oracle (oracle/srz_oracle.c) under AddressSanitizer + UBSan on seeded adversarial frames — odd sizes, every
 * shader type incl. BUMP / DISPLACEMENT on a 3 x 5 texture, 0..5 lights, triangles that are degenerate, far off screen, huge,
 * non-finite, exactly on pixel centres.  Built by `make -C oracle asan`, run by tests/test_host_fuzz.py.  The oracle is test
 * infrastructure; this only checks that the checker itself reads and writes inside its buffers.
 *   oracle_asan <frames> <seed>     prints "frames=<n> visible=<sum>" and exits 0 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/srz.h"

int orc_draw(int primitive, const srz_frame *fr, float *z, float *c0, float *c1, float *c2, srz_stats *st);
int orc_draw_omp(const srz_frame *fr, float *z, float *c0, float *c1, float *c2, int band, int *threads_used, int num_threads);
int orc_draw_rows(const srz_frame *fr, float *z, float *c0, float *c1, float *c2, int row0, int row1);
int orc_texture_set(int tex_id, const uint8_t *bgr, int w, int h, int row_stride);
void orc_resolve8(int W, int H, const float *c0, const float *c1, const float *c2, uint8_t *bgr8);

static uint64_t st;
static uint64_t rnd(void) {
  st ^= st << 13, st ^= st >> 7, st ^= st << 17;
  return st;
}
static float unit(void) { return (float)(rnd() >> 40) / 16777216.0f; }
static float coord(int size) {
  switch (rnd() % 12) {
  case 0: return -1e6f * unit();
  case 1: return 1e6f * unit();
  case 2: return (float)(rnd() % (uint64_t)(size + 1));        /* exactly on a pixel corner / centre */
  case 3: return NAN;
  case 4: return (rnd() & 1) ? INFINITY : -INFINITY;
  case 5: return -0.0f;
  default: return unit() * (float)size * 1.2f - 0.1f * (float)size;
  }
}

int main(int argc, char **argv) {
  const int n_frames = argc > 1 ? atoi(argv[1]) : 40;
  st = 0x2545f4914f6cdd1dull ^ (uint64_t)(argc > 2 ? atoll(argv[2]) : 1);
  uint8_t tex[3 * 5 * 3];
  for (size_t i = 0; i < sizeof tex; ++i) tex[i] = (uint8_t)rnd();
  if (orc_texture_set(0, tex, 3, 5, 9) != SRZ_OK) return 1;
  unsigned long long visible = 0;
  for (int f = 0; f < n_frames; ++f) {
    const int W = 1 + (int)(rnd() % 97), H = 1 + (int)(rnd() % 71);
    const uint32_t n_batches = 1 + (uint32_t)(rnd() % 4), n_lights = (uint32_t)(rnd() % 6);
    srz_light lights[6];
    for (uint32_t l = 0; l < n_lights; ++l)
      for (int k = 0; k < 3; ++k) lights[l].pos[k] = coord(W), lights[l].intensity[k] = 500.0f * unit();
    srz_batch batches[4];
    srz_tri *tris[4];
    for (uint32_t b = 0; b < n_batches; ++b) {
      const uint32_t n = (uint32_t)(rnd() % 40);
      tris[b] = (srz_tri *)malloc(sizeof(srz_tri) * (n ? n : 1));
      for (uint32_t t = 0; t < n; ++t) {
        srz_tri *q = &tris[b][t];
        const int wild = (int)(rnd() % 4) == 0;
        const float cx = unit() * (float)W, cy = unit() * (float)H, r = 1.0f + unit() * 30.0f;
        for (int v = 0; v < 3; ++v) {
          q->pos[v][0] = wild ? coord(W) : cx + (unit() - 0.5f) * r, q->pos[v][1] = wild ? coord(H) : cy + (unit() - 0.5f) * r;
          q->pos[v][2] = wild ? coord(50) : 0.1f + unit() * 49.0f;
          for (int k = 0; k < 3; ++k) q->nrm[v][k] = (rnd() % 16 == 0) ? 0.0f : unit() * 2.0f - 1.0f;
          q->uv[v][0] = (rnd() % 8 == 0) ? coord(2) : unit(), q->uv[v][1] = (rnd() % 8 == 0) ? 1.0f : unit();
        }
        if (rnd() % 10 == 0) memcpy(q->pos[1], q->pos[0], sizeof q->pos[0]); /* degenerate */
      }
      batches[b].shader = (int32_t)(rnd() % 5), batches[b].tex_id = 0, batches[b].n_tris = n, batches[b]._pad = 0, batches[b].tris = tris[b];
    }
    srz_frame fr;
    memset(&fr, 0, sizeof fr);
    fr.width = W, fr.height = H;
    for (int k = 0; k < 3; ++k) fr.eye[k] = coord(W), fr.ka[k] = 0.005f, fr.ks[k] = 0.7937f;
    static const float ps[] = {150.0f, 0.0f, 1.0f, 7.5f, 3000.0f};
    fr.p = ps[rnd() % 5], fr.kh = 0.2f, fr.kn = 0.1f;
    fr.n_lights = n_lights, fr.n_batches = n_batches, fr.lights = lights, fr.batches = batches;
    fr.flags = (rnd() & 1) ? SRZ_FUSED_CLEAR : 0u;
    if (rnd() % 8 == 0) fr.flags |= SRZ_UNIFIED;
    const size_t px = (size_t)W * H;
    float *pl = (float *)malloc(sizeof(float) * 4 * px); /* exactly W*H per plane: ASan sees any stray access */
    float *z = pl, *c0 = pl + px, *c1 = pl + 2 * px, *c2 = pl + 3 * px;
    for (size_t i = 0; i < px; ++i) z[i] = (rnd() % 50 == 0) ? NAN : INFINITY, c0[i] = c1[i] = c2[i] = 0.0f;
    srz_stats s;
    int used = 0;
    int rc = orc_draw(SRZ_PRIMITIVE_TRIANGLES, &fr, z, c0, c1, c2, &s);
    if (rc == SRZ_OK) visible += s.visible;
    rc |= orc_draw_omp(&fr, z, c0, c1, c2, 8, &used, 2) < 0;
    rc |= orc_draw_rows(&fr, z, c0, c1, c2, H / 3, H);
    uint8_t *img = (uint8_t *)malloc(3 * px);
    orc_resolve8(W, H, c0, c1, c2, img);
    free(img), free(pl);
    for (uint32_t b = 0; b < n_batches; ++b) free(tris[b]);
    if (rc != SRZ_OK) {
      fprintf(stderr, "oracle_asan: frame %d returned %d\n", f, rc);
      return 1;
    }
  }
  printf("frames=%d visible=%llu\n", n_frames, visible);
  return 0;
}
