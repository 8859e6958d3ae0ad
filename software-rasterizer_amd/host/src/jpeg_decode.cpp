// jpeg_decode.cpp — JPEG (JFIF) → what cv::imread(path) gives the reference's TextureLoader (src/TextureLoader.cpp:3-12): 8-bit BGR,
// top row first.  The reference ships one JPEG on the raster path, the height map of the bump / displacement shaders
// (examples/models/spot/hmap.jpg: 800 x 800, PROGRESSIVE, 4:2:0), and reads it through OpenCV, i.e. through libjpeg(-turbo) with its
// default settings.  OpenCV is absent from this image, so the decoder is restated here from the published algorithms:
//
//   * ITU T.81 (ISO 10918-1) baseline / extended sequential (SOF0, SOF1) and progressive (SOF2) Huffman decoding, 8-bit samples,
//     1 or 3 components (grey, YCbCr; Adobe transform 0 = RGB), restart intervals, any sampling factors with ratios 1 or 2;
//   * the inverse DCT of libjpeg's default method JDCT_ISLOW (jidctint: 13-bit constants, two passes), bit for bit, including its
//     range-limit table's wrap-around;
//   * libjpeg's default "fancy" chroma upsampling (triangle filter: h2v1, h2v2, h1v2; plain replication where libjpeg falls back to
//     it) and its fixed-point YCbCr → RGB tables (16 fractional bits).
// Parity: tests/test_host_layer.py compares the decoded hmap.jpg and a set of synthetic files (baseline / progressive, 4:4:4 / 4:2:2 /
// 4:2:0 / grey, odd sizes, restart markers) with libjpeg-turbo's own output as Pillow returns it in this image (fixtures made
// by tests/golden/make_jpeg_golden.py).  Not handled (std::runtime_error like an empty cv::Mat): arithmetic coding, lossless /
// hierarchical, 12-bit, CMYK; EXIF orientation is ignored (cv::imread would rotate).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace SoftRasterizer {
namespace detail {
namespace {

const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  bool defined = false;
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  // canonical decoding (T.81 Annex F.2.2.3): per code length the largest code, the first code's index
  int32_t maxcode[18], valptr[17], mincode[17];
  uint8_t look_len[512], look_val[512]; // 9-bit look-ahead: code length (0 = longer) and symbol
  bool build() { // false: more codes of some length than a prefix code has room for (a corrupt table)
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k, mincode[l] = code;
      code += bits[l], k += bits[l];
      if (code > (1 << l)) return false;
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    std::memset(look_len, 0, sizeof look_len);
    code = 0, k = 0;
    for (int l = 1; l <= 9; ++l) {
      for (int i = 0; i < bits[l]; ++i, ++k, ++code)
        for (int f = 0; f < (1 << (9 - l)); ++f) look_len[(code << (9 - l)) | f] = (uint8_t)l, look_val[(code << (9 - l)) | f] = vals[k];
      code <<= 1;
    }
    return true;
  }
};

struct Comp {
  int id = 0, h = 1, v = 1, tq = 0;
  int bw = 0, bh = 0;     // blocks per row / column, padded to whole MCUs of an interleaved scan
  int cw = 0, ch = 0;     // samples: ceil(W * h / hmax), ceil(H * v / vmax)
  int dc_tbl = 0, ac_tbl = 0, dc_pred = 0;
  std::vector<int16_t> coef; // bw * bh * 64, natural order
  std::vector<uint8_t> pix;  // bw * 8 x bh * 8 samples after the inverse DCT
};

struct Bits { // entropy-coded segment reader: byte stuffing (FF 00), stops at a marker (then feeds zeros, as libjpeg does)
  const uint8_t *p, *end;
  uint32_t acc = 0;
  int n = 0;
  int marker = 0;
  void fill() {
    while (n <= 24) {
      uint32_t b = 0;
      if (!marker && p < end) {
        b = *p++;
        if (b == 0xff) {
          while (p < end && *p == 0xff) ++p; // fill bytes
          const uint32_t m = p < end ? *p++ : 0xd9;
          if (m != 0) marker = (int)m, b = 0;
        }
      }
      acc |= b << (24 - n);
      n += 8;
    }
  }
  int peek(int k) { // k <= 16
    if (n < k) fill();
    return (int)(acc >> (32 - k));
  }
  void skip(int k) { acc <<= k, n -= k; }
  int get(int k) {
    if (k == 0) return 0;
    const int v = peek(k);
    skip(k);
    return v;
  }
  int bit() { return get(1); }
  void reset() { acc = 0, n = 0, marker = 0; }
};

inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; } // T.81 F.2.2.1

struct Decoder {
  const std::vector<uint8_t> &d;
  const std::string &path;
  int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1;
  bool progressive = false, adobe = false;
  int n_scans = 0;
  int adobe_transform = -1, restart_interval = 0;
  uint16_t qt[4][64];
  bool qt_ok[4] = {false, false, false, false};
  Huff hdc[4], hac[4];
  Comp comp[4];
  int eobrun = 0;
  Decoder(const std::vector<uint8_t> &data, const std::string &pth) : d(data), path(pth) {}

  [[noreturn]] void fail(const char *why) const { throw std::runtime_error("Cannot open file (" + std::string(why) + "): " + path); }

  int decode_sym(Bits &b, const Huff &h) {
    const int look = b.peek(9);
    if (h.look_len[look]) {
      b.skip(h.look_len[look]);
      return h.look_val[look];
    }
    int code = b.peek(16), l = 10;
    for (; l <= 16; ++l)
      if ((code >> (16 - l)) <= h.maxcode[l]) break;
    if (l > 16) return 0; // corrupt data: libjpeg warns and returns 0
    b.skip(l);
    return h.vals[(h.valptr[l] + (code >> (16 - l)) - h.mincode[l]) & 255];
  }

  void parse() {
    size_t p = 2;
    if (d.size() < 4 || d[0] != 0xff || d[1] != 0xd8) fail("not a JPEG");
    bool have_sof = false;
    while (p + 4 <= d.size()) {
      if (d[p] != 0xff) {
        ++p;
        continue;
      }
      const int m = d[p + 1];
      if (m == 0xff) {
        ++p;
        continue;
      }
      p += 2;
      if (m == 0xd9) break;                                  // EOI
      if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;   // TEM, stray RSTn
      if (p + 2 > d.size()) break;
      const size_t len = (size_t)d[p] << 8 | d[p + 1];
      if (len < 2 || p + len > d.size()) fail("truncated segment");
      const uint8_t *s = &d[p + 2];
      const size_t n = len - 2;
      if (m == 0xdb) { // DQT
        size_t i = 0;
        while (i < n) {
          const int pq = s[i] >> 4, tq = s[i] & 15;
          ++i;
          if (tq > 3 || i + (pq ? 128 : 64) > n) fail("bad DQT");
          for (int k = 0; k < 64; ++k) {
            qt[tq][ZIGZAG[k]] = pq ? (uint16_t)(s[i] << 8 | s[i + 1]) : s[i];
            i += pq ? 2 : 1;
          }
          qt_ok[tq] = true;
        }
      } else if (m == 0xc4) { // DHT
        size_t i = 0;
        while (i + 17 <= n) {
          const int tc = s[i] >> 4, th = s[i] & 15;
          if (tc > 1 || th > 3) fail("bad DHT");
          Huff &h = tc ? hac[th] : hdc[th];
          int cnt = 0;
          for (int l = 1; l <= 16; ++l) h.bits[l] = s[i + l], cnt += s[i + l];
          i += 17;
          if (cnt > 256 || i + cnt > n) fail("bad DHT");
          std::memcpy(h.vals, s + i, cnt);
          i += cnt;
          h.defined = true;
          if (!h.build()) fail("bad DHT");
        }
      } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) { // SOF0 / 1 / 2
        if (have_sof) fail("two frame headers");
        if (n < 6 || s[0] != 8) fail("only 8-bit samples");
        H = s[1] << 8 | s[2], W = s[3] << 8 | s[4], ncomp = s[5];
        // (bounds the allocations below — coefficients, planes and the image together take ~15 bytes per pixel: at most 1 GB — and,
        // with the scan limit, the time a crafted progressive file can cost)
        if (W <= 0 || H <= 0 || W > 32768 || H > 32768 || (size_t)W * (size_t)H > ((size_t)1 << 26)) fail("bad size");
        if ((ncomp != 1 && ncomp != 3) || n < 6 + 3 * (size_t)ncomp) fail("only 1 or 3 components");
        for (int c = 0; c < ncomp; ++c) {
          comp[c].id = s[6 + 3 * c], comp[c].h = s[7 + 3 * c] >> 4, comp[c].v = s[7 + 3 * c] & 15, comp[c].tq = s[8 + 3 * c] & 3;
          if (comp[c].h < 1 || comp[c].h > 4 || comp[c].v < 1 || comp[c].v > 4) fail("bad sampling factors");
          hmax = std::max(hmax, comp[c].h), vmax = std::max(vmax, comp[c].v);
        }
        progressive = m == 0xc2;
        have_sof = true;
        const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        for (int c = 0; c < ncomp; ++c) {
          Comp &k = comp[c];
          k.bw = mcux * k.h, k.bh = mcuy * k.v;
          k.cw = (W * k.h + hmax - 1) / hmax, k.ch = (H * k.v + vmax - 1) / vmax;
          k.coef.assign((size_t)k.bw * k.bh * 64, 0);
        }
      } else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) {
        fail("lossless / hierarchical / arithmetic-coded JPEG");
      } else if (m == 0xdd) { // DRI
        if (n >= 2) restart_interval = s[0] << 8 | s[1];
      } else if (m == 0xee) { // APP14 "Adobe"
        if (n >= 12 && !std::memcmp(s, "Adobe", 5)) adobe = true, adobe_transform = s[11];
      } else if (m == 0xda) { // SOS
        if (!have_sof) fail("scan before the frame header");
        p = scan(p + len, s, n);
        continue;
      }
      p += len;
    }
    if (!have_sof) fail("no frame header");
  }

  // one scan: header at s, entropy-coded data from `at` → the offset behind it
  size_t scan(size_t at, const uint8_t *s, size_t n) {
    if (++n_scans > 1024) fail("too many scans"); // (libjpeg-turbo's own guard against progressive files made of empty scans)
    if (n < 1) fail("bad SOS");
    const int ns = s[0];
    if (ns < 1 || ns > ncomp || n < 1 + 2 * (size_t)ns + 3) fail("bad SOS");
    int ci[4];
    for (int i = 0; i < ns; ++i) {
      int c = 0;
      while (c < ncomp && comp[c].id != s[1 + 2 * i]) ++c;
      if (c == ncomp) fail("scan names an unknown component");
      ci[i] = c;
      comp[c].dc_tbl = s[2 + 2 * i] >> 4, comp[c].ac_tbl = s[2 + 2 * i] & 15;
      if (comp[c].dc_tbl > 3 || comp[c].ac_tbl > 3) fail("bad table selector");
    }
    const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
    if (progressive) {
      if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13) fail("bad progressive scan parameters");
    } else if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) {
      fail("bad sequential scan parameters");
    }
    Bits b{&d[0] + at, &d[0] + d.size()};
    for (int i = 0; i < ns; ++i) comp[ci[i]].dc_pred = 0;
    eobrun = 0;
    // the scan's MCUs: interleaved = the frame's MCU grid; one component = its own blocks, ceil(samples / 8) each way
    int mx, my;
    if (ns > 1) {
      mx = (W + 8 * hmax - 1) / (8 * hmax), my = (H + 8 * vmax - 1) / (8 * vmax);
    } else {
      mx = (comp[ci[0]].cw + 7) / 8, my = (comp[ci[0]].ch + 7) / 8;
    }
    int until_restart = restart_interval, next_rst = 0;
    for (int y = 0; y < my; ++y)
      for (int x = 0; x < mx; ++x) {
        if (restart_interval && until_restart == 0) {
          // byte-align, take the RSTn marker (skipping whatever lies in front of it), reset the predictions
          b.n = 0, b.acc = 0;
          if (!b.marker) {
            while (b.p + 1 < b.end && !(b.p[0] == 0xff && b.p[1] >= 0xd0 && b.p[1] <= 0xd7)) ++b.p;
            if (b.p + 1 < b.end) b.p += 2;
          } else if (b.marker < 0xd0 || b.marker > 0xd7) {
            fail("restart marker expected");
          }
          b.marker = 0;
          next_rst = (next_rst + 1) & 7;
          for (int i = 0; i < ns; ++i) comp[ci[i]].dc_pred = 0;
          eobrun = 0;
          until_restart = restart_interval;
        }
        for (int i = 0; i < ns; ++i) {
          Comp &k = comp[ci[i]];
          const int nh = ns > 1 ? k.h : 1, nv = ns > 1 ? k.v : 1;
          for (int by = 0; by < nv; ++by)
            for (int bx = 0; bx < nh; ++bx) {
              int16_t *blk = &k.coef[((size_t)(y * nv + by) * k.bw + (x * nh + bx)) * 64];
              if (!progressive)
                block_sequential(b, k, blk);
              else if (Ss == 0)
                Ah == 0 ? block_dc_first(b, k, blk, Al) : block_dc_refine(b, blk, Al);
              else
                Ah == 0 ? block_ac_first(b, k, blk, Ss, Se, Al) : block_ac_refine(b, k, blk, Ss, Se, Al);
            }
        }
        --until_restart;
      }
    // behind the scan: the marker the reader stopped at, or the next one in the data
    if (b.marker) return (size_t)(b.p - &d[0]) - 2;
    size_t q = (size_t)(b.p - &d[0]);
    while (q + 1 < d.size() && !(d[q] == 0xff && d[q + 1] != 0 && d[q + 1] != 0xff && !(d[q + 1] >= 0xd0 && d[q + 1] <= 0xd7))) ++q;
    return q;
  }

  void block_sequential(Bits &b, Comp &k, int16_t *blk) {
    const Huff &dc = hdc[k.dc_tbl], &ac = hac[k.ac_tbl];
    if (!dc.defined || !ac.defined) fail("scan uses an undefined Huffman table");
    const int t = decode_sym(b, dc) & 15; // (a category above 11 only comes out of a corrupt table)
    const int diff = t ? extend(b.get(t), t) : 0;
    k.dc_pred += diff;
    blk[0] = (int16_t)k.dc_pred;
    for (int i = 1; i < 64;) {
      const int rs = decode_sym(b, ac), r = rs >> 4, s = rs & 15;
      if (s == 0) {
        if (r != 15) break; // EOB
        i += 16;
        continue;
      }
      i += r;
      if (i > 63) break;
      blk[ZIGZAG[i++]] = (int16_t)extend(b.get(s), s);
    }
  }
  void block_dc_first(Bits &b, Comp &k, int16_t *blk, int Al) {
    const Huff &dc = hdc[k.dc_tbl];
    if (!dc.defined) fail("scan uses an undefined Huffman table");
    const int t = decode_sym(b, dc) & 15;
    k.dc_pred += t ? extend(b.get(t), t) : 0;
    blk[0] = (int16_t)(k.dc_pred * (1 << Al));
  }
  void block_dc_refine(Bits &b, int16_t *blk, int Al) {
    if (b.bit()) blk[0] = (int16_t)(blk[0] | (1 << Al));
  }
  void block_ac_first(Bits &b, Comp &k, int16_t *blk, int Ss, int Se, int Al) {
    if (eobrun > 0) {
      --eobrun;
      return;
    }
    const Huff &ac = hac[k.ac_tbl];
    if (!ac.defined) fail("scan uses an undefined Huffman table");
    for (int i = Ss; i <= Se; ++i) {
      const int rs = decode_sym(b, ac), r = rs >> 4, s = rs & 15;
      if (s) {
        i += r;
        if (i > 63) break;
        blk[ZIGZAG[i]] = (int16_t)(extend(b.get(s), s) * (1 << Al));
      } else if (r == 15) {
        i += 15; // ZRL
      } else {
        eobrun = (1 << r) - 1 + (r ? b.get(r) : 0); // this block ends the band, eobrun more follow
        break;
      }
    }
  }
  // T.81 G.1.2.3 as libjpeg's decode_mcu_AC_refine does it
  void block_ac_refine(Bits &b, Comp &k, int16_t *blk, int Ss, int Se, int Al) {
    const int p1 = 1 << Al, m1 = -(1 << Al);
    const Huff &ac = hac[k.ac_tbl];
    if (!ac.defined) fail("scan uses an undefined Huffman table");
    auto correct = [&](int16_t &c) {
      if (b.bit() && (c & p1) == 0) c = (int16_t)(c >= 0 ? c + p1 : c + m1);
    };
    int i = Ss;
    if (eobrun == 0) {
      for (; i <= Se; ++i) {
        const int rs = decode_sym(b, ac);
        int r = rs >> 4, s = rs & 15;
        if (s) {
          s = b.bit() ? p1 : m1; // (a new coefficient is always +-1 at this bit position)
        } else if (r != 15) {
          eobrun = 1 << r;
          if (r) eobrun += b.get(r);
          break;
        }
        // over the already-nonzero coefficients (one correction bit each) and r zero ones
        for (; i <= Se; ++i) {
          int16_t &c = blk[ZIGZAG[i]];
          if (c != 0)
            correct(c);
          else if (--r < 0)
            break;
        }
        if (s && i <= Se) blk[ZIGZAG[i]] = (int16_t)s;
      }
    }
    if (eobrun > 0) {
      for (; i <= Se; ++i) {
        int16_t &c = blk[ZIGZAG[i]];
        if (c != 0) correct(c);
      }
      --eobrun;
    }
  }

  // ---- inverse DCT: libjpeg jidctint.c (JDCT_ISLOW), bit for bit ------------------------------------------------------------
  static inline uint8_t range_limit(int x) { // libjpeg's table: index (x & 1023) into [128..255 | 255 x 384 | 0 x 384 | 0..127]
    const int i = x & 1023;
    return (uint8_t)(i < 128 ? i + 128 : i < 512 ? 255 : i < 896 ? 0 : i - 896);
  }
  static void idct_islow(const int16_t *in, const uint16_t *q, uint8_t *out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr int64_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137,
                      F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
    auto descale = [](int64_t x, int n) { return (int)((x + ((int64_t)1 << (n - 1))) >> n); }; // (JLONG = 64 bits, as libjpeg on this platform)
    int ws[64];
    for (int c = 0; c < 8; ++c) {
      auto v = [&](int r) { return (int64_t)((int)in[r * 8 + c] * (int)q[r * 8 + c]); };
      int64_t z2 = v(2), z3 = v(6);
      int64_t z1 = (z2 + z3) * F0541;
      int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
      z2 = v(0), z3 = v(4);
      int64_t tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
      const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = v(7), tmp1 = v(5), tmp2 = v(3), tmp3 = v(1);
      z1 = tmp0 + tmp3, z2 = tmp1 + tmp2, z3 = tmp0 + tmp2;
      int64_t z4 = tmp1 + tmp3;
      const int64_t z5 = (z3 + z4) * F1175;
      tmp0 *= F0298, tmp1 *= F2053, tmp2 *= F3072, tmp3 *= F1501;
      z1 *= -F0899, z2 *= -F2562, z3 *= -F1961, z4 *= -F0390;
      z3 += z5, z4 += z5;
      tmp0 += z1 + z3, tmp1 += z2 + z4, tmp2 += z2 + z3, tmp3 += z1 + z4;
      ws[0 * 8 + c] = descale(tmp10 + tmp3, CB - P1), ws[7 * 8 + c] = descale(tmp10 - tmp3, CB - P1);
      ws[1 * 8 + c] = descale(tmp11 + tmp2, CB - P1), ws[6 * 8 + c] = descale(tmp11 - tmp2, CB - P1);
      ws[2 * 8 + c] = descale(tmp12 + tmp1, CB - P1), ws[5 * 8 + c] = descale(tmp12 - tmp1, CB - P1);
      ws[3 * 8 + c] = descale(tmp13 + tmp0, CB - P1), ws[4 * 8 + c] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; ++r) {
      const int *w = ws + r * 8;
      int64_t z2 = w[2], z3 = w[6];
      int64_t z1 = (z2 + z3) * F0541;
      int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
      int64_t tmp0 = ((int64_t)w[0] + w[4]) * (1 << CB), tmp1 = ((int64_t)w[0] - w[4]) * (1 << CB);
      const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = w[7], tmp1 = w[5], tmp2 = w[3], tmp3 = w[1];
      z1 = tmp0 + tmp3, z2 = tmp1 + tmp2, z3 = tmp0 + tmp2;
      int64_t z4 = tmp1 + tmp3;
      const int64_t z5 = (z3 + z4) * F1175;
      tmp0 *= F0298, tmp1 *= F2053, tmp2 *= F3072, tmp3 *= F1501;
      z1 *= -F0899, z2 *= -F2562, z3 *= -F1961, z4 *= -F0390;
      z3 += z5, z4 += z5;
      tmp0 += z1 + z3, tmp1 += z2 + z4, tmp2 += z2 + z3, tmp3 += z1 + z4;
      uint8_t *o = out + r * stride;
      constexpr int S = CB + P1 + 3;
      o[0] = range_limit(descale(tmp10 + tmp3, S)), o[7] = range_limit(descale(tmp10 - tmp3, S));
      o[1] = range_limit(descale(tmp11 + tmp2, S)), o[6] = range_limit(descale(tmp11 - tmp2, S));
      o[2] = range_limit(descale(tmp12 + tmp1, S)), o[5] = range_limit(descale(tmp12 - tmp1, S));
      o[3] = range_limit(descale(tmp13 + tmp0, S)), o[4] = range_limit(descale(tmp13 - tmp0, S));
    }
  }

  // ---- upsampling to W x H: libjpeg jdsample.c with do_fancy_upsampling (its default) ---------------------------------------
  // rows / columns beyond the component's real samples are never read: the edges replicate the last real sample, as libjpeg's
  // context rows and its first / last column special cases do
  void upsample(const Comp &k, std::vector<uint8_t> &out) const {
    out.assign((size_t)W * H, 0);
    const int stride = k.bw * 8, rh = hmax / k.h, rv = vmax / k.v;
    if (hmax % k.h || vmax % k.v || rh > 2 || rv > 2) fail("unsupported sampling ratio");
    const uint8_t *src = k.pix.data();
    auto row = [&](int y) { return src + (size_t)std::min(std::max(y, 0), k.ch - 1) * stride; };
    if (rh == 1 && rv == 1) {
      for (int y = 0; y < H; ++y) std::memcpy(&out[(size_t)y * W], row(y), W);
      return;
    }
    const bool fancy = k.cw > 2; // (libjpeg: downsampled_width > 2, else plain replication)
    std::vector<uint8_t> line((size_t)k.cw * 2 + 2);
    for (int y = 0; y < H; ++y) {
      const int sy = rv == 2 ? y >> 1 : y;
      if (!fancy) {
        const uint8_t *r = row(sy);
        for (int x = 0; x < W; ++x) out[(size_t)y * W + x] = r[rh == 2 ? x >> 1 : x];
        continue;
      }
      const uint8_t *r0 = row(sy);
      uint8_t *o = line.data();
      const int n = k.cw;
      if (rv == 1) { // h2v1: 3/4 nearer + 1/4 farther, rounding pattern 1, 2
        o[0] = r0[0], o[1] = (uint8_t)((r0[0] * 3 + r0[1] + 2) >> 2);
        for (int i = 1; i < n - 1; ++i) o[2 * i] = (uint8_t)((r0[i] * 3 + r0[i - 1] + 1) >> 2), o[2 * i + 1] = (uint8_t)((r0[i] * 3 + r0[i + 1] + 2) >> 2);
        o[2 * n - 2] = (uint8_t)((r0[n - 1] * 3 + r0[n - 2] + 1) >> 2), o[2 * n - 1] = r0[n - 1];
      } else {
        const uint8_t *r1 = row((y & 1) ? sy + 1 : sy - 1); // the nearer neighbour row: above for even output rows, below for odd
        if (rh == 1) { // h1v2
          const int bias = (y & 1) ? 2 : 1;
          for (int i = 0; i < n; ++i) o[i] = (uint8_t)((r0[i] * 3 + r1[i] + bias) >> 2);
        } else { // h2v2: vertical 3:1 sums, then the horizontal filter on them, rounding pattern 8, 7
          auto cs = [&](int i) { return (int)r0[i] * 3 + (int)r1[i]; };
          int thiscol = cs(0), nextcol = cs(1), lastcol;
          o[0] = (uint8_t)((thiscol * 4 + 8) >> 4), o[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
          lastcol = thiscol, thiscol = nextcol;
          for (int i = 1; i < n - 1; ++i) {
            nextcol = cs(i + 1);
            o[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4), o[2 * i + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol, thiscol = nextcol;
          }
          o[2 * n - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4), o[2 * n - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
        }
      }
      std::memcpy(&out[(size_t)y * W], o, W);
    }
  }

  void run(std::vector<uint8_t> &bgr, int &w_out, int &h_out) {
    parse();
    for (int c = 0; c < ncomp; ++c) {
      Comp &k = comp[c];
      if (!qt_ok[k.tq]) fail("component uses an undefined quantisation table");
      k.pix.assign((size_t)k.bw * 8 * k.bh * 8, 0);
      for (int by = 0; by < k.bh; ++by)
        for (int bx = 0; bx < k.bw; ++bx)
          idct_islow(&k.coef[((size_t)by * k.bw + bx) * 64], qt[k.tq], &k.pix[((size_t)by * 8) * k.bw * 8 + bx * 8], k.bw * 8);
    }
    w_out = W, h_out = H;
    bgr.assign((size_t)W * H * 3, 0);
    std::vector<uint8_t> p0, p1, p2;
    upsample(comp[0], p0);
    if (ncomp == 1) {
      for (size_t i = 0; i < (size_t)W * H; ++i) bgr[3 * i] = bgr[3 * i + 1] = bgr[3 * i + 2] = p0[i];
      return;
    }
    upsample(comp[1], p1), upsample(comp[2], p2);
    const bool ycc = !(adobe && adobe_transform == 0) && !(!adobe && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B');
    if (!ycc) {
      for (size_t i = 0; i < (size_t)W * H; ++i) bgr[3 * i] = p2[i], bgr[3 * i + 1] = p1[i], bgr[3 * i + 2] = p0[i];
      return;
    }
    // libjpeg jdcolor.c: 16 fractional bits, the Cb => G table carries the rounding half
    int cr_r[256], cb_b[256], cr_g[256], cb_g[256];
    for (int i = 0; i < 256; ++i) {
      const int x = i - 128;
      cr_r[i] = (91881 * x + 32768) >> 16;   // FIX(1.40200)
      cb_b[i] = (116130 * x + 32768) >> 16;  // FIX(1.77200)
      cr_g[i] = -46802 * x;                  // FIX(0.71414)
      cb_g[i] = -22554 * x + 32768;          // FIX(0.34414)
    }
    auto clamp = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
    for (size_t i = 0; i < (size_t)W * H; ++i) {
      const int y = p0[i], cb = p1[i], cr = p2[i];
      bgr[3 * i + 2] = clamp(y + cr_r[cr]);
      bgr[3 * i + 1] = clamp(y + ((cb_g[cb] + cr_g[cr]) >> 16));
      bgr[3 * i] = clamp(y + cb_b[cb]);
    }
  }
};

} // namespace

void decode_jpeg(const std::vector<uint8_t> &d, const std::string &path, std::vector<uint8_t> &bgr, int &W, int &H) {
  Decoder dec(d, path);
  dec.run(bgr, W, H);
}

} // namespace detail
} // namespace SoftRasterizer
