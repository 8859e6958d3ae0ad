// image_io.cpp — decode an image file to what cv::imread(path) (default flags) gives the reference's TextureLoader
// (src/TextureLoader.cpp:3-12): 8-bit, 3 channels, BGR order, alpha dropped, top row first.
// OpenCV is absent from this image; PNG (the textures of the raster path) is decoded here with zlib, JPEG (the height map of the
// bump / displacement shaders, examples/models/spot/hmap.jpg) in jpeg_decode.cpp, plus binary PPM (P6) for tools.  Unsupported
// input → std::runtime_error, like an empty cv::Mat does in the reference.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace SoftRasterizer {
namespace detail {

static std::vector<uint8_t> read_file(const std::string &path) {
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("Cannot open file: " + path);
  std::vector<uint8_t> buf;
  uint8_t tmp[65536];
  size_t n;
  while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  std::fclose(f);
  return buf;
}

static uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

static int paeth(int a, int b, int c) {
  int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

static void decode_png(const std::vector<uint8_t> &d, const std::string &path, std::vector<uint8_t> &bgr, int &W, int &H) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (d.size() < 33 || std::memcmp(d.data(), sig, 8) != 0) throw std::runtime_error("Cannot open file: " + path);
  size_t off = 8;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte;
  bool have_hdr = false;
  while (off + 12 <= d.size()) {
    uint32_t len = be32(&d[off]);
    const uint8_t *type = &d[off + 4];
    if (off + 12 + (size_t)len > d.size()) break;
    const uint8_t *data = &d[off + 8];
    if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
      W = (int)be32(data), H = (int)be32(data + 4), depth = data[8], ctype = data[9], interlace = data[12];
      have_hdr = true;
    } else if (!std::memcmp(type, "PLTE", 4)) {
      plte.assign(data, data + len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    off += 12 + (size_t)len;
  }
  if (!have_hdr || W <= 0 || H <= 0 || idat.empty()) throw std::runtime_error("Cannot open file: " + path);
  if (interlace) throw std::runtime_error("Cannot open file (interlaced PNG not supported): " + path);
  int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  // the layouts of the PNG specification: grey 1/2/4/8/16, palette 1/2/4/8 (never 16), truecolour / with alpha 8/16
  const bool depth_ok = ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                   : (depth == 8 || depth == 16 || (ctype == 0 && (depth == 1 || depth == 2 || depth == 4)));
  if (!channels || !depth_ok) throw std::runtime_error("Cannot open file (unsupported PNG layout): " + path);
  if (W > 32768 || H > 32768) // (what srz_texture_upload accepts; also bounds the allocations below for a malformed header)
    throw std::runtime_error("Cannot open file (image larger than 32768 x 32768): " + path);
  const size_t bpp_bits = (size_t)channels * depth, stride = ((size_t)W * bpp_bits + 7) / 8, bpp = (bpp_bits + 7) / 8;
  std::vector<uint8_t> raw((stride + 1) * (size_t)H);
  uLongf out_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size())
    throw std::runtime_error("Cannot open file (corrupt PNG stream): " + path);
  // unfilter in place
  std::vector<uint8_t> prev(stride, 0);
  std::vector<uint8_t> rows(stride * (size_t)H);
  for (int y = 0; y < H; ++y) {
    const uint8_t ft = raw[(stride + 1) * y];
    const uint8_t *src = &raw[(stride + 1) * y + 1];
    uint8_t *dst = &rows[stride * y];
    for (size_t i = 0; i < stride; ++i) {
      int a = i >= bpp ? dst[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, x = src[i];
      switch (ft) {
      case 0: break;
      case 1: x += a; break;
      case 2: x += b; break;
      case 3: x += (a + b) >> 1; break;
      case 4: x += paeth(a, b, c); break;
      default: throw std::runtime_error("Cannot open file (bad PNG filter): " + path);
      }
      dst[i] = (uint8_t)x;
    }
    std::memcpy(prev.data(), dst, stride);
  }
  bgr.assign((size_t)W * H * 3, 0);
  for (int y = 0; y < H; ++y) {
    const uint8_t *r = &rows[stride * y];
    for (int x = 0; x < W; ++x) {
      uint8_t R = 0, G = 0, B = 0;
      auto sample = [&](int ch) -> uint8_t { // 8/16-bit sample ch of pixel x (16-bit: high byte, like cv 16→8)
        size_t idx = ((size_t)x * channels + ch) * (depth / 8);
        return r[idx];
      };
      if (ctype == 2 || ctype == 6) {
        R = sample(0), G = sample(1), B = sample(2);
      } else if (ctype == 0 || ctype == 4) {
        uint8_t g;
        if (depth >= 8)
          g = sample(0);
        else {
          int per = 8 / depth, v = (r[x / per] >> ((per - 1 - x % per) * depth)) & ((1 << depth) - 1);
          g = (uint8_t)(v * 255 / ((1 << depth) - 1));
        }
        R = G = B = g;
      } else { // palette
        int idx;
        if (depth == 8)
          idx = r[x];
        else {
          int per = 8 / depth;
          idx = (r[x / per] >> ((per - 1 - x % per) * depth)) & ((1 << depth) - 1);
        }
        if ((size_t)idx * 3 + 2 < plte.size()) R = plte[idx * 3], G = plte[idx * 3 + 1], B = plte[idx * 3 + 2];
      }
      uint8_t *o = &bgr[((size_t)y * W + x) * 3];
      o[0] = B, o[1] = G, o[2] = R;
    }
  }
}

static void decode_ppm(const std::vector<uint8_t> &d, const std::string &path, std::vector<uint8_t> &bgr, int &W, int &H) {
  size_t p = 2;
  auto next_int = [&]() {
    while (p < d.size() && (d[p] == ' ' || d[p] == '\n' || d[p] == '\r' || d[p] == '\t' || d[p] == '#')) {
      if (d[p] == '#')
        while (p < d.size() && d[p] != '\n') ++p;
      else
        ++p;
    }
    long v = 0; // (a run of digits that no image size can be stops growing instead of overflowing)
    while (p < d.size() && d[p] >= '0' && d[p] <= '9') v = v < 100000000L ? v * 10 + (d[p] - '0') : v, ++p;
    return (int)v;
  };
  W = next_int(), H = next_int();
  int mx = next_int();
  ++p;
  if (W <= 0 || H <= 0 || mx != 255 || p + (size_t)W * H * 3 > d.size()) throw std::runtime_error("Cannot open file: " + path);
  bgr.resize((size_t)W * H * 3);
  for (size_t i = 0; i < (size_t)W * H; ++i) bgr[i * 3] = d[p + i * 3 + 2], bgr[i * 3 + 1] = d[p + i * 3 + 1], bgr[i * 3 + 2] = d[p + i * 3];
}

// BMP as cv::imread(path) (IMREAD_COLOR) reads it: uncompressed 8-bit palettised, 24-bit and 32-bit (alpha dropped) images with a
// BITMAPINFOHEADER or a later header, bottom-up or top-down rows padded to four bytes.  RLE, 1 / 4 / 16-bit and OS/2 headers are refused.
static void decode_bmp(const std::vector<uint8_t> &d, const std::string &path, std::vector<uint8_t> &bgr, int &W, int &H) {
  auto fail = [&]() -> void { throw std::runtime_error("Cannot open file: " + path); };
  auto le32 = [&](size_t o) { return (uint32_t)d[o] | (uint32_t)d[o + 1] << 8 | (uint32_t)d[o + 2] << 16 | (uint32_t)d[o + 3] << 24; };
  auto le16 = [&](size_t o) { return (uint32_t)d[o] | (uint32_t)d[o + 1] << 8; };
  if (d.size() < 54) fail();
  const uint32_t off = le32(10), hdr = le32(14);
  if (hdr < 40 || 14 + (size_t)hdr > d.size()) fail();
  const int32_t w = (int32_t)le32(18), h = (int32_t)le32(22);
  const uint32_t planes = le16(26), bpp = le16(28), comp = le32(30);
  const bool top_down = h < 0;
  const int64_t ah = top_down ? -(int64_t)h : (int64_t)h;
  if (w <= 0 || ah <= 0 || w > 32768 || ah > 32768 || planes != 1 || (bpp != 8 && bpp != 24 && bpp != 32)) fail();
  if (!(comp == 0 || (comp == 3 && bpp == 32))) fail(); // BI_RGB, or BI_BITFIELDS on 32 bits (the usual BGRA masks are assumed)
  const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
  if (off > d.size() || stride * (size_t)ah > d.size() - off) fail();
  uint32_t n_pal = 0;
  const size_t pal = 14 + (size_t)hdr;
  if (bpp == 8) {
    n_pal = le32(46) ? le32(46) : 256u;
    if (n_pal > 256 || pal + 4 * (size_t)n_pal > d.size()) fail();
  }
  W = w, H = (int)ah;
  bgr.resize((size_t)W * H * 3);
  for (int y = 0; y < H; ++y) {
    const uint8_t *row = &d[off + stride * (size_t)(top_down ? y : H - 1 - y)];
    uint8_t *o = &bgr[(size_t)y * W * 3];
    for (int x = 0; x < W; ++x, o += 3) {
      if (bpp == 8) {
        const uint32_t i = row[x];
        if (i >= n_pal) fail();
        o[0] = d[pal + 4 * i], o[1] = d[pal + 4 * i + 1], o[2] = d[pal + 4 * i + 2];
      } else {
        const uint8_t *px = row + (size_t)x * (bpp / 8);
        o[0] = px[0], o[1] = px[1], o[2] = px[2];
      }
    }
  }
}

void decode_jpeg(const std::vector<uint8_t> &d, const std::string &path, std::vector<uint8_t> &bgr, int &W, int &H); // jpeg_decode.cpp

void load_image_bgr(const std::string &path, std::vector<uint8_t> &bgr, int &W, int &H) {
  std::vector<uint8_t> d = read_file(path);
  if (d.size() >= 2 && d[0] == 'P' && d[1] == '6')
    decode_ppm(d, path, bgr, W, H);
  else if (d.size() >= 3 && d[0] == 0xff && d[1] == 0xd8 && d[2] == 0xff)
    decode_jpeg(d, path, bgr, W, H);
  else if (d.size() >= 2 && d[0] == 'B' && d[1] == 'M')
    decode_bmp(d, path, bgr, W, H);
  else
    decode_png(d, path, bgr, W, H);
}

} // namespace detail
} // namespace SoftRasterizer
