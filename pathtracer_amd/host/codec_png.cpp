// PNG decode / encode on top of zlib.
#include <zlib.h>

#include <cstdlib>
#include <cstring>

#include "codec.h"

namespace pt {

namespace {
uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]); }
void put32(std::vector<uint8_t>& v, uint32_t x)
{
  v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x));
}
int paeth(int a, int b, int c)
{
  int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}
}  // namespace

bool decode_png(const std::vector<uint8_t>& f, int& width, int& height, std::vector<uint8_t>& rgb, std::string& why)
{
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (f.size() < 8 || std::memcmp(f.data(), sig, 8) != 0) { why = "not a PNG file"; return false; }
  size_t pos = 8;
  int bit_depth = 0, color_type = 0, interlace = 0;
  std::vector<uint8_t> idat, palette;
  bool have_header = false;
  while (pos + 12 <= f.size()) {
    uint32_t len = be32(&f[pos]);
    const uint8_t* type = &f[pos + 4];
    const uint8_t* body = &f[pos + 8];
    if (pos + 12 + (size_t)len > f.size()) { why = "truncated PNG chunk"; return false; }
    if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
      width = (int)be32(body); height = (int)be32(body + 4);
      bit_depth = body[8]; color_type = body[9]; interlace = body[12];
      have_header = true;
    } else if (!std::memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
    else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
    else if (!std::memcmp(type, "IEND", 4)) break;
    pos += 12 + (size_t)len;
  }
  if (!have_header || width <= 0 || height <= 0) { why = "missing PNG header"; return false; }
  if (interlace != 0) { why = "interlaced PNG is not supported"; return false; }
  int channels;
  switch (color_type) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: why = "bad PNG colour type"; return false;
  }
  const bool small_depth = bit_depth == 1 || bit_depth == 2 || bit_depth == 4;  // the only depths below 8 that PNG defines
  if (!(bit_depth == 8 || (bit_depth == 16 && color_type != 3) || (small_depth && (color_type == 0 || color_type == 3)))) { why = "unsupported PNG bit depth"; return false; }
  size_t bits_per_pixel = (size_t)channels * (size_t)bit_depth;
  size_t row_bytes = ((size_t)width * bits_per_pixel + 7) / 8;
  size_t bpp = bits_per_pixel < 8 ? 1 : bits_per_pixel / 8;  // filter distance
  // A damaged header must not turn into a huge allocation: deflate expands by at most about 1032 times.
  if ((uint64_t)width * (uint64_t)height > (1ull << 28) || (row_bytes + 1) * (size_t)height > 1040 * idat.size() + 64) {
    why = "PNG image size does not fit its data";
    return false;
  }
  std::vector<uint8_t> raw((row_bytes + 1) * (size_t)height);
  uLongf out_len = (uLongf)raw.size();
  int zr = uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size());
  if (zr != Z_OK || out_len != raw.size()) { why = "PNG inflate failed"; return false; }
  // undo the per-row filters in place
  std::vector<uint8_t> prev(row_bytes, 0);
  for (int y = 0; y < height; y++) {
    uint8_t* row = &raw[(row_bytes + 1) * (size_t)y];
    int filter = row[0];
    uint8_t* cur = row + 1;
    for (size_t i = 0; i < row_bytes; i++) {
      int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int add;
      switch (filter) {
        case 0: add = 0; break;
        case 1: add = a; break;
        case 2: add = b; break;
        case 3: add = (a + b) >> 1; break;
        case 4: add = paeth(a, b, c); break;
        default: why = "bad PNG filter"; return false;
      }
      cur[i] = uint8_t(cur[i] + add);
    }
    std::memcpy(prev.data(), cur, row_bytes);
  }
  rgb.resize((size_t)width * (size_t)height * 3);
  for (int y = 0; y < height; y++) {
    const uint8_t* cur = &raw[(row_bytes + 1) * (size_t)y + 1];
    uint8_t* out = &rgb[(size_t)y * (size_t)width * 3];
    for (int x = 0; x < width; x++) {
      auto sample = [&](int ch) -> int {  // 8-bit value of channel ch of pixel x
        if (bit_depth == 8) return cur[(size_t)x * channels + ch];
        if (bit_depth == 16) return cur[((size_t)x * channels + ch) * 2];  // high byte
        size_t bit = (size_t)x * bit_depth;
        int v = (cur[bit >> 3] >> (8 - bit_depth - (bit & 7))) & ((1 << bit_depth) - 1);
        return color_type == 3 ? v : v * 255 / ((1 << bit_depth) - 1);
      };
      if (color_type == 3) {
        size_t idx = (size_t)sample(0) * 3;
        if (idx + 2 < palette.size()) { out[3 * x] = palette[idx]; out[3 * x + 1] = palette[idx + 1]; out[3 * x + 2] = palette[idx + 2]; }
        else out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = 0;
      } else if (channels <= 2) {
        out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = uint8_t(sample(0));
      } else {
        out[3 * x] = uint8_t(sample(0)); out[3 * x + 1] = uint8_t(sample(1)); out[3 * x + 2] = uint8_t(sample(2));
      }
    }
  }
  return true;
}

std::vector<uint8_t> encode_png(const uint8_t* rgb, int width, int height, int stride)
{
  std::vector<uint8_t> raw;
  raw.reserve(((size_t)width * 3 + 1) * (size_t)height);
  for (int y = 0; y < height; y++) {
    raw.push_back(0);  // filter: none
    raw.insert(raw.end(), rgb + (size_t)y * stride, rgb + (size_t)y * stride + (size_t)width * 3);
  }
  uLongf zlen = compressBound((uLong)raw.size());
  std::vector<uint8_t> z(zlen);
  compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6);
  z.resize(zlen);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  auto chunk = [&](const char* type, const std::vector<uint8_t>& body) {
    put32(out, (uint32_t)body.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), body.begin(), body.end());
    put32(out, (uint32_t)crc32(0, &out[start], (uInt)(out.size() - start)));
  };
  std::vector<uint8_t> hdr;
  put32(hdr, (uint32_t)width); put32(hdr, (uint32_t)height);
  hdr.push_back(8); hdr.push_back(2); hdr.push_back(0); hdr.push_back(0); hdr.push_back(0);
  chunk("IHDR", hdr);
  chunk("IDAT", z);
  chunk("IEND", {});
  return out;
}

}  // namespace pt
