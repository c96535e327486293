// Baseline (sequential, Huffman, 8-bit) JPEG decoder: enough for the LDR equirectangular environment
// maps the reference's configs point at (background_texture, src/main.cpp:147-155).  Progressive and
// arithmetic-coded files are rejected with a reason.  Chroma is upsampled by replication and the
// inverse DCT is evaluated in floating point, so texels can differ by a level or two from other
// decoders (every JPEG decoder is allowed that much).
#include <cmath>
#include <algorithm>
#include <cstring>

#include "codec.h"

namespace pt {

namespace {

struct Huff {
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  int mincode[17], maxcode[18], valptr[17];
  bool present = false;
  void build()
  {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
      valptr[l] = k;
      mincode[l] = code;
      code += bits[l];
      k += bits[l];
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    present = true;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int dc_pred = 0;
  int blocks_w = 0, blocks_h = 0;  // in 8x8 blocks, padded to whole MCUs
  std::vector<uint8_t> plane;      // blocks_w*8 x blocks_h*8
};

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint32_t acc = 0;
  int n = 0;
  bool hit_marker = false;
  void fill()
  {
    while (n <= 24) {
      int b = 0;
      if (!hit_marker && p < end) {
        b = *p++;
        if (b == 0xFF) {
          int b2 = p < end ? *p : 0;
          if (b2 == 0) p++;  // stuffed zero
          else { hit_marker = true; p--; b = 0; }
        }
      }
      acc |= (uint32_t)b << (24 - n);
      n += 8;
    }
  }
  int get(int count)
  {
    if (count == 0) return 0;
    if (n < count) fill();
    int v = (int)(acc >> (32 - count));
    acc <<= count;
    n -= count;
    return v;
  }
  void reset() { acc = 0; n = 0; hit_marker = false; }
};

int decode_symbol(BitReader& br, const Huff& h)
{
  int code = 0;
  for (int l = 1; l <= 16; l++) {
    code = (code << 1) | br.get(1);
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
  }
  return -1;
}

int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

void idct8x8(const float* in, uint8_t* out, int stride)
{
  static float c[8][8];
  static bool init = false;
  if (!init) {
    for (int x = 0; x < 8; x++)
      for (int u = 0; u < 8; u++) c[x][u] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979f / 16.0f);
    init = true;
  }
  float tmp[64];
  for (int y = 0; y < 8; y++)
    for (int x = 0; x < 8; x++) {
      float s = 0;
      for (int u = 0; u < 8; u++) s += c[x][u] * in[y * 8 + u];
      tmp[y * 8 + x] = s;
    }
  for (int x = 0; x < 8; x++)
    for (int y = 0; y < 8; y++) {
      float s = 0;
      for (int v = 0; v < 8; v++) s += c[y][v] * tmp[v * 8 + x];
      int val = (int)std::lround(s + 128.0f);
      out[y * stride + x] = (uint8_t)(val < 0 ? 0 : (val > 255 ? 255 : val));
    }
}

}  // namespace

bool decode_jpeg(const std::vector<uint8_t>& f, int& width, int& height, std::vector<uint8_t>& rgb, std::string& why)
{
  if (f.size() < 4 || f[0] != 0xFF || f[1] != 0xD8) { why = "not a JPEG file"; return false; }
  uint16_t qt[4][64] = {{0}};
  Huff dc[4], ac[4];
  std::vector<Component> comps;
  int restart_interval = 0;
  int hmax = 1, vmax = 1;
  size_t pos = 2;
  bool have_frame = false;
  while (pos + 4 <= f.size()) {
    if (f[pos] != 0xFF) { pos++; continue; }
    int marker = f[pos + 1];
    pos += 2;
    if (marker == 0xD8 || (marker >= 0xD0 && marker <= 0xD7) || marker == 0x01 || marker == 0xFF) { if (marker == 0xFF) pos--; continue; }
    if (marker == 0xD9) break;
    size_t len = ((size_t)f[pos] << 8) | f[pos + 1];
    if (len < 2 || pos + len > f.size()) { why = "truncated JPEG segment"; return false; }
    const uint8_t* s = &f[pos + 2];
    size_t n = len - 2;
    if (marker == 0xDB) {  // DQT
      size_t i = 0;
      while (i < n) {
        int pq = s[i] >> 4, tq = s[i] & 15;
        i++;
        if (tq > 3 || pq > 1 || i + (pq ? 128u : 64u) > n) { why = "bad JPEG quantisation table"; return false; }
        for (int k = 0; k < 64; k++) {
          if (pq) { qt[tq][kZigzag[k]] = (uint16_t)((s[i] << 8) | s[i + 1]); i += 2; }
          else qt[tq][kZigzag[k]] = s[i++];
        }
      }
    } else if (marker == 0xC4) {  // DHT
      size_t i = 0;
      while (i + 17 <= n) {
        int tc = s[i] >> 4, th = s[i] & 15;
        if (th > 3) { why = "bad JPEG Huffman table"; return false; }
        Huff& h = tc ? ac[th] : dc[th];
        int total = 0;
        for (int l = 1; l <= 16; l++) { h.bits[l] = s[i + l]; total += h.bits[l]; }
        i += 17;
        if (total > 256 || i + total > n) { why = "bad JPEG Huffman table"; return false; }
        std::memcpy(h.vals, s + i, total);
        i += total;
        h.build();
      }
    } else if (marker == 0xC0 || marker == 0xC1) {  // SOF0 / SOF1
      if (n < 6) { why = "truncated JPEG frame header"; return false; }
      if (s[0] != 8) { why = "only 8-bit JPEG is supported"; return false; }
      height = (s[1] << 8) | s[2];
      width = (s[3] << 8) | s[4];
      int nc = s[5];
      if (width <= 0 || height <= 0 || (nc != 1 && nc != 3)) { why = "unsupported JPEG frame"; return false; }
      if (n < 6 + 3 * (size_t)nc) { why = "truncated JPEG frame header"; return false; }
      comps.resize(nc);
      for (int i = 0; i < nc; i++) {
        comps[i].id = s[6 + 3 * i];
        comps[i].h = s[7 + 3 * i] >> 4;
        comps[i].v = s[7 + 3 * i] & 15;
        comps[i].tq = s[8 + 3 * i];
        if (comps[i].h < 1 || comps[i].h > 4 || comps[i].v < 1 || comps[i].v > 4 || comps[i].tq > 3) { why = "bad JPEG component"; return false; }
        hmax = std::max(hmax, comps[i].h);
        vmax = std::max(vmax, comps[i].v);
      }
      have_frame = true;
    } else if (marker == 0xC2 || (marker >= 0xC5 && marker <= 0xCF && marker != 0xC8 && marker != 0xCC)) {
      why = "progressive / arithmetic / lossless JPEG is not supported";
      return false;
    } else if (marker == 0xDD) {
      if (n < 2) { why = "truncated JPEG restart interval"; return false; }
      restart_interval = (s[0] << 8) | s[1];
    } else if (marker == 0xDA) {  // SOS: entropy-coded data follows
      if (!have_frame) { why = "JPEG scan before frame header"; return false; }
      if (n < 1) { why = "truncated JPEG scan header"; return false; }
      int ns = s[0];
      if (ns != (int)comps.size()) { why = "non-interleaved JPEG scans are not supported"; return false; }
      if (n < 1 + 2 * (size_t)ns + 3) { why = "truncated JPEG scan header"; return false; }
      for (int i = 0; i < ns; i++) {
        int cid = s[1 + 2 * i];
        int td = s[2 + 2 * i] >> 4, ta = s[2 + 2 * i] & 15;
        if (td > 3 || ta > 3) { why = "bad JPEG Huffman table selector"; return false; }  // they index dc[4] / ac[4]
        for (auto& c : comps)
          if (c.id == cid) { c.td = td; c.ta = ta; }
      }
      int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
      int mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
      // A damaged frame header must not turn into a huge allocation: every coded 8x8 block takes at least
      // two bits (one DC and one AC code).
      if ((uint64_t)width * (uint64_t)height > (1ull << 28) || (uint64_t)mcus_x * (uint64_t)mcus_y / 4 > f.size() - (pos + len)) {
        why = "JPEG frame size does not fit its data";
        return false;
      }
      for (auto& c : comps) {
        c.blocks_w = mcus_x * c.h;
        c.blocks_h = mcus_y * c.v;
        c.plane.assign((size_t)c.blocks_w * 8 * (size_t)c.blocks_h * 8, 0);
        c.dc_pred = 0;
        if (!dc[c.td].present || !ac[c.ta].present) { why = "JPEG scan uses an undefined Huffman table"; return false; }
      }
      BitReader br{&f[pos + len], f.data() + f.size()};
      int until_restart = restart_interval;
      for (int my = 0; my < mcus_y; my++) {
        for (int mx = 0; mx < mcus_x; mx++) {
          if (restart_interval && until_restart == 0) {
            // byte-align, skip the RSTn marker, reset predictors
            br.reset();
            while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) br.p++;
            if (br.p + 1 < br.end) br.p += 2;
            for (auto& c : comps) c.dc_pred = 0;
            until_restart = restart_interval;
          }
          for (auto& c : comps) {
            for (int by = 0; by < c.v; by++)
              for (int bx = 0; bx < c.h; bx++) {
                float block[64] = {0};
                int t = decode_symbol(br, dc[c.td]);
                if (t < 0 || t > 11) { why = "corrupt JPEG data (DC)"; return false; }
                int diff = extend(br.get(t), t);
                c.dc_pred += diff;
                block[0] = (float)(c.dc_pred * qt[c.tq][0]);
                for (int k = 1; k < 64;) {
                  int rs = decode_symbol(br, ac[c.ta]);
                  if (rs < 0) { why = "corrupt JPEG data (AC)"; return false; }
                  int r = rs >> 4, sz = rs & 15;
                  if (sz == 0) {
                    if (r == 15) { k += 16; continue; }
                    break;
                  }
                  k += r;
                  if (k > 63) { why = "corrupt JPEG data (run)"; return false; }
                  int zz = kZigzag[k];
                  block[zz] = (float)(extend(br.get(sz), sz) * qt[c.tq][zz]);
                  k++;
                }
                size_t stride = (size_t)c.blocks_w * 8;
                size_t ox = ((size_t)mx * c.h + bx) * 8, oy = ((size_t)my * c.v + by) * 8;
                idct8x8(block, &c.plane[oy * stride + ox], (int)stride);
              }
          }
          if (restart_interval) until_restart--;
        }
      }
      // bring subsampled components to full resolution: triangle filter (weights 3/4, 1/4 towards the
      // neighbouring sample) for 2x factors, replication otherwise
      std::vector<std::vector<uint8_t>> full(comps.size());
      for (size_t ci = 0; ci < comps.size(); ci++) {
        const Component& c = comps[ci];
        int fx = hmax / c.h, fy = vmax / c.v;
        if ((fx == 1 && fy == 1) || !((fx == 1 || fx == 2) && (fy == 1 || fy == 2)) || hmax % c.h || vmax % c.v) continue;
        int cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax;  // valid samples
        size_t stride = (size_t)c.blocks_w * 8;
        full[ci].resize((size_t)width * (size_t)height);
        std::vector<int> row(cw);
        for (int y = 0; y < height; y++) {
          int sy = fy == 2 ? y >> 1 : y;
          int far_y = sy;
          if (fy == 2) far_y = (y & 1) ? std::min(sy + 1, ch - 1) : std::max(sy - 1, 0);
          const uint8_t* near_row = &c.plane[(size_t)sy * stride];
          const uint8_t* far_row = &c.plane[(size_t)far_y * stride];
          // vertical pass in units of 1/4 (fy == 2) or 1 (fy == 1)
          for (int i = 0; i < cw; i++) row[i] = fy == 2 ? 3 * near_row[i] + far_row[i] : near_row[i];
          uint8_t* out = &full[ci][(size_t)y * width];
          int vscale = fy == 2 ? 4 : 1;
          for (int x = 0; x < width; x++) {
            if (fx == 1) { out[x] = (uint8_t)((row[x] + vscale / 2) / vscale); continue; }
            int i = x >> 1;
            int j = (x & 1) ? std::min(i + 1, cw - 1) : std::max(i - 1, 0);
            int v = 3 * row[i] + row[j];
            out[x] = (uint8_t)((v + 2 * vscale) / (4 * vscale));
          }
        }
      }
      rgb.resize((size_t)width * (size_t)height * 3);
      for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
          auto at = [&](const Component& c) -> float {
            size_t ci = (size_t)(&c - &comps[0]);
            if (!full[ci].empty()) return (float)full[ci][(size_t)y * width + x];
            int cx = x * c.h / hmax, cy = y * c.v / vmax;
            return (float)c.plane[(size_t)cy * c.blocks_w * 8 + cx];
          };
          uint8_t* o = &rgb[((size_t)y * width + x) * 3];
          if (comps.size() == 1) { o[0] = o[1] = o[2] = (uint8_t)at(comps[0]); continue; }
          float Y = at(comps[0]), cb = at(comps[1]) - 128.0f, cr = at(comps[2]) - 128.0f;
          float r = Y + 1.402f * cr, g = Y - 0.344136f * cb - 0.714136f * cr, b = Y + 1.772f * cb;
          auto clamp8 = [](float v) { int i = (int)std::lround(v); return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i)); };
          o[0] = clamp8(r); o[1] = clamp8(g); o[2] = clamp8(b);
        }
      return true;
    }
    pos += len;
  }
  why = "JPEG has no image scan";
  return false;
}

}  // namespace pt
