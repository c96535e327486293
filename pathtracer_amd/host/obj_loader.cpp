#include "obj_loader.h"

#include <algorithm>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace pt {

namespace {

struct Index {
  int v = 0, vt = 0, vn = 0;  // 1-based after resolving; 0 = absent
};

void load_mtl(const std::filesystem::path& path, std::vector<ObjMaterial>& mats, std::map<std::string, int>& by_name,
              std::string& warning)
{
  std::ifstream in(path);
  if (!in) { warning += "Material file [ " + path.string() + " ] not found.\n"; return; }
  std::string line;
  ObjMaterial* cur = nullptr;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string key;
    if (!(ss >> key) || key[0] == '#') continue;
    if (key == "newmtl") {
      std::string name;
      ss >> name;
      mats.emplace_back();
      cur = &mats.back();
      cur->name = name;
      by_name[name] = (int)mats.size() - 1;
    } else if (!cur) continue;
    else if (key == "Kd") ss >> cur->diffuse[0] >> cur->diffuse[1] >> cur->diffuse[2];
    else if (key == "Ke") ss >> cur->emission[0] >> cur->emission[1] >> cur->emission[2];
    else if (key == "illum") ss >> cur->illum;
    else if (key == "Pr") ss >> cur->roughness;
    else if (key == "Pm") ss >> cur->metallic;
    else if (key == "map_Kd") {
      std::string rest;
      std::getline(ss, rest);
      size_t a = rest.find_first_not_of(" \t"), b = rest.find_last_not_of(" \t\r");
      cur->diffuse_texname = a == std::string::npos ? std::string() : rest.substr(a, b - a + 1);
    }
  }
}

// A forward cursor over one line of the file (the whole file is read into memory first: formatted
// stream extraction costs several seconds on a 100 MB OBJ).
struct Cursor {
  const char* p;
  const char* end;  // end of the line
  void skip_space() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) p++; }
  bool at_end() { skip_space(); return p >= end; }
  // next blank-delimited token
  bool token(const char*& b, const char*& e)
  {
    skip_space();
    if (p >= end) return false;
    b = p;
    while (p < end && *p != ' ' && *p != '\t' && *p != '\r') p++;
    e = p;
    return true;
  }
  // a real number, read as double and narrowed to float like the reference's parser; 0 when absent
  float real()
  {
    const char *b, *e;
    if (!token(b, e)) return 0.0f;
    if (b < e && *b == '+') b++;  // from_chars takes no leading plus sign
    double d = 0.0;
    std::from_chars(b, e, d);  // leaves 0 when the token is not a number
    return (float)d;
  }
};

// "7", "7/3", "7//2", "7/3/2" in [b, e); negative values count back from the end
bool parse_index_range(const char* b, const char* e, int nv, int nvt, int nvn, Index& out)
{
  int vals[3] = {0, 0, 0};
  bool have[3] = {false, false, false};
  int k = 0;
  const char* q = b;
  while (q <= e && k < 3) {
    const char* s0 = q;
    bool neg = false;
    if (q < e && (*q == '-' || *q == '+')) { neg = *q == '-'; q++; }
    long v = 0;
    const char* digits = q;
    while (q < e && *q >= '0' && *q <= '9') { v = v * 10 + (*q - '0'); if (v > 0x7fffffffL) return false; q++; }
    if (q > digits) { vals[k] = (int)(neg ? -v : v); have[k] = true; }
    else if (q != s0) return false;  // a sign without digits
    if (q < e && *q == '/') { q++; k++; continue; }
    break;
  }
  if (q != e || !have[0]) return false;
  auto fix = [](int i, int n) { return i > 0 ? i : (i < 0 ? n + i + 1 : 0); };
  out.v = fix(vals[0], nv);
  out.vt = have[1] ? fix(vals[1], nvt) : 0;
  out.vn = have[2] ? fix(vals[2], nvn) : 0;
  return out.v >= 1 && out.v <= nv && out.vt >= 0 && out.vt <= nvt && out.vn >= 0 && out.vn <= nvn;
}

}  // namespace

bool load_obj_file(const std::filesystem::path& filename, const std::filesystem::path& mtl_search_path, ObjMesh& out,
                   std::string& error)
{
  std::string text;
  {
    FILE* f = std::fopen(filename.string().c_str(), "rb");
    if (!f) { error = "Cannot open file [" + filename.string() + "]\n"; return false; }
    std::fseek(f, 0, SEEK_END);
    long size = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    text.resize(size > 0 ? (size_t)size : 0);
    size_t got = text.empty() ? 0 : std::fread(&text[0], 1, text.size(), f);
    std::fclose(f);
    text.resize(got);
  }
  std::vector<float> v, vt, vn;
  v.reserve(text.size() / 40);
  out.vertices.reserve(text.size() / 12);
  out.material_ids.reserve(text.size() / 36);
  std::map<std::string, int> mat_by_name;
  int current_material = -1;
  size_t line_no = 0;
  std::vector<Index> face;
  const char* p = text.data();
  const char* file_end = p + text.size();
  while (p < file_end) {
    const char* nl = (const char*)std::memchr(p, '\n', (size_t)(file_end - p));
    Cursor c{p, nl ? nl : file_end};
    p = nl ? nl + 1 : file_end;
    line_no++;
    const char *kb, *ke;
    if (!c.token(kb, ke) || *kb == '#') continue;
    const size_t klen = (size_t)(ke - kb);
    auto key_is = [&](const char* k) { return std::strlen(k) == klen && std::memcmp(kb, k, klen) == 0; };
    if (key_is("v")) {
      float x = c.real(), y = c.real(), z = c.real();
      v.insert(v.end(), {x, y, z});
    } else if (key_is("vt")) {
      float a = c.real(), b = c.real();
      vt.insert(vt.end(), {a, b});
    } else if (key_is("vn")) {
      float x = c.real(), y = c.real(), z = c.real();
      vn.insert(vn.end(), {x, y, z});
    } else if (key_is("mtllib")) {
      const char *b, *e;
      while (c.token(b, e)) load_mtl(mtl_search_path / std::string(b, e), out.materials, mat_by_name, out.warning);
    } else if (key_is("usemtl")) {
      const char *b, *e;
      std::string name = c.token(b, e) ? std::string(b, e) : std::string();
      auto it = mat_by_name.find(name);
      if (it == mat_by_name.end()) { out.warning += "material [ '" + name + "' ] not found in .mtl\n"; current_material = -1; }
      else current_material = it->second;
    } else if (key_is("f")) {
      face.clear();
      const char *b, *e;
      while (c.token(b, e)) {
        Index idx;
        if (!parse_index_range(b, e, (int)v.size() / 3, (int)vt.size() / 2, (int)vn.size() / 3, idx)) {
          error = "Failed to parse `f' line (e.g. a zero or out-of-range vertex index) at line " + std::to_string(line_no) + "\n";
          return false;
        }
        face.push_back(idx);
      }
      if (face.size() < 3) { out.warning += "Degenerated face found\n."; continue; }
      int tris[2][3];
      int n_tris = 0;
      auto emit = [&](int a, int b2, int c2) {
        const int t[3] = {a, b2, c2};
        for (int k = 0; k < 3; k++) {
          const Index& idx = face[(size_t)t[k]];
          ObjVertex ov;
          for (int ax = 0; ax < 3; ax++) ov.pos[ax] = v[3 * (size_t)(idx.v - 1) + ax];
          if (idx.vt > 0) { ov.uv[0] = vt[2 * (size_t)(idx.vt - 1)]; ov.uv[1] = vt[2 * (size_t)(idx.vt - 1) + 1]; ov.has_uv = true; }
          if (idx.vn > 0) { for (int ax = 0; ax < 3; ax++) ov.normal[ax] = vn[3 * (size_t)(idx.vn - 1) + ax]; ov.has_normal = true; }
          out.vertices.push_back(ov);
        }
        out.material_ids.push_back(current_material);
      };
      if (face.size() == 3) {
        emit(0, 1, 2);
      } else if (face.size() == 4) {
        // split along the shorter diagonal; on a tie take [0,1,3],[1,2,3] (float arithmetic, as tinyobj)
        const float* p0 = &v[3 * (size_t)(face[0].v - 1)];
        const float* p1 = &v[3 * (size_t)(face[1].v - 1)];
        const float* p2 = &v[3 * (size_t)(face[2].v - 1)];
        const float* p3 = &v[3 * (size_t)(face[3].v - 1)];
        float e02x = p2[0] - p0[0], e02y = p2[1] - p0[1], e02z = p2[2] - p0[2];
        float e13x = p3[0] - p1[0], e13y = p3[1] - p1[1], e13z = p3[2] - p1[2];
        float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
        float sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
        if (sqr02 < sqr13) { tris[0][0] = 0; tris[0][1] = 1; tris[0][2] = 2; tris[1][0] = 0; tris[1][1] = 2; tris[1][2] = 3; }
        else { tris[0][0] = 0; tris[0][1] = 1; tris[0][2] = 3; tris[1][0] = 1; tris[1][1] = 2; tris[1][2] = 3; }
        n_tris = 2;
        for (int t = 0; t < n_tris; t++) emit(tris[t][0], tris[t][1], tris[t][2]);
      } else {
        for (size_t k = 1; k + 1 < face.size(); k++) emit(0, (int)k, (int)k + 1);
      }
    }
  }
  return true;
}

}  // namespace pt
