#include "obj_loader.h"

#include <algorithm>
#include <charconv>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <thread>

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

// One vertex reference of a face as written: "7", "7/3", "7//2", "7/3/2" in [b, e).  Values keep their
// sign (negative = relative to the elements read so far); 0 = absent.
struct RawIndex {
  int v = 0, vt = 0, vn = 0;
};

bool parse_raw_index(const char* b, const char* e, RawIndex& out)
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
  if (q != e || !have[0] || vals[0] == 0) return false;
  out.v = vals[0];
  out.vt = have[1] ? vals[1] : 0;
  out.vn = have[2] ? vals[2] : 0;
  return true;
}

// A contiguous piece of the file (whole lines) and what one thread read out of it.
struct Chunk {
  const char* begin = nullptr;
  const char* end = nullptr;
  std::vector<float> v, vt, vn;
  struct Face {
    uint32_t first;       // into `refs`
    uint32_t count;       // vertex references
    uint32_t nv, nvt, nvn;  // elements of this chunk read before the face (for relative indices)
    uint32_t line;        // line within the chunk, from 1
    int32_t event;        // index of the last usemtl of this chunk before the face, -1 = none yet
  };
  std::vector<Face> faces;
  std::vector<RawIndex> refs;
  struct Event {  // usemtl / mtllib lines, resolved in file order after all chunks are read
    bool is_lib;
    std::vector<std::string> names;
    int material = -1;  // usemtl: resolved id
  };
  std::vector<Event> events;
  size_t lines = 0;
  size_t error_line = 0;  // first unparsable face line (within the chunk), 0 = none
  size_t triangles = 0;
  // filled between the passes
  size_t v_before = 0, vt_before = 0, vn_before = 0, lines_before = 0, tri_before = 0;
  int material_at_start = -1;
  std::string warning;
  size_t range_error_line = 0;
};

void read_chunk(Chunk& c)
{
  const size_t bytes = (size_t)(c.end - c.begin);
  c.v.reserve(bytes / 40);
  c.refs.reserve(bytes / 24);
  c.faces.reserve(bytes / 72);
  const char* p = c.begin;
  int32_t last_event = -1;
  while (p < c.end) {
    const char* nl = (const char*)std::memchr(p, '\n', (size_t)(c.end - p));
    Cursor cur{p, nl ? nl : c.end};
    p = nl ? nl + 1 : c.end;
    c.lines++;
    const char *kb, *ke;
    if (!cur.token(kb, ke) || *kb == '#') continue;
    const size_t klen = (size_t)(ke - kb);
    auto key_is = [&](const char* k) { return std::strlen(k) == klen && std::memcmp(kb, k, klen) == 0; };
    if (key_is("v")) {
      float x = cur.real(), y = cur.real(), z = cur.real();
      c.v.insert(c.v.end(), {x, y, z});
    } else if (key_is("vt")) {
      float a = cur.real(), b = cur.real();
      c.vt.insert(c.vt.end(), {a, b});
    } else if (key_is("vn")) {
      float x = cur.real(), y = cur.real(), z = cur.real();
      c.vn.insert(c.vn.end(), {x, y, z});
    } else if (key_is("mtllib") || key_is("usemtl")) {
      Chunk::Event ev;
      ev.is_lib = key_is("mtllib");
      const char *b, *e;
      while (cur.token(b, e)) { ev.names.emplace_back(b, e); if (!ev.is_lib) break; }
      if (!ev.is_lib && ev.names.empty()) ev.names.emplace_back();
      c.events.push_back(std::move(ev));
      if (!c.events.back().is_lib) last_event = (int32_t)c.events.size() - 1;
    } else if (key_is("f")) {
      Chunk::Face f;
      f.first = (uint32_t)c.refs.size();
      f.nv = (uint32_t)(c.v.size() / 3); f.nvt = (uint32_t)(c.vt.size() / 2); f.nvn = (uint32_t)(c.vn.size() / 3);
      f.line = (uint32_t)c.lines;
      f.event = last_event;
      const char *b, *e;
      bool ok = true;
      while (cur.token(b, e)) {
        RawIndex r;
        if (!parse_raw_index(b, e, r)) { ok = false; break; }
        c.refs.push_back(r);
      }
      if (!ok) { if (!c.error_line) c.error_line = c.lines; c.refs.resize(f.first); continue; }
      f.count = (uint32_t)c.refs.size() - f.first;
      c.triangles += f.count >= 3 ? f.count - 2 : 0;
      c.faces.push_back(f);
    }
  }
}

// Second pass over a chunk: with all positions known, resolve the references and write the triangles.
void emit_chunk(Chunk& c, const std::vector<float>& v, const std::vector<float>& vt, const std::vector<float>& vn, ObjMesh& out)
{
  size_t tri = c.tri_before;
  for (const Chunk::Face& f : c.faces) {
    if (c.range_error_line) return;
    if (f.count < 3) { c.warning += "Degenerated face found\n."; continue; }
    const int nv = (int)(c.v_before + f.nv), nvt = (int)(c.vt_before + f.nvt), nvn = (int)(c.vn_before + f.nvn);
    auto fix = [](int i, int n) { return i > 0 ? i : (i < 0 ? n + i + 1 : 0); };
    const RawIndex* raw = &c.refs[f.first];
    // resolved in place into a small buffer (faces are short; large polygons take the heap once)
    Index small[8];
    std::vector<Index> big;
    Index* idx = small;
    if (f.count > 8) { big.resize(f.count); idx = big.data(); }
    bool ok = true;
    for (uint32_t k = 0; k < f.count; k++) {
      idx[k].v = fix(raw[k].v, nv);
      idx[k].vt = fix(raw[k].vt, nvt);
      idx[k].vn = fix(raw[k].vn, nvn);
      ok = ok && idx[k].v >= 1 && idx[k].v <= nv && idx[k].vt >= 0 && idx[k].vt <= nvt && idx[k].vn >= 0 && idx[k].vn <= nvn;
    }
    if (!ok) { c.range_error_line = f.line; return; }
    const int material = f.event >= 0 ? c.events[(size_t)f.event].material : c.material_at_start;
    auto emit = [&](int a, int b2, int c2) {
      const int t[3] = {a, b2, c2};
      for (int k = 0; k < 3; k++) {
        const Index& ix = idx[t[k]];
        ObjVertex ov;
        for (int ax = 0; ax < 3; ax++) ov.pos[ax] = v[3 * (size_t)(ix.v - 1) + ax];
        if (ix.vt > 0) { ov.uv[0] = vt[2 * (size_t)(ix.vt - 1)]; ov.uv[1] = vt[2 * (size_t)(ix.vt - 1) + 1]; ov.has_uv = true; }
        if (ix.vn > 0) { for (int ax = 0; ax < 3; ax++) ov.normal[ax] = vn[3 * (size_t)(ix.vn - 1) + ax]; ov.has_normal = true; }
        out.vertices[3 * tri + (size_t)k] = ov;
      }
      out.material_ids[tri] = material;
      tri++;
    };
    if (f.count == 3) {
      emit(0, 1, 2);
    } else if (f.count == 4) {
      // split along the shorter diagonal; on a tie take [0,1,3],[1,2,3] (float arithmetic, as tinyobj)
      const float* p0 = &v[3 * (size_t)(idx[0].v - 1)];
      const float* p1 = &v[3 * (size_t)(idx[1].v - 1)];
      const float* p2 = &v[3 * (size_t)(idx[2].v - 1)];
      const float* p3 = &v[3 * (size_t)(idx[3].v - 1)];
      float e02x = p2[0] - p0[0], e02y = p2[1] - p0[1], e02z = p2[2] - p0[2];
      float e13x = p3[0] - p1[0], e13y = p3[1] - p1[1], e13z = p3[2] - p1[2];
      float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
      float sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
      if (sqr02 < sqr13) { emit(0, 1, 2); emit(0, 2, 3); }
      else { emit(0, 1, 3); emit(1, 2, 3); }
    } else {
      for (uint32_t k = 1; k + 1 < f.count; k++) emit(0, (int)k, (int)k + 1);
    }
  }
}

template <class F>
void run_on_threads(size_t n, F&& f)
{
  std::vector<std::thread> workers;
  for (size_t i = 1; i < n; i++) workers.emplace_back([&f, i] { f(i); });
  f(0);
  for (auto& w : workers) w.join();
}

}  // namespace

// Large files are cut into one piece per thread at line ends.  Pass 1 (parallel) parses every piece on its
// own; then the pieces' element counts give each piece its global offsets, and the mtllib / usemtl lines are
// replayed in file order (so a usemtl sees exactly the materials loaded before it); pass 2 (parallel) resolves
// the face references and writes the triangles to their final places.
bool load_obj_file(const std::filesystem::path& filename, const std::filesystem::path& mtl_search_path, ObjMesh& out,
                   std::string& error)
{
  const bool timing = std::getenv("PT_TIMING") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[pt obj] %s: %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  };
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
  lap("read");
  size_t n_chunks = 1;
  if (text.size() >= (size_t)4 << 20) {
    n_chunks = std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));  // 16 = one GPU's share of a node's cores
    if (const char* e = std::getenv("PT_LOAD_THREADS")) n_chunks = (size_t)std::max(1, std::atoi(e));
  }
  std::vector<Chunk> chunks(n_chunks);
  {
    const char* p = text.data();
    const char* file_end = p + text.size();
    for (size_t i = 0; i < n_chunks; i++) {
      chunks[i].begin = p;
      const char* want = text.data() + text.size() * (i + 1) / n_chunks;
      if (i + 1 == n_chunks || want >= file_end) p = file_end;
      else {
        if (want < p) want = p;
        const char* nl = (const char*)std::memchr(want, '\n', (size_t)(file_end - want));
        p = nl ? nl + 1 : file_end;
      }
      chunks[i].end = p;
    }
  }
  run_on_threads(n_chunks, [&](size_t i) { read_chunk(chunks[i]); });

  lap("pass 1");
  // offsets, the first syntax error, materials in file order
  size_t nv = 0, nvt = 0, nvn = 0, lines = 0, tris = 0;
  for (Chunk& c : chunks) {
    c.v_before = nv / 3; c.vt_before = nvt / 2; c.vn_before = nvn / 3; c.lines_before = lines; c.tri_before = tris;
    nv += c.v.size(); nvt += c.vt.size(); nvn += c.vn.size(); lines += c.lines; tris += c.triangles;
  }
  std::map<std::string, int> mat_by_name;
  int current_material = -1;
  for (Chunk& c : chunks) {
    c.material_at_start = current_material;
    for (Chunk::Event& ev : c.events) {
      if (ev.is_lib) {
        for (const std::string& name : ev.names) load_mtl(mtl_search_path / name, out.materials, mat_by_name, out.warning);
      } else {
        auto it = mat_by_name.find(ev.names[0]);
        if (it == mat_by_name.end()) { out.warning += "material [ '" + ev.names[0] + "' ] not found in .mtl\n"; current_material = -1; }
        else current_material = it->second;
        ev.material = current_material;
      }
    }
  }
  std::vector<float> v(nv), vt(nvt), vn(nvn);
  run_on_threads(n_chunks, [&](size_t i) {
    const Chunk& c = chunks[i];
    if (!c.v.empty()) std::memcpy(&v[3 * c.v_before], c.v.data(), c.v.size() * sizeof(float));
    if (!c.vt.empty()) std::memcpy(&vt[2 * c.vt_before], c.vt.data(), c.vt.size() * sizeof(float));
    if (!c.vn.empty()) std::memcpy(&vn[3 * c.vn_before], c.vn.data(), c.vn.size() * sizeof(float));
  });
  lap("gather");
  out.vertices.resize(3 * tris);
  out.material_ids.resize(tris);
  lap("resize");
  run_on_threads(n_chunks, [&](size_t i) { emit_chunk(chunks[i], v, vt, vn, out); });
  lap("pass 2");

  // the first bad `f` line in file order, whether its syntax or its indices were at fault
  for (const Chunk& c : chunks) {
    size_t bad = 0;
    if (c.error_line && c.range_error_line) bad = std::min(c.error_line, c.range_error_line);
    else bad = c.error_line ? c.error_line : c.range_error_line;
    if (bad) {
      error = "Failed to parse `f' line (e.g. a zero or out-of-range vertex index) at line " + std::to_string(c.lines_before + bad) + "\n";
      return false;
    }
    out.warning += c.warning;
  }
  return true;
}

}  // namespace pt
