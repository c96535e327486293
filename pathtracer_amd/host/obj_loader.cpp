#include "obj_loader.h"

#include <array>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>

namespace pt {

namespace {

struct Index {
  int v = 0, vt = 0, vn = 0;  // 1-based after resolving; 0 = absent
};

// "7", "7/3", "7//2", "7/3/2"; negative values count back from the end
bool parse_index(const std::string& tok, int nv, int nvt, int nvn, Index& out)
{
  int vals[3] = {0, 0, 0};
  bool have[3] = {false, false, false};
  size_t start = 0;
  for (int k = 0; k < 3 && start <= tok.size(); k++) {
    size_t slash = tok.find('/', start);
    std::string part = tok.substr(start, slash == std::string::npos ? std::string::npos : slash - start);
    if (!part.empty()) { vals[k] = std::atoi(part.c_str()); have[k] = true; }
    if (slash == std::string::npos) break;
    start = slash + 1;
  }
  if (!have[0]) return false;
  auto fix = [](int i, int n) { return i > 0 ? i : (i < 0 ? n + i + 1 : 0); };
  out.v = fix(vals[0], nv);
  out.vt = have[1] ? fix(vals[1], nvt) : 0;
  out.vn = have[2] ? fix(vals[2], nvn) : 0;
  return out.v >= 1 && out.v <= nv && out.vt <= nvt && out.vn <= nvn;
}

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

}  // namespace

bool load_obj_file(const std::filesystem::path& filename, const std::filesystem::path& mtl_search_path, ObjMesh& out,
                   std::string& error)
{
  std::ifstream in(filename);
  if (!in) { error = "Cannot open file [" + filename.string() + "]\n"; return false; }
  std::vector<float> v, vt, vn;
  std::map<std::string, int> mat_by_name;
  int current_material = -1;
  std::string line;
  size_t line_no = 0;
  while (std::getline(in, line)) {
    line_no++;
    std::istringstream ss(line);
    std::string key;
    if (!(ss >> key) || key[0] == '#') continue;
    if (key == "v") {
      float x = 0, y = 0, z = 0;
      ss >> x >> y >> z;
      v.insert(v.end(), {x, y, z});
    } else if (key == "vt") {
      float a = 0, b = 0;
      ss >> a >> b;
      vt.insert(vt.end(), {a, b});
    } else if (key == "vn") {
      float x = 0, y = 0, z = 0;
      ss >> x >> y >> z;
      vn.insert(vn.end(), {x, y, z});
    } else if (key == "mtllib") {
      std::string name;
      while (ss >> name) load_mtl(mtl_search_path / name, out.materials, mat_by_name, out.warning);
    } else if (key == "usemtl") {
      std::string name;
      ss >> name;
      auto it = mat_by_name.find(name);
      if (it == mat_by_name.end()) { out.warning += "material [ '" + name + "' ] not found in .mtl\n"; current_material = -1; }
      else current_material = it->second;
    } else if (key == "f") {
      std::vector<Index> face;
      std::string tok;
      while (ss >> tok) {
        Index idx;
        if (!parse_index(tok, (int)v.size() / 3, (int)vt.size() / 2, (int)vn.size() / 3, idx)) {
          error = "Failed to parse `f' line (e.g. a zero or out-of-range vertex index) at line " + std::to_string(line_no) + "\n";
          return false;
        }
        face.push_back(idx);
      }
      if (face.size() < 3) { out.warning += "Degenerated face found\n."; continue; }
      std::vector<std::array<int, 3>> tris;
      if (face.size() == 3) tris.push_back({0, 1, 2});
      else if (face.size() == 4) {
        // split along the shorter diagonal; on a tie take [0,1,3],[1,2,3] (float arithmetic, as tinyobj)
        const float* p0 = &v[3 * (face[0].v - 1)];
        const float* p1 = &v[3 * (face[1].v - 1)];
        const float* p2 = &v[3 * (face[2].v - 1)];
        const float* p3 = &v[3 * (face[3].v - 1)];
        float e02x = p2[0] - p0[0], e02y = p2[1] - p0[1], e02z = p2[2] - p0[2];
        float e13x = p3[0] - p1[0], e13y = p3[1] - p1[1], e13z = p3[2] - p1[2];
        float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
        float sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
        if (sqr02 < sqr13) { tris.push_back({0, 1, 2}); tris.push_back({0, 2, 3}); }
        else { tris.push_back({0, 1, 3}); tris.push_back({1, 2, 3}); }
      } else {
        for (size_t k = 1; k + 1 < face.size(); k++) tris.push_back({0, (int)k, (int)k + 1});
      }
      for (const auto& t : tris) {
        for (int k = 0; k < 3; k++) {
          const Index& idx = face[t[k]];
          ObjVertex ov;
          for (int a = 0; a < 3; a++) ov.pos[a] = v[3 * (idx.v - 1) + a];
          if (idx.vt > 0) { ov.uv[0] = vt[2 * (idx.vt - 1)]; ov.uv[1] = vt[2 * (idx.vt - 1) + 1]; ov.has_uv = true; }
          if (idx.vn > 0) { for (int a = 0; a < 3; a++) ov.normal[a] = vn[3 * (idx.vn - 1) + a]; ov.has_normal = true; }
          out.vertices.push_back(ov);
        }
        out.material_ids.push_back(current_material);
      }
    }
  }
  return true;
}

}  // namespace pt
