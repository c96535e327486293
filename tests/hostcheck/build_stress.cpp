// TEST HARNESS ONLY.  Builds the host BVH over N random triangles (argv: N seed crowd) and checks the
// tree; compiled with -fsanitize=address,undefined by tests/test_host.py so that an out-of-range node slot
// in the threaded builder (thread count from PTK_BUILD_THREADS) is caught as a heap overflow.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../pathtracer_amd/csrc/ptk_build.h"

int main(int argc, char** argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 140000;
  const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1u;
  const bool crowd = argc > 3 && atoi(argv[3]) != 0;
  std::mt19937 gen(seed);
  std::uniform_real_distribution<double> pos(-10.0, 10.0), ext(-0.05, 0.05);
  // crowd: half of the triangles are crowded into one corner: lopsided splits high in the tree
  std::vector<double> p(9 * (size_t)n), nr(9 * (size_t)n, 0.0), uv(6 * (size_t)n, 0.0);
  std::vector<int32_t> mat((size_t)n, 0);
  for (int i = 0; i < n; i++) {
    double c[3] = {pos(gen), pos(gen), pos(gen)};
    if (crowd && (i & 1)) for (double& x : c) x = 9.0 + 0.1 * x;
    for (int v = 0; v < 3; v++)
      for (int k = 0; k < 3; k++) p[9 * (size_t)i + 3 * v + k] = c[k] + ext(gen);
    nr[9 * (size_t)i + 1] = nr[9 * (size_t)i + 4] = nr[9 * (size_t)i + 7] = 1.0;
  }
  ptk_material m = {};
  m.type = PTK_DIFFUSE; m.texture = -1; m.albedo[0] = m.albedo[1] = m.albedo[2] = 0.5; m.refraction_index = 1.52;
  ptk_scene_desc d = {};
  d.n_triangles = n; d.tri_pos = p.data(); d.tri_nrm = nr.data(); d.tri_uv = uv.data(); d.tri_material = mat.data();
  d.n_materials = 1; d.materials = &m;
  d.background_texture = -1;
  ptk::HostScene hs;
  std::string err = ptk::build_host_scene(d, hs);
  if (!err.empty()) { fprintf(stderr, "build failed: %s\n", err.c_str()); return 2; }
  if ((int)hs.prims.size() != n || hs.max_depth > ptk::kMaxTreeDepth || hs.nodes.empty()) { fprintf(stderr, "bad tree\n"); return 3; }
  printf("ok n=%d nodes=%zu depth=%d leaf=%d\n", n, hs.nodes.size(), hs.max_depth, hs.max_leaf);
  return 0;
}
