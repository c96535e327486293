// TEST HARNESS ONLY.  Decodes every file named on the command line with the host image codecs and prints
// one line per file; compiled with -fsanitize=address,undefined by tests/test_host.py so that a read past a
// segment, a table index out of range or an undefined shift in the decoders ends the run.
#include <cstdio>
#include <fstream>
#include <iterator>

#include "../../pathtracer_amd/host/codec.h"

int main(int argc, char** argv)
{
  int decoded = 0, rejected = 0;
  for (int i = 1; i < argc; i++) {
    std::ifstream in(argv[i], std::ios::binary);
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    int w = 0, h = 0;
    std::vector<uint8_t> rgb;
    std::string why;
    const bool png = f.size() >= 4 && f[1] == 'P' && f[2] == 'N' && f[3] == 'G';
    const bool ok = png ? pt::decode_png(f, w, h, rgb, why) : pt::decode_jpeg(f, w, h, rgb, why);
    if (ok && rgb.size() != (size_t)w * (size_t)h * 3) { printf("%s: size mismatch\n", argv[i]); return 3; }
    ok ? decoded++ : rejected++;
  }
  printf("decoded %d rejected %d\n", decoded, rejected);
  return 0;
}
