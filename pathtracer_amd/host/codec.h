// Image file codecs used by Image (host-side I/O, no GPU involvement).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace pt {

// Decode to 8-bit RGB, rows top-down.  Returns false (and a reason) on unsupported or corrupt files.
bool decode_png(const std::vector<uint8_t>& file, int& width, int& height, std::vector<uint8_t>& rgb, std::string& why);
bool decode_jpeg(const std::vector<uint8_t>& file, int& width, int& height, std::vector<uint8_t>& rgb, std::string& why);
// Encode 8-bit RGB (rows top-down, `stride` bytes apart) as a PNG file image.
std::vector<uint8_t> encode_png(const uint8_t* rgb, int width, int height, int stride);

}  // namespace pt
