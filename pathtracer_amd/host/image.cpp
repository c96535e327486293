// Image: file decode / encode and pixel access (src/image.cpp:15-58, 87-95).
#include "image.h"

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include "codec.h"

Image::Image() : m_data(nullptr), m_width(0), m_height(0), m_channels(0) {}

Image::Image(int width, int height, int channels) : m_data(nullptr), m_width(width), m_height(height), m_channels(channels)
{
  m_data = (unsigned char*)std::malloc((size_t)width * (size_t)height * (size_t)channels);
}

Image::~Image() { free_data(); }

bool Image::load(const std::filesystem::path& path)
{
  free_data();
  std::ifstream in(path, std::ios::binary);
  if (!in) return false;
  std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  std::vector<uint8_t> rgb;
  std::string why;
  int w = 0, h = 0;
  bool ok = false;
  if (file.size() >= 2 && file[0] == 0xFF && file[1] == 0xD8) ok = pt::decode_jpeg(file, w, h, rgb, why);
  else ok = pt::decode_png(file, w, h, rgb, why);
  if (!ok) {
    std::cerr << "Image::load " << path << ": " << why << std::endl;
    return false;
  }
  m_width = w; m_height = h; m_channels = 3;
  m_data = (unsigned char*)std::malloc(rgb.size());
  // rows bottom-up, as the reference's loader leaves them (stbi_set_flip_vertically_on_load(true))
  for (int y = 0; y < h; y++) std::memcpy(m_data + (size_t)y * w * 3, rgb.data() + (size_t)(h - 1 - y) * w * 3, (size_t)w * 3);
  return true;
}

bool Image::write(const std::filesystem::path& path) const
{
  if (!valid() || m_channels != 3) return false;
  std::vector<uint8_t> png = pt::encode_png(m_data, m_width, m_height, m_width * 3);
  std::ofstream out(path, std::ios::binary);
  if (!out) return false;
  out.write((const char*)png.data(), (std::streamsize)png.size());
  return bool(out);
}

void Image::set_pixel(int x, int y, unsigned char* pixel)
{
  if (valid() && (0 <= x && x < m_width) && (0 <= y && y < m_height)) {
    size_t i = ((size_t)y * m_width + x) * 3;
    for (int c = 0; c < m_channels; c++) m_data[i + c] = pixel[c];
  }
}

void Image::free_data()
{
  if (m_data) {
    std::free(m_data);
    m_data = nullptr;
  }
}

bool Image::valid() const { return (m_data != nullptr) && (0 < m_width) && (0 < m_height) && (0 < m_channels); }
