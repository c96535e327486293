// 8-bit RGB image: the public surface of src/image.h:7-28.  Files are decoded to 3 channels with rows
// stored bottom-up (what Image::load produces there through stbi with the vertical flip, src/image.cpp:25-31).
// Texture sampling for rendering happens on the GPU; sample() here is not provided.
// Decoders: PNG (non-interlaced, 8/16 bit, grey / RGB / palette / alpha variants) and baseline JPEG.
#pragma once

#include <filesystem>
#include <string>

class Image
{
 public:
  Image();
  Image(int width, int height, int channels);
  ~Image();
  Image(const Image&) = delete;
  Image& operator=(const Image&) = delete;
  bool load(const std::filesystem::path& path);
  bool write(const std::filesystem::path& path) const;  // PNG, row 0 of the buffer = top row of the file
  int width() const { return m_width; }
  int height() const { return m_height; }
  int channels() const { return m_channels; }
  void set_pixel(int x, int y, unsigned char* pixel);
  bool valid() const;
  const unsigned char* data() const { return m_data; }

 private:
  unsigned char* m_data;
  int m_width, m_height, m_channels;
  void free_data();
};
