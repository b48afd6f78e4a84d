// Image<T, N>: row-major interleaved pixels in a std::vector, at(x, y, c) = data[(w*y + x)*N + c]
// (same layout as the reference's include/vacancy/image.h:22-74, which the carve path reads as
// a raw float buffer).  Load() decodes 8-bit PNG with zlib (the reference uses stb, absent here).
#pragma once

#include <cassert>
#include <cstdint>
#include <string>
#include <vector>

#include "vacancy/common.h"

namespace vacancy {

bool LoadPng8(const std::string& path, int* width, int* height, int* channels, std::vector<uint8_t>* pixels);
bool WritePng8(const std::string& path, int width, int height, int channels, const uint8_t* pixels);

template <typename T, int N>
class Image {
 public:
  Image() {}
  Image(int width, int height) { Init(width, height); }
  Image(int width, int height, T val) { Init(width, height, val); }
  void Init(int width, int height, T val = 0) {
    width_ = width;
    height_ = height;
    data_.assign(static_cast<size_t>(width) * height * N, val);
  }
  void Clear() { data_.clear(); width_ = height_ = -1; }
  bool empty() const { return width_ < 0 || height_ < 0 || data_.empty(); }
  int width() const { return width_; }
  int height() const { return height_; }
  int channel() const { return N; }
  const std::vector<T>& data() const { return data_; }
  std::vector<T>* data_ptr() { return &data_; }
  T& at(int x, int y, int c) { return data_[(static_cast<size_t>(width_) * y + x) * N + c]; }
  const T& at(int x, int y, int c) const { return data_[(static_cast<size_t>(width_) * y + x) * N + c]; }
  T* at(int x, int y) { return &data_[(static_cast<size_t>(width_) * y + x) * N]; }
  const T* at(int x, int y) const { return &data_[(static_cast<size_t>(width_) * y + x) * N]; }

  bool Load(const std::string& path) {
    static_assert(sizeof(T) == 1, "PNG I/O is 8 bits per channel");
    int w = 0, h = 0, ch = 0;
    std::vector<uint8_t> px;
    if (!LoadPng8(path, &w, &h, &ch, &px)) return false;
    if (ch != N) {
      LOGE("desired channel %d, actual %d\n", N, ch);
      return false;
    }
    width_ = w;
    height_ = h;
    data_.assign(px.begin(), px.end());
    return true;
  }
  bool WritePng(const std::string& path) const {
    static_assert(sizeof(T) == 1, "PNG I/O is 8 bits per channel");
    if (empty()) return false;
    return WritePng8(path, width_, height_, N, reinterpret_cast<const uint8_t*>(data_.data()));
  }

 private:
  std::vector<T> data_;
  int width_{-1};
  int height_{-1};
};

using Image1b = Image<uint8_t, 1>;
using Image3b = Image<uint8_t, 3>;
using Image1w = Image<uint16_t, 1>;
using Image1f = Image<float, 1>;
using Image3f = Image<float, 3>;

}  // namespace vacancy
