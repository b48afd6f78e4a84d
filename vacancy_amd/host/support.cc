// Host utilities of the facade: logger, PLY writer, 8-bit PNG codec on zlib.
#include <zlib.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "vacancy/image.h"
#include "vacancy/log.h"
#include "vacancy/mesh.h"

namespace vacancy {

// ---- logger ---------------------------------------------------------------------------------
namespace {
LogLevel g_level = LogLevel::kVerbose;
void Emit(LogLevel lvl, const char* tag, const char* format, va_list ap) {
  if (static_cast<int>(lvl) < static_cast<int>(g_level)) return;
  std::fputs(tag, stdout);
  std::vprintf(format, ap);
}
}  // namespace
void set_log_level(LogLevel level) { g_level = level; }
LogLevel get_log_level() { return g_level; }
#define VACANCY_LOG_FN(NAME, LVL, TAG)            \
  void NAME(const char* format, ...) {            \
    va_list ap;                                   \
    va_start(ap, format);                         \
    Emit(LogLevel::LVL, TAG, format, ap);         \
    va_end(ap);                                   \
  }
VACANCY_LOG_FN(LOGD, kDebug, "[D] ")
VACANCY_LOG_FN(LOGI, kInfo, "[I] ")
VACANCY_LOG_FN(LOGW, kWarning, "[W] ")
VACANCY_LOG_FN(LOGE, kError, "[E] ")

// ---- PLY --------------------------------------------------------------------------------------
// ASCII layout of the reference writer (mesh.cc:583-631): "x y z \n" with ostream default
// formatting (== %g) and "3 a b c \n".
bool Mesh::WritePly(const std::string& path) const {
  std::FILE* f = std::fopen(path.c_str(), "w");
  if (!f) {
    LOGE("couldn't open ply: %s\n", path.c_str());
    return false;
  }
  const bool color = !vertex_colors_.empty() && vertex_colors_.size() == vertices_.size();
  std::fprintf(f, "ply\nformat ascii 1.0\nelement vertex %zu\nproperty float x\nproperty float y\nproperty float z\n",
               vertices_.size());
  if (color) std::fprintf(f, "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty uchar alpha\n");
  std::fprintf(f, "element face %zu\nproperty list uchar int vertex_indices\nend_header\n", vertex_indices_.size());
  for (size_t i = 0; i < vertices_.size(); ++i) {
    std::fprintf(f, "%g %g %g ", vertices_[i][0], vertices_[i][1], vertices_[i][2]);
    if (color)
      std::fprintf(f, "%d %d %d 255 ", (int)std::lround(vertex_colors_[i][0]), (int)std::lround(vertex_colors_[i][1]),
                   (int)std::lround(vertex_colors_[i][2]));
    std::fputc('\n', f);
  }
  for (size_t i = 0; i < vertex_indices_.size(); ++i)
    std::fprintf(f, "3 %d %d %d \n", vertex_indices_[i][0], vertex_indices_[i][1], vertex_indices_[i][2]);
  std::fclose(f);
  return true;
}

bool Mesh::WritePlyBinary(const std::string& path) const {
  std::FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) {
    LOGE("couldn't open ply: %s\n", path.c_str());
    return false;
  }
  std::fprintf(f,
               "ply\nformat binary_little_endian 1.0\nelement vertex %zu\nproperty float x\nproperty float y\n"
               "property float z\nelement face %zu\nproperty list uchar int vertex_indices\nend_header\n",
               vertices_.size(), vertex_indices_.size());
  if (!vertices_.empty()) std::fwrite(vertices_.data(), sizeof(float) * 3, vertices_.size(), f);
  std::vector<unsigned char> rec(13 * 4096);
  for (size_t i = 0; i < vertex_indices_.size();) {
    size_t n = std::min<size_t>(4096, vertex_indices_.size() - i);
    for (size_t k = 0; k < n; ++k) {
      rec[13 * k] = 3;
      std::memcpy(&rec[13 * k + 1], &vertex_indices_[i + k], 12);
    }
    std::fwrite(rec.data(), 13, n, f);
    i += n;
  }
  std::fclose(f);
  return true;
}

// ---- PNG (8 bits per channel, non-interlaced; gray / gray+alpha / RGB / RGBA) -------------------
namespace {
uint32_t Be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
int Paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace

bool LoadPng8(const std::string& path, int* width, int* height, int* channels, std::vector<uint8_t>* pixels) {
  std::ifstream in(path, std::ios::binary);
  if (!in) {
    LOGE("couldn't open png: %s\n", path.c_str());
    return false;
  }
  std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  static const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 || std::memcmp(file.data(), kSig, 8) != 0) {
    LOGE("not a png: %s\n", path.c_str());
    return false;
  }
  int w = 0, h = 0, ch = 0;
  std::vector<uint8_t> idat;
  for (size_t pos = 8; pos + 12 <= file.size();) {
    const uint32_t len = Be32(&file[pos]);
    const char* type = reinterpret_cast<const char*>(&file[pos + 4]);
    const uint8_t* body = &file[pos + 8];
    if (pos + 12 + len > file.size()) break;
    if (!std::memcmp(type, "IHDR", 4)) {
      w = (int)Be32(body);
      h = (int)Be32(body + 4);
      const int depth = body[8], ctype = body[9], interlace = body[12];
      ch = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
      if (depth != 8 || ch == 0 || interlace != 0) {
        LOGE("unsupported png (depth %d, colour type %d, interlace %d): %s\n", depth, ctype, interlace, path.c_str());
        return false;
      }
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + len;
  }
  if (w <= 0 || h <= 0 || idat.empty()) return false;
  const size_t stride = (size_t)w * ch;
  std::vector<uint8_t> raw((stride + 1) * h);
  uLongf raw_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) {
    LOGE("corrupt png data: %s\n", path.c_str());
    return false;
  }
  pixels->assign(stride * h, 0);
  for (int y = 0; y < h; ++y) {
    const uint8_t filter = raw[(stride + 1) * y];
    const uint8_t* src = &raw[(stride + 1) * y + 1];
    uint8_t* dst = &(*pixels)[stride * y];
    const uint8_t* up = y ? dst - stride : nullptr;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= (size_t)ch ? dst[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
      int v = src[i];
      switch (filter) {
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) / 2; break;
        case 4: v += Paeth(a, b, c); break;
        default: break;
      }
      dst[i] = (uint8_t)v;
    }
  }
  *width = w;
  *height = h;
  *channels = ch;
  return true;
}

bool WritePng8(const std::string& path, int width, int height, int channels, const uint8_t* px) {
  const int ctype = channels == 1 ? 0 : channels == 2 ? 4 : channels == 3 ? 2 : channels == 4 ? 6 : -1;
  if (ctype < 0) return false;
  const size_t stride = (size_t)width * channels;
  std::vector<uint8_t> raw((stride + 1) * height);
  for (int y = 0; y < height; ++y) {
    raw[(stride + 1) * y] = 0;
    std::memcpy(&raw[(stride + 1) * y + 1], px + stride * y, stride);
  }
  uLongf zlen = compressBound((uLong)raw.size());
  std::vector<uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
  std::FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  auto chunk = [&](const char* type, const uint8_t* data, uint32_t len) {
    uint8_t hdr[8] = {uint8_t(len >> 24), uint8_t(len >> 16), uint8_t(len >> 8), uint8_t(len),
                      (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
    std::fwrite(hdr, 1, 8, f);
    if (len) std::fwrite(data, 1, len, f);
    uLong crc = crc32(0L, hdr + 4, 4);
    if (len) crc = crc32(crc, data, len);
    uint8_t c[4] = {uint8_t(crc >> 24), uint8_t(crc >> 16), uint8_t(crc >> 8), uint8_t(crc)};
    std::fwrite(c, 1, 4, f);
  };
  static const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::fwrite(kSig, 1, 8, f);
  uint8_t ihdr[13] = {uint8_t(width >> 24), uint8_t(width >> 16), uint8_t(width >> 8), uint8_t(width),
                      uint8_t(height >> 24), uint8_t(height >> 16), uint8_t(height >> 8), uint8_t(height),
                      8, (uint8_t)ctype, 0, 0, 0};
  chunk("IHDR", ihdr, 13);
  chunk("IDAT", z.data(), (uint32_t)zlen);
  chunk("IEND", nullptr, 0);
  std::fclose(f);
  return true;
}

}  // namespace vacancy
