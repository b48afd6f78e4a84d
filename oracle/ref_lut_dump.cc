// Test infrastructure: dumps the marching-cubes tables of the REFERENCE ITSELF.
//
// The one translation unit of the reference's path that compiles in this image without Eigen is
// src/vacancy/marching_cubes_lut.cc (kEdgeTable :15-41, kTriTable :42-298).  oracle/ref.mk compiles
// it UNMODIFIED, where it lies under /root/reference, into oracle/_ref/ and links this dumper
// against it; the dumper only reads the two arrays through the reference's own header.
//   ref_lut_dump <out.bin>   writes int32 kEdgeTable[256] then int32 kTriTable[256][16], little endian
// The same object is also linked into oracle/_ref/libref_mc_lut.so with the C accessors below.
#include <cstdint>
#include <cstdio>

#include "vacancy/marching_cubes_lut.h"  // the reference's header, from /root/reference/src

extern "C" {
int ref_edge_table(int c) { return vacancy::marching_cubes_lut::kEdgeTable[c]; }
int ref_tri_table(int c, int k) { return vacancy::marching_cubes_lut::kTriTable[c][k]; }
}

#ifdef REF_LUT_DUMP_MAIN
int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s <out.bin>\n", argv[0]);
    return 2;
  }
  std::FILE* f = std::fopen(argv[1], "wb");
  if (!f) return 1;
  for (int c = 0; c < 256; ++c) {
    const int32_t v = ref_edge_table(c);
    std::fwrite(&v, 4, 1, f);
  }
  for (int c = 0; c < 256; ++c)
    for (int k = 0; k < 16; ++k) {
      const int32_t v = ref_tri_table(c, k);
      std::fwrite(&v, 4, 1, f);
    }
  std::fclose(f);
  return 0;
}
#endif
