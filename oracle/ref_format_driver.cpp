// oracle/ref_format_driver.cpp -- TEST INFRASTRUCTURE ONLY.
// Thin extern "C" driver around the GENUINE reference file-format headers, compiled from where they lie under
// /root/reference (oracle/Makefile):
//   default build            : convert/filehelper.hpp (readJegou / readJegouHeader / readBatchJegou / write / read / header,
//                              :8-319) + utils/filereader.hpp (FileReader<T>, FileReader<int>, readFloat, :7-180)
//   -DREF_FORMAT_CPU_VERSION : cpu_version/filehelper.hpp (the same function names; its own copy of the code)
// The headers need only the standard library.  They rely on their includers for <fstream>, <cstdint>, `uint` and
// `using namespace std` (tool_query.cpp:1-20 and convert/convert_fvecs.cpp:1-12 provide them the same way), so the lines
// below are what a reference translation unit has in front of the include -- no stand-in for anything the image lacks.
// Output goes to oracle/_ref/ only.  Pins the product's reading of .umem / .imem / .fmem and of the TEXMEX vecs files
// (tests/golden/ref_formats.npz, tests/golden/make_golden.py).
#include <sys/types.h>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
using namespace std;
#include "filehelper.hpp"
#ifndef REF_FORMAT_CPU_VERSION
#include "filereader.hpp"
#endif

namespace {
template <typename F> int guarded(F f) {
  try { f(); return 0; } catch (const std::exception&) { return 1; } catch (...) { return 2; }
}
}  // namespace

extern "C" {
// ---- convert/filehelper.hpp:252-319 (cpu_version/filehelper.hpp:260-330) ----
int reffmt_write_u8(const char* fs, unsigned num, unsigned dim, uint8_t* p, unsigned len, unsigned off) { return guarded([&] { write<uint8_t>(fs, num, dim, p, len, off); }); }
int reffmt_write_f32(const char* fs, unsigned num, unsigned dim, float* p, unsigned len, unsigned off) { return guarded([&] { write<float>(fs, num, dim, p, len, off); }); }
int reffmt_write_i32(const char* fs, unsigned num, unsigned dim, int* p, unsigned len, unsigned off) { return guarded([&] { write<int>(fs, num, dim, p, len, off); }); }
int reffmt_read_u8(const char* fs, unsigned* num, unsigned* dim, uint8_t* p, unsigned len, unsigned off) { return guarded([&] { read<uint8_t>(fs, *num, *dim, p, len, off); }); }
int reffmt_read_f32(const char* fs, unsigned* num, unsigned* dim, float* p, unsigned len, unsigned off) { return guarded([&] { read<float>(fs, *num, *dim, p, len, off); }); }
int reffmt_read_i32(const char* fs, unsigned* num, unsigned* dim, int* p, unsigned len, unsigned off) { return guarded([&] { read<int>(fs, *num, *dim, p, len, off); }); }
int reffmt_header(const char* fs, unsigned* num, unsigned* dim) { return guarded([&] { header(fs, *num, *dim); }); }
// ---- TEXMEX readers, convert/filehelper.hpp:8-250 ----
int reffmt_jegou_header_f32(const char* path, unsigned* n, unsigned* d) { return guarded([&] { readJegouHeader<float>(path, *n, *d); }); }
int reffmt_jegou_header_i32(const char* path, unsigned* n, unsigned* d) { return guarded([&] { readJegouHeader<int>(path, *n, *d); }); }
int reffmt_jegou_header_u8(const char* path, unsigned* n, unsigned* d) { return guarded([&] { readJegouHeader<uint8_t>(path, *n, *d); }); }
// out must hold n * d elements (take them from the header call)
int reffmt_jegou_f32(const char* path, float* out, unsigned* n, unsigned* d) {
  return guarded([&] { float* p = readJegou<float>(path, *n, *d); std::memcpy(out, p, (size_t)*n * *d * 4); delete[] p; });
}
int reffmt_jegou_i32(const char* path, int* out, unsigned* n, unsigned* d) {
  return guarded([&] { int* p = readJegou<int>(path, *n, *d); std::memcpy(out, p, (size_t)*n * *d * 4); delete[] p; });
}
int reffmt_jegou_u8(const char* path, uint8_t* out, unsigned* n, unsigned* d) {
  return guarded([&] { uint8_t* p = readJegou<uint8_t>(path, *n, *d); std::memcpy(out, p, (size_t)*n * *d); delete[] p; });
}
// readBatchJegou (:167-222; record size hard-wired to 132 bytes = 128-dimensional bvecs): out must hold num * d bytes
int reffmt_jegou_batch_u8(const char* path, unsigned start, unsigned num, unsigned d, uint8_t* out) {
  return guarded([&] { uint8_t* p = readBatchJegou(path, start, num); std::memcpy(out, p, (size_t)num * d); delete[] p; });
}
#ifndef REF_FORMAT_CPU_VERSION
// ---- utils/filereader.hpp:7-75 (FileReader<T>: uint8 payload widened to T), :77-136 (FileReader<int>), :163-180 (readFloat) ----
int reffmt_filereader_f32(const char* fs, float* out, unsigned* n, unsigned* d, size_t num, size_t off) {
  return guarded([&] {
    FileReader<float> r(fs);
    *n = r.num(); *d = r.dim();
    if (out) { float* p = r.data(num, off); std::memcpy(out, p, num * r.dim() * 4); delete[] p; }
  });
}
int reffmt_filereader_u8(const char* fs, uint8_t* out, unsigned* n, unsigned* d, size_t num, size_t off) {
  return guarded([&] {
    FileReader<uint8_t> r(fs);
    *n = r.num(); *d = r.dim();
    if (out) { uint8_t* p = r.data(num, off); std::memcpy(out, p, num * r.dim()); delete[] p; }
  });
}
int reffmt_filereader_i32(const char* fs, int* out, unsigned* n, unsigned* d, size_t num, size_t off) {
  return guarded([&] {
    FileReader<int> r(fs);
    *n = r.entries(); *d = r.dimension();
    if (out) { int* p = r.data(num, off); std::memcpy(out, p, num * r.dimension() * 4); delete[] p; }
  });
}
#endif
}
