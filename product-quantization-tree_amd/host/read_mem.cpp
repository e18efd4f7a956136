// read_mem.cpp -- reads a .umem / .imem vector file through the host layer's FileReader (utils/filereader.hpp, the reader tool_query and
// tool_createdb use) and writes the values as raw little-endian words: the tests compare them with what the reference's own readers
// (utils/filereader.hpp:7-136, convert/filehelper.hpp:284-319 of the reference) returned for the same bytes (tests/golden/ref_formats.npz).
//   read_mem --in x.umem --as f32|u8|i32 [--num N --offset O] --out raw.bin      prints "<num> <dim>" of the header
#include <stdint.h>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include "utils/filereader.hpp"

template <class T, class TT>
static int run(const std::string& in, const std::string& out, long num, long off) {
  FileReader<T, TT> r(in);
  std::cout << r.num() << " " << r.dim() << std::endl;
  const std::vector<T> v = num < 0 ? r.data() : r.data((size_t)num, (size_t)off);
  FILE* f = fopen(out.c_str(), "wb");
  if (!f) return 2;
  fwrite(v.data(), sizeof(T), v.size(), f);
  fclose(f);
  return 0;
}

int main(int argc, char** argv) {
  std::string in, out, as = "f32";
  long num = -1, off = 0;
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::string k = argv[i], v = argv[i + 1];
    if (k == "--in") in = v; else if (k == "--out") out = v; else if (k == "--as") as = v;
    else if (k == "--num") num = atol(v.c_str()); else if (k == "--offset") off = atol(v.c_str());
    else { std::cerr << "unknown flag " << k << std::endl; return 1; }
  }
  try {
    if (as == "f32") return run<float, uint8_t>(in, out, num, off);   // FileReader<float>: uint8 payload widened (reference utils/filereader.hpp:33-49)
    if (as == "u8") return run<uint8_t, uint8_t>(in, out, num, off);
    if (as == "i32") return run<int32_t, int32_t>(in, out, num, off);  // FileReader<int> (:77-136)
    std::cerr << "--as f32|u8|i32" << std::endl;
    return 1;
  } catch (const std::exception& e) {
    std::cerr << "read_mem: " << e.what() << std::endl;
    return 3;
  }
}
