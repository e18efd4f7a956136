// convert/convert_vecs.cpp -- one source for the three converters of the reference (convert/convert_fvecs.cpp,
// convert_bvecs.cpp, convert_ivecs.cpp): TEXMEX *.fvecs / *.bvecs / *.ivecs  ->  *.umem / *.imem
// (ASCII header "<num>\n<dim>\n", payload at byte 20; convert/filehelper.hpp:252-282).
// Built three times with -DPQT_CONVERT_{FVECS,BVECS,IVECS}; flag names are the reference's.
//   fvecs: the reference writes the FLOAT payload into the .umem although every reader expects uint8
//   (SURVEY 3.4); here the floats are range-checked and stored as uint8, which is what the readers consume.
#include <stdint.h>
#include <cmath>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <vector>
#include "../flags.hpp"

#if defined(PQT_CONVERT_FVECS)
typedef float in_t; typedef uint8_t out_t;
static const char* kIn = "fvecs"; static const char* kOut = "umem";
#elif defined(PQT_CONVERT_BVECS)
typedef uint8_t in_t; typedef uint8_t out_t;
static const char* kIn = "bvecs"; static const char* kOut = "umem";
#else
typedef int32_t in_t; typedef int32_t out_t;
static const char* kIn = "ivecs"; static const char* kOut = "imem";
#endif

int main(int argc, char* argv[]) {
  Flags F;
  F.def(kIn, std::string("in.") + kIn, "(input) path to the TEXMEX file");
  F.def(kOut, std::string("/tmp/out.") + kOut, "(output) path to the converted file");
  F.def("chunkSize", "100000", "number of vectors per chunk");
  if (!F.parse(argc, argv)) return 1;
  try {
    std::ifstream in(F.str(kIn).c_str(), std::ios::binary);
    if (!in.good()) throw std::runtime_error("cannot open file " + F.str(kIn));
    int32_t dim = 0;
    in.read((char*)&dim, 4);
    if (!in.good() || dim <= 0) throw std::runtime_error("bad header");
    in.seekg(0, std::ios::end);
    const uint64_t bytes = (uint64_t)in.tellg(), rec = 4 + (uint64_t)dim * sizeof(in_t);
    if (bytes % rec) throw std::runtime_error("file size is not a multiple of the record size");
    const uint64_t num = bytes / rec;
    std::cout << "header dim " << dim << std::endl << "header num " << num << std::endl;
    std::ofstream out(F.str(kOut).c_str(), std::ios::binary);
    if (!out.good()) throw std::runtime_error("cannot open file " + F.str(kOut));
    std::string hdr = std::to_string(num) + "\n" + std::to_string(dim) + "\n";
    hdr.resize(20, '\0');
    out.write(hdr.data(), 20);
    in.seekg(0, std::ios::beg);
    const uint64_t chunk = (uint64_t)F.num("chunkSize");
    std::vector<char> raw(chunk * rec);
    std::vector<out_t> conv(chunk * dim);
    for (uint64_t done = 0; done < num; done += chunk) {
      const uint64_t n = std::min(chunk, num - done);
      in.read(raw.data(), n * rec);
      if (!in.good()) throw std::runtime_error("short read");
      for (uint64_t i = 0; i < n; ++i) {
        if (*(const int32_t*)(raw.data() + i * rec) != dim) throw std::runtime_error("inconsistent dimension in record");
        const in_t* v = (const in_t*)(raw.data() + i * rec + 4);
        for (int32_t d = 0; d < dim; ++d) {
#if defined(PQT_CONVERT_FVECS)
          const float x = v[d];
          if (!(x >= 0.f && x <= 255.f) || std::floor(x) != x) throw std::runtime_error("fvecs value is not an integer in [0,255]: cannot be stored in .umem");
          conv[i * dim + d] = (uint8_t)x;
#else
          conv[i * dim + d] = (out_t)v[d];
#endif
        }
      }
      out.write((const char*)conv.data(), n * dim * sizeof(out_t));
    }
    std::cout << "written " << F.str(kOut) << std::endl;
  } catch (const std::exception& e) {
    std::cerr << "convert: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
