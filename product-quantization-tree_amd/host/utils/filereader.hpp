// utils/filereader.hpp -- reader of the reference's .umem/.imem/.fmem vector files: ASCII header "<num>\n<dim>\n",
// payload from byte 20 (convert/filehelper.hpp:252-282 writes it; utils/filereader.hpp:58-70 and
// cpu_version/iterator/memiterator.hpp:28-71 read it).  Throws std::runtime_error like the reference.
#ifndef PQT_HOST_FILEREADER_HPP
#define PQT_HOST_FILEREADER_HPP
#include <stdint.h>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

template <typename T /* output type */, typename TT = uint8_t /* stored type */>
class FileReader {
 public:
  explicit FileReader(const std::string& fs) : filename(fs) {
    std::ifstream fin(fs.c_str(), std::ios_base::in | std::ios_base::binary);
    if (!fin.good()) throw std::runtime_error("cannot open file " + fs);
    fin >> n_ >> d_;
    if (!fin.good()) throw std::runtime_error("bad header in " + fs);
  }
  unsigned num() const { return n_; }
  unsigned dim() const { return d_; }
  std::vector<T> data(size_t num, size_t offset = 0) const {
    if (offset + num > n_) throw std::runtime_error("read beyond end of " + filename);
    std::ifstream fin(filename.c_str(), std::ios_base::in | std::ios_base::binary);
    std::vector<TT> raw(num * d_);
    fin.seekg(20 + sizeof(TT) * offset * d_, std::ios::beg);
    fin.read(reinterpret_cast<char*>(raw.data()), raw.size() * sizeof(TT));
    if (!fin.good()) throw std::runtime_error("short read in " + filename);
    return std::vector<T>(raw.begin(), raw.end());
  }
  std::vector<T> data() const { return data(n_, 0); }
  static void write(const std::string& fs, const TT* v, unsigned num, unsigned dim) {
    std::ofstream f(fs.c_str(), std::ios_base::out | std::ios_base::binary);
    if (!f.good()) throw std::runtime_error("cannot open file " + fs);
    std::string hdr = std::to_string(num) + "\n" + std::to_string(dim) + "\n";
    hdr.resize(20, '\0');
    f.write(hdr.data(), 20);
    f.write(reinterpret_cast<const char*>(v), (size_t)num * dim * sizeof(TT));
  }
 private:
  std::string filename;
  unsigned n_ = 0, d_ = 0;
};
#endif
