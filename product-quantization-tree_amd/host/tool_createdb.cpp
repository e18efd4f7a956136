// tool_createdb -- front-end kept from the reference (tool_createdb.cpp): same flags and output file names.
//   <basename>_<dim>_<p>_<c1>_<c2>.ppqt            tree: read if present, else trained on the first --train vectors
//   ....bins                                        the cpu_version dump (exact bins + line codes, treequantizer::saveBins)
//   ..._<lineparts>.lines / .prefix / .count / .dbIdx   the CUDA library's dumps (dense hashed CSR of --hashsize slots +
//                                                   4-byte line codes in vector-id order; tool_createdb.cpp:111-138),
//                                                   written unless --hashed 0
// The WHOLE dataset is processed, --chunksize vectors at a time (per-chunk assign + line-encode on the GPU through the C-ABI,
// then one host-side CSR merge: test/test1B.cpp:783-871); the reference's tool stops after its first chunk.
#include <stdio.h>
#include <sys/stat.h>
#include <fstream>
#include <iostream>
#include "flags.hpp"
#include "pqt/PerturbationProTree.hh"
#include "utils/filereader.hpp"

using namespace pqt;
static bool file_exists(const std::string& n) { struct stat b; return stat(n.c_str(), &b) == 0; }

int main(int argc, char* argv[]) {
  Flags F;
  F.def("device", "0", "selected HIP device");
  F.def("c1", "4", "number of clusters in first level");
  F.def("c2", "4", "number of refinements in second level");
  F.def("p", "2", "parts per vector");
  F.def("dim", "128", "expected dimension for each vector");
  F.def("lineparts", "32", "vectorparts for reranking informations");
  F.def("chunksize", "10000000", "number of vectors per chunk");
  F.def("hashsize", "400000000", "maximal number of bins");
  F.def("basename", "tmp", "prefix for generated data");
  F.def("dataset", "base.umem", "path to vector dataset");
  F.def("w", "2", "first-level cells expanded per part (treequantizer W)");
  F.def("hashed", "1", "also write the CUDA library's dump family (.prefix/.count/.dbIdx/.lines)");
  F.def("train", "0", "if > 0 and no codebook exists: createTree on the first <train> vectors (the reference trains on 20000, tool_createdb.cpp:76)");
  if (!F.parse(argc, argv)) return 1;
  try {
    const uint dim = F.num("dim"), p = F.num("p"), c1 = F.num("c1"), c2 = F.num("c2"), lp = F.num("lineparts");
    const std::string pre = F.str("basename") + "_" + std::to_string(dim) + "_" + std::to_string(p) + "_" + std::to_string(c1) + "_" + std::to_string(c2);
    FileReader<float> reader(F.str("dataset"));
    if (reader.dim() != dim) { std::cerr << "dataset dim " << reader.dim() << " != --dim " << dim << std::endl; return 1; }
    PerturbationProTree ppt(dim, p, p);
    ppt.setDevice((int)F.num("device"));
    ppt.setW((uint)F.num("w"));
    ppt.prepareEmptyLambda(0, lp);
    const std::string cb = pre + ".ppqt";
    if (!file_exists(cb)) {
      const size_t nt = std::min<size_t>((size_t)F.num("train"), reader.num());
      if (nt == 0) { std::cout << "you need to generate a codebook first. No codebook found in " << cb << std::endl; return 1; }
      std::cout << "building the codebook from " << nt << " vectors" << std::endl;
      std::vector<float> tr = reader.data(nt);
      ppt.createTree(c1, c2, tr.data(), (uint)nt);
      ppt.writeTreeToFile(cb);
      std::cout << "written " << cb << std::endl;
    } else {
      std::cout << "codebook exists, reading from " << cb << std::endl;
      ppt.readTreeFromFile(cb);
    }
    if (ppt.getNClusters() != c1 || ppt.getClusters2() != c2) { std::cerr << "codebook c1/c2 differ from flags" << std::endl; return 1; }
    const size_t n = reader.num(), chunk = std::max<size_t>(1, (size_t)F.num("chunksize"));
    size_t nchunks = 0;
    for (size_t off = 0; off < n; off += chunk, ++nchunks) {
      const size_t m = std::min(chunk, n - off);
      std::vector<float> data = reader.data(m, off);
      ppt.buildKBestDBChunk(data.data(), (uint)m, (uint)off);
    }
    ppt.finishDB();
    ppt.saveBins(pre + ".bins");
    std::cout << "written " << pre << ".bins" << std::endl;
    if (F.num("hashed")) {
      ppt.saveHashedDB(pre, (uint)F.num("hashsize"));
      std::cout << "written " << pre << "_" << lp << ".lines" << std::endl << "written " << pre << ".prefix" << std::endl
                << "written " << pre << ".count" << std::endl << "written " << pre << ".dbIdx" << std::endl;
    }
    std::cout << "vectors " << n << "  chunks " << nchunks << "  bins " << ppt.binIds().size() << std::endl;
  } catch (const std::exception& e) {
    std::cerr << "tool_createdb: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
