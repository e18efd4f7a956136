// oracle/ref_cpu_driver.cpp -- TEST INFRASTRUCTURE, part of the oracle-pin recipe (oracle/make_ref_fixtures.sh).
//
// A driver around the UNMODIFIED reference headers (cpu_version/quantizer/treequantizer.hpp and what it includes): loads a
// tree dump, inserts the database vectors with the reference's own insert() (tools/build_db.cpp:24-41), writes the bins with
// its saveBins(), and prints query(boundVectors, boundBins, vec, out) (treequantizer.hpp:323-350) of every query vector.
// The reference headers need the Eigen library, which is neither vendored in the reference nor present in the build image
// (cpu_version/CMakeLists.txt:5,22), so this file is NOT compiled by oracle/Makefile / __graft_entry__.build(); it is compiled
// only by make_ref_fixtures.sh on a machine where a real Eigen exists.  No stand-in for Eigen is used anywhere.
//
// Template parameters: the reference tools' defaults (cpu_version/tools/query.cpp:10-15) = tests/golden/dump_small.*.
//   usage: ref_cpu_driver <in.tree> <base.raw f32> <n> <queries.raw f32> <nq> <boundVectors> <boundBins> <out.bins> <out.lists>
//   out.lists: per query  u32 n, then n x (u32 id, f32 dist)  -- the reference's sorted candidate list.
#include <stdio.h>
#include <stdlib.h>
#include <fstream>
#include <iostream>
#include <vector>
#include "helper.hpp"
#include "iterator/iterator.hpp"
#include "quantizer/treequantizer.hpp"

const uint D = 128, P = 2, C1 = 16, C2 = 8, H1 = 4, RE = 32;
typedef float T;

static std::vector<float> slurp(const char* path, size_t count) {
  std::vector<float> v(count);
  std::ifstream f(path, std::ios::binary);
  f.read((char*)v.data(), count * sizeof(float));
  if (!f.good()) { std::cerr << "cannot read " << path << std::endl; exit(2); }
  return v;
}

int main(int argc, char** argv) {
  if (argc != 10) { std::cerr << "usage: see the header of ref_cpu_driver.cpp" << std::endl; return 2; }
  const size_t n = (size_t)atoll(argv[3]), nq = (size_t)atoll(argv[5]);
  const uint bv = (uint)atoll(argv[6]), bb = (uint)atoll(argv[7]);
  std::vector<float> base = slurp(argv[2], n * D), queries = slurp(argv[4], nq * D);
  treequantizer<T, D, C1, C2, P, H1, RE> Q;
  Q.loadTree(argv[1]);
  Q.notify(n);
  for (size_t i = 0; i < n; ++i) {
    Eigen::Matrix<T, D, 1> v = Eigen::Map<Eigen::Matrix<T, D, 1> >(base.data() + i * D);
    Q.insert(v);
  }
  Q.saveBins(argv[8]);
  std::ofstream out(argv[9], std::ios::binary);
  for (size_t i = 0; i < nq; ++i) {
    Eigen::Matrix<T, D, 1> v = Eigen::Map<Eigen::Matrix<T, D, 1> >(queries.data() + i * D);
    std::vector<std::pair<uint, T> > cand;
    Q.query(bv, bb, v, cand);
    const uint m = (uint)cand.size();
    out.write((const char*)&m, 4);
    for (uint j = 0; j < m; ++j) { out.write((const char*)&cand[j].first, 4); out.write((const char*)&cand[j].second, 4); }
  }
  std::cout << "reference run: " << n << " vectors inserted, " << nq << " queries" << std::endl;
  return 0;
}
