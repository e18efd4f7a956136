// tool_query -- front-end kept from the reference (tool_query.cpp): same flags, same file naming, same call order
// (read queries -> readTreeFromFile -> load DB -> batches of <=4096 queries -> queryKNN), on the HIP engine.
// Prints what the reference's harnesses print: avg. query time and recall @R (cpu_version/tools/query.cpp:29-82,133-138)
// when --groundtruth is given.
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <sys/stat.h>
#include <chrono>
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
  F.def("chunksize", "100000", "number of query vectors");
  F.def("hashsize", "400000000", "maximal number of bins");
  F.def("basename", "tmp", "prefix for generated data");
  F.def("dataset", "base.umem", "path to vector dataset (unused by the query)");
  F.def("queryset", "query.umem", "path to query vectors");
  F.def("groundtruth", "", "optional .imem with the true neighbours (prints recall)");
  F.def("w", "2", "first-level cells expanded per part");
  F.def("boundvectors", "20000", "query(boundVectors, .)");
  F.def("boundbins", "500", "query(., boundBins)");
  F.def("hashed", "0", "1: load the CUDA library's dump family even when a .bins dump exists");
  F.def("nvec", "4096", "results per query (queryKNN _nVec)");
  F.def("sync", "0", "1: one batch at a time (queryKNN) like the reference's loop; 0: the next batch is issued before the current one is collected");
  F.def("gpus", "1", "range-shard the database over the devices 0 .. gpus-1 of this node (one process, one handle)");
  F.def("devices", "", "explicit device list for the shards, e.g. 0,1,2,3 (overrides --gpus / --device; a device may repeat)");
  if (!F.parse(argc, argv)) return 1;
  try {
    const uint dim = F.num("dim"), p = F.num("p"), c1 = F.num("c1"), c2 = F.num("c2"), lp = F.num("lineparts");
    const std::string pre = F.str("basename") + "_" + std::to_string(dim) + "_" + std::to_string(p) + "_" + std::to_string(c1) + "_" + std::to_string(c2);
    FileReader<float> qr(F.str("queryset"));
    if (qr.dim() != dim) { std::cerr << "query dim mismatch" << std::endl; return 1; }
    const size_t qn = std::min<size_t>(qr.num(), (size_t)F.num("chunksize"));
    std::vector<float> qh = qr.data(qn);
    PerturbationProTree ppt(dim, p, p);
    ppt.setDevice((int)F.num("device"));
    int queryDevice = (int)F.num("device");  // where the query batch is uploaded: the (first) device of the database
    {
      std::vector<int> devs;
      const std::string dl = F.str("devices");
      for (size_t a = 0; a < dl.size();) { size_t b = dl.find(',', a); if (b == std::string::npos) b = dl.size(); if (b > a) devs.push_back(atoi(dl.substr(a, b - a).c_str())); a = b + 1; }
      if (devs.empty() && F.num("gpus") > 1) for (int g = 0; g < (int)F.num("gpus"); ++g) devs.push_back(g);
      if (devs.size() > 1) { ppt.setDevices(devs); std::cout << "database range-sharded over " << devs.size() << " devices" << std::endl; }
      else if (devs.size() == 1) { ppt.setDevice(devs[0]); queryDevice = devs[0]; }  // --devices N overrides --device like the longer lists do
      if (devs.size() > 1) queryDevice = devs[0];
    }
    ppt.setW((uint)F.num("w"));
    ppt.prepareEmptyLambda(0, lp);
    ppt.setBounds((uint)F.num("boundvectors"), (uint)F.num("boundbins"));
    const std::string cb = pre + ".ppqt";
    if (!file_exists(cb)) { std::cout << "you need to generate a codebook first. No codebook found in " << cb << std::endl; return 1; }
    std::cout << "codebook exists, reading from " << cb << std::endl;
    ppt.readTreeFromFile(cb);
    // database: the cpu_version dump if present, else the CUDA library's dump family (tool_query.cpp:104-147)
    if (file_exists(pre + ".bins") && !F.num("hashed")) {
      ppt.loadBins(pre + ".bins");
      std::cout << "read " << pre << ".bins" << std::endl;
    } else {
      struct stat sb;
      if (stat((pre + ".dbIdx").c_str(), &sb) != 0) { std::cout << "no database dump found (" << pre << ".bins or .prefix/.count/.dbIdx/.lines)" << std::endl; return 1; }
      const uint nbase = (uint)(sb.st_size / 4);
      ppt.loadHashedDB(pre, nbase, (uint)F.num("hashsize"));
      std::cout << "read " << pre << ".prefix" << std::endl << "read " << pre << ".count" << std::endl << "read " << pre << ".dbIdx" << std::endl
                << "read " << pre << "_" << lp << ".lines" << std::endl;
    }
    if (hipSetDevice(queryDevice) != hipSuccess) { std::cerr << "no device" << std::endl; return 1; }
    float* qd = nullptr;
    if (hipMalloc((void**)&qd, qh.size() * 4) != hipSuccess || hipMemcpy(qd, qh.data(), qh.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
      std::cerr << "query upload failed" << std::endl; return 1;
    }
    const uint nvec = (uint)F.num("nvec");
    std::vector<uint> resIdx, all((size_t)qn * nvec);
    std::vector<float> resDist;
    // resIdx / resDist live for the whole loop and are only read between calls: the padding the previous hand-over left is still there
    ppt.setKeepPadding(true);
    auto t0 = std::chrono::steady_clock::now();
    // the reference's loop (tool_query.cpp:152-160) answers one batch at a time; here batch i + 1 is issued before batch i is collected, so
    // its kernels run under batch i's copies and the host-side scatter (PerturbationProTree::queryKNNAsync / queryKNNCollect); --sync 1 =
    // the reference's form
    const bool syncLoop = F.num("sync") != 0;
    int pending = -1;
    size_t pendA = 0;
    for (size_t a = 0; a < qn; a += 4096) {
      const uint len = (uint)std::min<size_t>(4096, qn - a);
      if (syncLoop) {
        ppt.queryKNN(resIdx, resDist, qd + a * dim, len, nvec);
        std::copy(resIdx.begin(), resIdx.end(), all.begin() + a * nvec);
        continue;
      }
      const int tk = ppt.queryKNNAsync(qd + a * dim, len, nvec);
      if (pending >= 0) {
        ppt.queryKNNCollect(pending, resIdx, resDist);
        std::copy(resIdx.begin(), resIdx.end(), all.begin() + pendA * nvec);
      }
      pending = tk; pendA = a;
    }
    if (pending >= 0) {
      ppt.queryKNNCollect(pending, resIdx, resDist);
      std::copy(resIdx.begin(), resIdx.end(), all.begin() + pendA * nvec);
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    (void)hipFree(qd);
    std::cout << "avg. query time   " << ms / qn << "ms" << std::endl;
    std::cout << "total. query time " << ms / 1e3 << "s" << std::endl;
    if (!F.str("groundtruth").empty()) {
      FileReader<int, int> gt(F.str("groundtruth"));
      std::vector<int> g = gt.data(qn);
      const uint Rs[6] = {1, 10, 100, 1000, 10000, 100000};
      for (uint R : Rs) {
        size_t good = 0;
        for (size_t q = 0; q < qn; ++q) {
          const uint want = (uint)g[q * gt.dim()];
          for (uint s = 0; s < std::min(R, nvec); ++s) if (all[q * nvec + s] == want) { ++good; break; }
        }
        std::cout << "@R" << R << ": " << (double)good / qn << std::endl;
      }
    }
  } catch (const std::exception& e) {
    std::cerr << "tool_query: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
