// test_classes.cpp -- exercises the reference-shaped class surface (pqt::PerturbationProTree) the way the reference's
// cpu_version tools use treequantizer: loadTree -> loadBins -> query(boundVectors, boundBins, vec, out) per vector,
// saveTree/saveBins round trip, and the batch entry queryKNN.  Driven by tests/test_gpu_tools.py, which compares the
// dumped results with the oracle.
//   usage: test_classes <dim> <p> <lineparts> <w> <tree> <bins> <queries.fmem-like raw f32 file> <nq> <bv> <bb> <out.bin> [<base.raw> <n> <hashsize>]
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fstream>
#include <iostream>
#include <vector>
#include "pqt/PerturbationProTree.hh"

using namespace pqt;

int main(int argc, char** argv) {
  if (argc < 12) { std::cerr << "bad usage" << std::endl; return 2; }
  const uint dim = atoi(argv[1]), p = atoi(argv[2]), lp = atoi(argv[3]), w = atoi(argv[4]);
  const std::string tree = argv[5], bins = argv[6], qfile = argv[7], out = argv[11];
  const uint nq = atoi(argv[8]), bv = atoi(argv[9]), bb = atoi(argv[10]);
  try {
    std::vector<float> q((size_t)nq * dim);
    std::ifstream fq(qfile.c_str(), std::ios::binary);
    fq.read((char*)q.data(), q.size() * 4);
    if (!fq.good()) throw std::runtime_error("cannot read queries");
    PerturbationProTree t(dim, p, p);
    t.setW(w);
    t.prepareEmptyLambda(0, lp);
    // error behaviour of the reference's readers: missing file -> std::runtime_error
    bool threw = false;
    try { t.loadTree(tree + ".does-not-exist"); } catch (const std::runtime_error&) { threw = true; }
    if (!threw) throw std::runtime_error("loadTree did not throw on a missing file");
    t.loadTree(tree);
    t.loadBins(bins);
    // round trip of both dumps
    t.saveTree(out + ".tree");
    t.saveBins(out + ".bins");
    std::ofstream fo(out.c_str(), std::ios::binary);
    // per-vector query(): the whole sorted candidate list
    for (uint i = 0; i < nq; ++i) {
      std::vector<std::pair<uint, float> > cand;
      t.query(bv, bb, q.data() + (size_t)i * dim, cand);
      const uint n = (uint)cand.size();
      fo.write((const char*)&n, 4);
      for (auto& c : cand) { fo.write((const char*)&c.first, 4); fo.write((const char*)&c.second, 4); }
    }
    // batch queryKNN with a device pointer, like tool_query
    float* qd = nullptr;
    if (hipMalloc((void**)&qd, q.size() * 4) != hipSuccess || hipMemcpy(qd, q.data(), q.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
      throw std::runtime_error("upload failed");
    t.setBounds(bv, bb);
    std::vector<uint> ri; std::vector<float> rd;
    t.queryKNN(ri, rd, qd, nq, 16);
    (void)hipFree(qd);
    fo.write((const char*)ri.data(), ri.size() * 4);
    fo.write((const char*)rd.data(), rd.size() * 4);
    // the same dumps behind ONE object over two range shards (both on device 0 here): queryKNN and query() must return what
    // the single-device object returned, bit for bit
    {
      PerturbationProTree tm(dim, p, p);
      tm.setDevices(std::vector<int>(2, 0));
      tm.setW(w);
      tm.prepareEmptyLambda(0, lp);
      tm.loadTree(tree);
      tm.loadBins(bins);
      tm.setBounds(bv, bb);
      float* qd2 = nullptr;
      if (hipMalloc((void**)&qd2, q.size() * 4) != hipSuccess || hipMemcpy(qd2, q.data(), q.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
        throw std::runtime_error("upload failed");
      std::vector<uint> mi; std::vector<float> md;
      tm.queryKNN(mi, md, qd2, nq, 16);
      (void)hipFree(qd2);
      if (mi != ri || memcmp(md.data(), rd.data(), rd.size() * 4) != 0) throw std::runtime_error("two-shard queryKNN differs from the single-device result");
      for (uint i = 0; i < nq; ++i) {
        std::vector<std::pair<uint, float> > a, b;
        t.query(bv, bb, q.data() + (size_t)i * dim, a);
        tm.query(bv, bb, q.data() + (size_t)i * dim, b);
        if (a.size() != b.size()) throw std::runtime_error("two-shard query(): list length differs");
        for (size_t j = 0; j < a.size(); ++j)
          if (a[j].first != b[j].first || memcmp(&a[j].second, &b[j].second, 4) != 0) throw std::runtime_error("two-shard query(): list differs");
      }
      bool refused = false;
      try { (void)tm.getDBIdx(); } catch (const std::runtime_error&) { refused = true; }
      if (!refused) throw std::runtime_error("getDBIdx with several devices did not refuse");
      std::cout << "multi ok " << tm.getNDevices() << std::endl;
      // p = 4: the 1B path's 2-D anisotropic sequences with the reference's call (test/test1B.cpp:941 prepare2DDistSequence(512)) on both
      // objects: identical results, written to <out>.2d for the caller's comparison with the checker; the default table is set again after
      if (p == 4) {
        t.prepare2DDistSequence(512);
        tm.prepare2DDistSequence(512);
        float* qd3 = nullptr;
        if (hipMalloc((void**)&qd3, q.size() * 4) != hipSuccess || hipMemcpy(qd3, q.data(), q.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
          throw std::runtime_error("upload failed");
        std::vector<uint> ai, bi; std::vector<float> ad, bd2;
        t.queryKNN(ai, ad, qd3, nq, 16);
        tm.queryKNN(bi, bd2, qd3, nq, 16);
        (void)hipFree(qd3);
        if (ai != bi || memcmp(ad.data(), bd2.data(), ad.size() * 4) != 0) throw std::runtime_error("two-shard queryKNN differs under the 2-D sequences");
        std::ofstream f2((out + ".2d").c_str(), std::ios::binary);
        f2.write((const char*)ai.data(), ai.size() * 4);
        f2.write((const char*)ad.data(), ad.size() * 4);
        t.prepareDistSequence((uint)bb);
        std::cout << "2d ok" << std::endl;
      }
    }
    // optional: <base.raw f32> <n> <hashsize> -- the device-pointer overloads and the reference's device getters
    if (argc >= 15) {
      const std::string bfile = argv[12];
      const uint n = atoi(argv[13]), hs = atoi(argv[14]);
      std::vector<float> base((size_t)n * dim);
      std::ifstream fb(bfile.c_str(), std::ios::binary);
      fb.read((char*)base.data(), base.size() * 4);
      if (!fb.good()) throw std::runtime_error("cannot read base vectors");
      float* bd = nullptr;
      if (hipMalloc((void**)&bd, base.size() * 4) != hipSuccess || hipMemcpy(bd, base.data(), base.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
        throw std::runtime_error("upload failed");
      PerturbationProTree t2(dim, p, p);
      t2.setW(w);
      t2.prepareEmptyLambda(0, lp);
      t2.loadTree(tree);
      t2.buildKBestDB(bd, n, DEVICE_PTR);  // device pointer, like the reference (PerturbationProTree.hh:53)
      (void)hipFree(bd);
      t2.saveBins(out + ".devbuild.bins");
      t2.saveHashedDB(out + ".h", hs);
      // getters hand out DEVICE arrays (PerturbationProTree.hh:97-103)
      std::vector<uint> dbidx(n), prefix(hs), counts(hs), codes((size_t)n * lp), codesBin((size_t)n * lp);
      if (hipMemcpy(dbidx.data(), t2.getDBIdx(), (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(codes.data(), t2.getLine(), codes.size() * 4, hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(codesBin.data(), t2.getLineBinOrder(), codesBin.size() * 4, hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(prefix.data(), t2.getBinPrefix(hs), (size_t)hs * 4, hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(counts.data(), t2.getBinCounts(hs), (size_t)hs * 4, hipMemcpyDeviceToHost) != hipSuccess)
        throw std::runtime_error("getter read-back failed");
      std::ofstream g((out + ".getters").c_str(), std::ios::binary);
      g.write((const char*)dbidx.data(), dbidx.size() * 4);
      g.write((const char*)codes.data(), codes.size() * 4);
      g.write((const char*)prefix.data(), prefix.size() * 4);
      g.write((const char*)counts.data(), counts.size() * 4);
      g.write((const char*)codesBin.data(), codesBin.size() * 4);
      // after a hashed load the exact bin ids are gone: the dense getters must refuse instead of handing out stale arrays
      PerturbationProTree t3(dim, p, p);
      t3.setW(w);
      t3.prepareEmptyLambda(0, lp);
      t3.loadTree(tree);
      t3.loadBins(bins);
      (void)t3.getBinPrefix(hs);
      t3.loadHashedDB(out + ".h", n, hs);
      bool refused = false;
      try { (void)t3.getBinPrefix(hs); } catch (const std::runtime_error&) { refused = true; }
      if (!refused) throw std::runtime_error("getBinPrefix after loadHashedDB returned a stale array");
    }
    std::cout << "ok " << t.getNClusters() << " " << t.getClusters2() << std::endl;
  } catch (const std::exception& e) {
    std::cerr << "test_classes: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
