// san_driver.cpp -- TEST INFRASTRUCTURE (scripts/r05_sanitize.sh): drives the host layer's threaded and asynchronous paths for the sanitizer
// builds (make SAN=address|thread): loadTree / loadBins, then queryKNN with a large _nVec so that the packed hand-over runs (device
// compaction, pinned staging, the HostPool's padding and scatter phases, the padding memory), alternating two batches; the same through
// queryKNNAsync / queryKNNCollect with two batches in flight; setKeepPadding off; one two-shard object (pqt_multi, host threads of its own).
//   usage: san_driver <dim> <p> <lineparts> <w> <tree> <bins> <queries raw f32> <nq>
#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <string.h>
#include <fstream>
#include <iostream>
#include <vector>
#include "pqt/PerturbationProTree.hh"
using namespace pqt;

int main(int argc, char** argv) {
  if (argc < 9) { std::cerr << "bad usage" << std::endl; return 2; }
  const uint dim = atoi(argv[1]), p = atoi(argv[2]), lp = atoi(argv[3]), w = atoi(argv[4]), nq = atoi(argv[8]);
  try {
    std::vector<float> q((size_t)nq * dim);
    std::ifstream fq(argv[7], std::ios::binary);
    fq.read((char*)q.data(), q.size() * 4);
    if (!fq.good()) throw std::runtime_error("cannot read queries");
    std::vector<float> q2(q.rbegin(), q.rend());
    float *qd = nullptr, *qd2 = nullptr;
    if (hipMalloc((void**)&qd, q.size() * 4) != hipSuccess || hipMalloc((void**)&qd2, q.size() * 4) != hipSuccess) throw std::runtime_error("hipMalloc");
    hipMemcpy(qd, q.data(), q.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(qd2, q2.data(), q.size() * 4, hipMemcpyHostToDevice);
    unsigned long long sum = 0;
    for (int obj = 0; obj < 2; ++obj) {
      PerturbationProTree t(dim, p, p);
      if (obj == 1) t.setDevices(std::vector<int>{0, 0});  // two range shards on this device: the multi handle's own threads
      t.setW(w);
      t.prepareEmptyLambda(0, lp);
      t.loadTree(argv[5]);
      t.loadBins(argv[6]);
      t.setBounds(1500, 400);
      std::vector<uint> ri; std::vector<float> rd;
      for (int r = 0; r < 6; ++r) { t.queryKNN(ri, rd, (r & 1) ? qd2 : qd, nq, 2048); sum += ri[0] + ri[ri.size() / 2]; }
      int pending = -1;
      for (int r = 0; r < 6; ++r) {
        const int tk = t.queryKNNAsync((r & 1) ? qd2 : qd, nq, 2048);
        if (pending >= 0) { t.queryKNNCollect(pending, ri, rd); sum += ri[1]; }
        pending = tk;
      }
      t.queryKNNCollect(pending, ri, rd);
      t.setKeepPadding(false);
      t.queryKNN(ri, rd, qd, nq, 2048);
      t.queryKNN(ri, rd, qd, nq, 16);   // dense hand-over
      std::vector<std::pair<uint, float> > cand;
      t.query(1500, 400, q.data(), cand);
      sum += cand.size();
    }
    (void)hipFree(qd); (void)hipFree(qd2);
    std::cout << "san_driver ok " << sum << std::endl;
  } catch (const std::exception& e) { std::cerr << "san_driver: " << e.what() << std::endl; return 1; }
  return 0;
}
