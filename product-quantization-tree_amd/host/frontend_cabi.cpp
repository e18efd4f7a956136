// frontend_cabi.cpp -- libpqt_frontend.so: a C wrapper around pqt::PerturbationProTree so that bench.py / the tests can time and check
// the KEPT C++ front-end (the call the reference's tool_query makes per batch, tool_query.cpp:153-161: queryKNN with a device query
// pointer and two std::vectors that are resized and filled) from Python without going through files.  Test/bench glue: the product
// surface is the class (host/pqt/PerturbationProTree.hh) and the C-ABI under it (include/pqt_hip.h).
#include <hip/hip_runtime_api.h>
#include <string.h>
#include <chrono>
#include <stdexcept>
#include <string>
#include <vector>
#include "pqt/PerturbationProTree.hh"

using namespace pqt;

namespace {
thread_local std::string g_err;
struct Fe {
  PerturbationProTree t;
  std::vector<uint> idx;
  std::vector<float> dist;
  Fe(uint dim, uint p) : t(dim, p, p) { t.setKeepPadding(true); }  // idx / dist are owned here for the object's life (pqtfe_set_keep_padding switches it)
};
}  // namespace

extern "C" {

const char* pqtfe_last_error(void) { return g_err.c_str(); }

// tree + database handed over as host arrays (the class copies them, like setTree / setBins / setLines); ndev > 1: range-sharded
void* pqtfe_create(uint32_t dim, uint32_t p, uint32_t c1, uint32_t c2, uint32_t w, uint32_t lp, const float* cb1, const float* cb2,
                   uint64_t nbins, const uint32_t* binIds, const uint32_t* binSizes, const uint32_t* members, const uint32_t* codes, uint64_t nvec,
                   const int* devices, int ndev) {
  try {
    Fe* f = new Fe(dim, p);
    if (ndev > 1) f->t.setDevices(std::vector<int>(devices, devices + ndev));
    else f->t.setDevice(ndev == 1 ? devices[0] : 0);
    f->t.setW(w);
    f->t.prepareEmptyLambda(0, lp);
    f->t.setTree(c1, c2, cb1, cb2);
    f->t.setBins((size_t)nbins, binIds, binSizes, members);
    f->t.setLines(reinterpret_cast<const lineDescr*>(codes), (size_t)nvec);
    return f;
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

void pqtfe_destroy(void* h) { delete static_cast<Fe*>(h); }

// `reps` calls of queryKNN(resIdx, resDist, q_dev, qn, nvec) on the SAME two vectors (tool_query's loop reuses them); timing[7] =
// means over the calls of {total, kernels, d2h, host} ms, bytes over PCIe and columns copied per row; the vectors of the last call are
// copied to out_idx / out_dist when those are not null
int pqtfe_queryKNN(void* h, const float* q_dev, uint32_t qn, uint32_t nvec, uint32_t bv, uint32_t bb, int reps, double* timing, uint32_t* out_idx,
                   float* out_dist) {
  try {
    Fe* f = static_cast<Fe*>(h);
    f->t.setBounds(bv, bb);
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
      f->t.queryKNN(f->idx, f->dist, q_dev, qn, nvec);
      const PerturbationProTree::CallTiming& c = f->t.lastCallTiming();
      acc[0] += c.total_ms; acc[1] += c.kernels_ms; acc[2] += c.d2h_ms; acc[3] += c.host_ms; acc[4] += (double)c.d2h_bytes; acc[5] += c.columns; acc[6] += c.packed ? 1.0 : 0.0;
    }
    if (timing) for (int i = 0; i < 7; ++i) timing[i] = acc[i] / (reps > 0 ? reps : 1);
    if (out_idx) memcpy(out_idx, f->idx.data(), f->idx.size() * 4);
    if (out_dist) memcpy(out_dist, f->dist.data(), f->dist.size() * 4);
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// the same `reps` batches with TWO in flight (queryKNNAsync / queryKNNCollect, the loop of host/tool_query.cpp): batch r + 1 is issued before
// batch r is collected; the batches alternate between q_dev and q_dev2 (q_dev2 may equal q_dev).  wall_ms[0] = wall clock per batch over the
// whole loop; the vectors of the LAST batch (q_dev2 when reps is even) go to out_idx / out_dist.  keep_padding: setKeepPadding.
int pqtfe_queryKNN_inflight(void* h, const float* q_dev, const float* q_dev2, uint32_t qn, uint32_t nvec, uint32_t bv, uint32_t bb, int reps, int keep_padding,
                            double* wall_ms, uint32_t* out_idx, float* out_dist) {
  try {
    Fe* f = static_cast<Fe*>(h);
    f->t.setBounds(bv, bb);
    f->t.setKeepPadding(keep_padding != 0);
    if (hipDeviceSynchronize() != hipSuccess) throw std::runtime_error("device");
    const auto t0 = std::chrono::steady_clock::now();
    int pending = -1;
    for (int r = 0; r < reps; ++r) {
      const int tk = f->t.queryKNNAsync((r & 1) ? q_dev2 : q_dev, qn, nvec);
      if (pending >= 0) f->t.queryKNNCollect(pending, f->idx, f->dist);
      pending = tk;
    }
    if (pending >= 0) f->t.queryKNNCollect(pending, f->idx, f->dist);
    if (wall_ms) wall_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (reps > 0 ? reps : 1);
    if (out_idx) memcpy(out_idx, f->idx.data(), f->idx.size() * 4);
    if (out_dist) memcpy(out_dist, f->dist.data(), f->dist.size() * 4);
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

void pqtfe_set_keep_padding(void* h, int on) { static_cast<Fe*>(h)->t.setKeepPadding(on != 0); }
void pqtfe_set_legacy_copy(void* h, int on) { static_cast<Fe*>(h)->t.setLegacyCopy(on != 0); }
// writes `value` into slot `col` of every row of the wrapper's result vectors (a caller that scribbles on them between calls: tests)
void pqtfe_scribble(void* h, uint32_t nvec, uint32_t col, uint32_t value) {
  Fe* f = static_cast<Fe*>(h);
  for (size_t i = col; i < f->idx.size(); i += nvec) f->idx[i] = value;
}

}  // extern "C"
