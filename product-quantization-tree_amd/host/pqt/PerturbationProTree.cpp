// pqt/PerturbationProTree.cpp -- the reference's class surface implemented over the C-ABI (see the header).
#include "PerturbationProTree.hh"

#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <fstream>
#include <limits>
#include <map>
#include <stdexcept>
#include <thread>
#include <condition_variable>
#include <functional>
#include <mutex>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace pqt {

// ---- host side of queryKNN's hand-over ----------------------------------------------------------------------------------------------
// A small persistent pool (the padded arrays of one 4096-query batch at _nVec = 4096 are 2 x 67 MB of host memory to write: one thread
// does 13 GB/s with plain stores, eight do 100 GB/s with streaming stores -- profiles/r04_hostfill.txt).
class HostPool {
 public:
  explicit HostPool(int n) : d_n(n), d_gen(0), d_left(0), d_stop(false) {
    for (int t = 1; t < n; ++t) d_th.emplace_back([this, t] { run(t); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(d_mu); d_stop = true; ++d_gen; }
    d_cv.notify_all();
    for (auto& t : d_th) t.join();
  }
  int size() const { return d_n; }
  // job(t, n) on every thread t of n (the caller is thread 0); returns when all are done
  void run_all(const std::function<void(int, int)>& job) {
    { std::lock_guard<std::mutex> lk(d_mu); d_job = &job; d_left = d_n - 1; ++d_gen; }
    d_cv.notify_all();
    job(0, d_n);
    std::unique_lock<std::mutex> lk(d_mu);
    d_done.wait(lk, [this] { return d_left == 0; });
    d_job = nullptr;
  }
 private:
  void run(int t) {
    unsigned long long seen = 0;
    for (;;) {
      const std::function<void(int, int)>* job;
      { std::unique_lock<std::mutex> lk(d_mu); d_cv.wait(lk, [&] { return d_gen != seen; }); seen = d_gen; if (d_stop) return; job = d_job; }
      if (job) (*job)(t, d_n);
      { std::lock_guard<std::mutex> lk(d_mu); if (--d_left == 0) d_done.notify_one(); }
    }
  }
  int d_n; std::vector<std::thread> d_th; std::mutex d_mu; std::condition_variable d_cv, d_done;
  const std::function<void(int, int)>* d_job = nullptr; unsigned long long d_gen; int d_left; bool d_stop;
};


ProQuantization::ProQuantization(uint _dim, uint _p) : d_dim(_dim), d_p(_p), d_vl(_p ? _dim / _p : 0), d_nClusters(0) {}
ProQuantization::~ProQuantization() {}

ProTree::ProTree(uint _dim, uint _p, uint _p2) : ProQuantization(_dim, _p), d_p2(_p2), d_nClusters2(0) {}

void ProTree::prepareDistSequence(uint _rows) {
  if (pqt_index_build_heuristic(handle(), _rows) != PQT_OK) throw std::runtime_error(pqt_last_error());
}

void ProTree::prepareDistSequence(int _maxCluster, int _groupParts) {
  if (_groupParts != (int)d_p) throw std::runtime_error("prepareDistSequence: groupParts must equal p");
  if (pqt_index_build_heuristic_cuda(handle(), (uint32_t)_maxCluster, 65536) != PQT_OK) throw std::runtime_error(pqt_last_error());
}

void ProTree::prepare2DDistSequence(int _maxCluster) {
  if (pqt_index_build_heuristic_2d(handle(), (uint32_t)_maxCluster) != PQT_OK) throw std::runtime_error(pqt_last_error());
}

PerturbationProTree::PerturbationProTree(uint _dim, uint _p, uint _p2)
    : ProTree(_dim, _p, _p2), d_idx(nullptr), d_multi(nullptr), d_issued(0), d_collected(0), d_lastSlot(0), d_keepPadding(false), d_padIdx(nullptr), d_padDist(nullptr), d_padQN(0), d_padNVec(0),
      d_legacyCopy(getenv("PQT_FRONTEND_LEGACY_COPY") != nullptr),
      d_packMin(getenv("PQT_FRONTEND_PACK_MIN_BYTES") ? (size_t)atoll(getenv("PQT_FRONTEND_PACK_MIN_BYTES")) : ((size_t)8 << 20)),
      d_poolThreads(getenv("PQT_FRONTEND_THREADS") ? std::max(1, atoi(getenv("PQT_FRONTEND_THREADS"))) : (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()))),
      d_pool(nullptr),
      d_lastTiming(), d_hashPrefix(nullptr),
      d_hashCounts(nullptr), d_hashSizeHeld(0), d_lineById(nullptr), d_lineByIdValid(false), d_device(0), d_w(2), d_lineParts(16), d_boundVectors(20000),
      d_boundBins(500), d_heurRows(0), d_N(0) {}

void PerturbationProTree::releaseSlot(KnnSlot& s) {
  if (s.d_resIdx) (void)hipFree(s.d_resIdx);
  if (s.d_resDist) (void)hipFree(s.d_resDist);
  if (s.d_resCnt) (void)hipFree(s.d_resCnt);
  if (s.h_stageIdx) (void)hipHostFree(s.h_stageIdx);
  if (s.h_stageDist) (void)hipHostFree(s.h_stageDist);
  if (s.h_offsets) (void)hipHostFree(s.h_offsets);
  if (s.d_offsets) (void)hipFree(s.d_offsets);
  if (s.d_packIdx) (void)hipFree(s.d_packIdx);
  if (s.d_packDist) (void)hipFree(s.d_packDist);
  if (s.evOff) (void)hipEventDestroy(s.evOff);
  if (s.evIdx) (void)hipEventDestroy(s.evIdx);
  if (s.evDist) (void)hipEventDestroy(s.evDist);
  if (s.stream) (void)hipStreamDestroy(s.stream);
  pqt_index* keep = s.h;
  s = KnnSlot();
  s.h = keep;  // (the view of slot 1 dies with its owner: pqt_index_destroy of the index)
}

void PerturbationProTree::releaseDeviceScratch() {
  for (KnnSlot& s : d_slots) releaseSlot(s);
  d_issued = d_collected = 0;
  d_padIdx = nullptr;
  if (d_hashPrefix) (void)hipFree(d_hashPrefix);
  if (d_hashCounts) (void)hipFree(d_hashCounts);
  if (d_lineById) (void)hipFree(d_lineById);
  d_hashPrefix = d_hashCounts = nullptr; d_hashSizeHeld = 0;
  d_lineById = nullptr; d_lineByIdValid = false;
}

PerturbationProTree::~PerturbationProTree() {
  delete d_pool;
  (void)hipSetDevice(d_device);
  releaseDeviceScratch();
  if (d_idx) pqt_index_destroy(d_idx);
  if (d_multi) pqt_multi_destroy(d_multi);
}

// result buffers of queryKNN live as long as the object and only grow (the reference allocates and frees all scratch per
// batch, PerturbationProTree.cu:8201-8238,8315-8321 -- the overhead SURVEY App. C lists first); staging = queryKNN's packed hand-over: pinned
// host staging and device buffers for the packed rows and their offsets
void PerturbationProTree::ensureSlot(KnnSlot& s, size_t _n, size_t _qn, bool staging) {
  if (!s.stream) {
    if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&s.evOff, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.evIdx, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.evDist, hipEventDisableTiming) != hipSuccess)
      throw std::runtime_error("stream / event creation failed");
  }
  if (_n > s.resCap) {
    if (s.d_resIdx) (void)hipFree(s.d_resIdx);
    if (s.d_resDist) (void)hipFree(s.d_resDist);
    s.d_resIdx = nullptr; s.d_resDist = nullptr; s.resCap = 0;
    if (hipMalloc((void**)&s.d_resIdx, _n * 4) != hipSuccess || hipMalloc((void**)&s.d_resDist, _n * 4) != hipSuccess)
      throw std::runtime_error("device allocation failed");
    s.resCap = _n;
  }
  if (!staging) return;
  if (_n > s.stageCap) {
    if (s.h_stageIdx) (void)hipHostFree(s.h_stageIdx);
    if (s.h_stageDist) (void)hipHostFree(s.h_stageDist);
    if (s.d_packIdx) (void)hipFree(s.d_packIdx);
    if (s.d_packDist) (void)hipFree(s.d_packDist);
    s.h_stageIdx = nullptr; s.h_stageDist = nullptr; s.d_packIdx = nullptr; s.d_packDist = nullptr; s.stageCap = 0;
    // (the packed rows of a batch that takes this path fill at most half of the padded size; the staging is sized for the worst case once)
    if (hipHostMalloc((void**)&s.h_stageIdx, _n * 4, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void**)&s.h_stageDist, _n * 4, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&s.d_packIdx, _n * 4) != hipSuccess || hipMalloc((void**)&s.d_packDist, _n * 4) != hipSuccess)
      throw std::runtime_error("staging allocation failed");
    s.stageCap = _n;
  }
  if (_qn + 1 > s.cntCap) {
    if (s.h_offsets) (void)hipHostFree(s.h_offsets);
    if (s.d_offsets) (void)hipFree(s.d_offsets);
    if (s.d_resCnt) (void)hipFree(s.d_resCnt);
    s.h_offsets = nullptr; s.d_offsets = nullptr; s.d_resCnt = nullptr; s.cntCap = 0;
    if (hipHostMalloc((void**)&s.h_offsets, (_qn + 1) * 4, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&s.d_offsets, (_qn + 1) * 4) != hipSuccess ||
        hipMalloc((void**)&s.d_resCnt, (_qn + 1) * 4) != hipSuccess)
      throw std::runtime_error("staging allocation failed");
    s.cntCap = _qn + 1;
  }
}

void PerturbationProTree::check(int rc, const char* what) {
  if (rc != PQT_OK) throw std::runtime_error(std::string(what) + ": " + pqt_last_error());
}

pqt_index* PerturbationProTree::handle() {
  if (d_multi) return pqt_multi_shard(d_multi, 0);  // tree-level calls (assign + encode, statistics): the first shard's copy
  if (!d_idx) throw std::runtime_error("no tree loaded (readTreeFromFile / loadTree / setTree first)");
  return d_idx;
}

void PerturbationProTree::singleDeviceOnly(const char* what) const {
  if (d_multi) throw std::runtime_error(std::string(what) + ": not available with several devices (setDevices); use one device");
}

void PerturbationProTree::prepareDistSequence(uint _rows) {
  if (d_multi) { if (pqt_multi_build_heuristic(d_multi, _rows) != PQT_OK) throw std::runtime_error(pqt_multi_last_error()); }
  else ProTree::prepareDistSequence(_rows);
  d_heurRows = _rows;
}

// the CUDA library's order (ProTree.cu:128-207) as the table: with several devices EVERY shard gets it (the traversal is sharded by
// query slice, so a table on shard 0 only would order one slice's bins differently from the others').  The table holds at most 65536
// rows; d_heurRows records it so that a later queryKNN does not silently replace it by the cpu_version table.
void PerturbationProTree::prepareDistSequence(int _maxCluster, int _groupParts) {
  if (_groupParts != (int)d_p) throw std::runtime_error("prepareDistSequence: groupParts must equal p");
  if (d_multi) { if (pqt_multi_build_heuristic_cuda(d_multi, (uint32_t)_maxCluster, 65536) != PQT_OK) throw std::runtime_error(pqt_multi_last_error()); }
  else ProTree::prepareDistSequence(_maxCluster, _groupParts);
  d_heurRows = 65536;
}

// the 2-D anisotropic sequences of the 1B path (ProTree.cu:50-126, test/test1B.cpp:941): per-query rows; at most 65536 cells per order
void PerturbationProTree::prepare2DDistSequence(int _maxCluster) {
  if (d_multi) { if (pqt_multi_build_heuristic_2d(d_multi, (uint32_t)_maxCluster) != PQT_OK) throw std::runtime_error(pqt_multi_last_error()); }
  else ProTree::prepare2DDistSequence(_maxCluster);
  d_heurRows = (uint32_t)std::min<uint64_t>(65536, (uint64_t)_maxCluster * (uint64_t)_maxCluster);  // what the library holds (pqt_index_build_heuristic_2d)
}

void PerturbationProTree::uploadLines(size_t _N) {
  if (d_multi) { if (pqt_multi_set_lines_host(d_multi, reinterpret_cast<const uint32_t*>(h_lines.data()), _N) != PQT_OK) throw std::runtime_error(pqt_multi_last_error()); }
  else check(pqt_index_set_lines_host(handle(), reinterpret_cast<const uint32_t*>(h_lines.data()), _N, 0), "pqt_index_set_lines_host");
  d_lineByIdValid = false;
}

void PerturbationProTree::setTree(uint _c1, uint _c2, const float* _cb1, const float* _cb2) {
  d_nClusters = _c1; d_nClusters2 = _c2;
  h_codeBook.assign(_cb1, _cb1 + (size_t)_c1 * d_dim);
  h_codeBook2.assign(_cb2, _cb2 + (size_t)_c1 * _c2 * d_dim);
  if (d_idx) { pqt_index_destroy(d_idx); d_idx = nullptr; }   // (the view of slot 1 is destroyed with it)
  if (d_multi) { pqt_multi_destroy(d_multi); d_multi = nullptr; }
  for (KnnSlot& sl : d_slots) { sl.h = nullptr; sl.busy = false; }
  d_issued = d_collected = 0;
  pqt_params prm = {d_dim, d_p, _c1, _c2, std::min(d_w, _c1), d_lineParts};
  if (d_devices.size() > 1) {
    if (pqt_multi_create(&prm, (int)d_devices.size(), d_devices.data(), &d_multi) != PQT_OK ||
        pqt_multi_set_codebooks(d_multi, h_codeBook.data(), h_codeBook2.data()) != PQT_OK)
      throw std::runtime_error(std::string("pqt_multi: ") + pqt_multi_last_error());
  } else {
    check(pqt_index_create(&prm, d_device, &d_idx), "pqt_index_create");
    check(pqt_index_set_codebooks(d_idx, h_codeBook.data(), h_codeBook2.data()), "pqt_index_set_codebooks");
  }
  d_heurRows = 0;
}

namespace {
// device buffer with RAII
template <class T> struct DevBuf {
  T* p = nullptr;
  explicit DevBuf(size_t n) { if (hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) throw std::runtime_error("device allocation failed"); }
  ~DevBuf() { if (p) (void)hipFree(p); }
  DevBuf(const DevBuf&) = delete;
};
void h2d(void* d, const void* h, size_t b) { if (b && hipMemcpy(d, h, b, hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("H2D copy failed"); }
void d2h(void* h, const void* d, size_t b) { if (b && hipMemcpy(h, d, b, hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("D2H copy failed"); }

// Lloyd iterations with centroid splitting on `dim` dims of the rows selected by `rows` (all rows if empty):
// restates productquantizer::generate for ONE part / vectorquantizer::generate (same schedule, see callers).
struct SplitKMeans {
  int device; const float* xDev; const float* xHost; uint32_t ld; uint32_t dim; uint32_t off;  // column window [off, off+dim)
  std::vector<uint32_t> rows; const uint32_t* rowsDev; size_t n;
  // E step on the device: assignment + distance of every selected row to the first `step` centroids
  void estep(const std::vector<float>& cen, uint32_t step, std::vector<uint32_t>& asg, std::vector<float>& dist,
             DevBuf<float>& dCen, DevBuf<uint32_t>& dAsg, DevBuf<float>& dDist) const {
    if (!n) return;
    h2d(dCen.p, cen.data(), (size_t)step * dim * 4);
    if (pqt_kmeans_assign(device, xDev + off, n, dim, ld, rowsDev, dCen.p, step, dim, dAsg.p, dDist.p, nullptr) != PQT_OK)
      throw std::runtime_error(std::string("pqt_kmeans_assign: ") + pqt_last_error());
    d2h(asg.data(), dAsg.p, n * 4);
    d2h(dist.data(), dDist.p, n * 4);
  }
  const float* row(size_t i) const { return xHost + (size_t)(rows.empty() ? i : rows[i]) * ld + off; }
};
}  // namespace

void PerturbationProTree::createTree(uint _k, uint _k2, const float* _Ain, uint _N, MemSpace _space) {
  if (hipSetDevice(d_device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
  const uint D = d_dim, P = d_p, S = D / P, C1 = _k, C2 = _k2;
  if (!C1 || (C1 & 1) || C1 > 256 || !C2 || (C2 > 1 && (C2 & 1)) || C2 > 256) throw std::runtime_error("createTree: cluster counts must be even (or C2 == 1) and <= 256");
  // the E step reads the vectors on the device, the sequential M step on the host: one copy in the missing direction
  DevBuf<float> dX(_space == HOST_PTR ? (size_t)_N * D : 0);
  std::vector<float> hostCopy;
  const float* _A = _Ain;
  if (_space == HOST_PTR) h2d(dX.p, _Ain, (size_t)_N * D * 4);
  else { hostCopy.resize((size_t)_N * D); d2h(hostCopy.data(), _Ain, hostCopy.size() * 4); _A = hostCopy.data(); }
  const float* const xDev = _space == HOST_PTR ? dX.p : _Ain;
  DevBuf<float> dCen((size_t)256 * std::max(S, 1u));
  DevBuf<uint32_t> dAsg(_N); DevBuf<float> dDist(_N); DevBuf<uint32_t> dRows(_N);
  std::vector<float> cb1((size_t)C1 * D, 0.f), cb2((size_t)P * C1 * C2 * S, 0.f);
  std::vector<uint32_t> asg(_N); std::vector<float> dist(_N);
  std::vector<std::vector<uint32_t> > assign1(P, std::vector<uint32_t>(_N, 0));

  // ---- level 1: productquantizer::generate (productquantizer.hpp:131-158).  The reference runs all parts inside one
  // loop with a COMMON loss (sum over parts); the parts only interact through that stopping test, so they are iterated
  // together here as well.
  {
    std::vector<std::vector<float> > cen(P, std::vector<float>((size_t)C1 * S, 0.f));
    std::vector<std::vector<uint32_t> > a(P, std::vector<uint32_t>(_N, 0));
    std::vector<std::vector<float> > dd(P, std::vector<float>(_N, 0.f));
    // iterator::center(): mean of all vectors (iterator/iterator.hpp:41-53), sequential sums
    for (uint p = 0; p < P; ++p) {
      std::vector<float> avg(S, 0.f);
      for (uint i = 0; i < _N; ++i) for (uint d = 0; d < S; ++d) avg[d] += _A[(size_t)i * D + p * S + d];
      for (uint d = 0; d < S; ++d) cen[p][d] = avg[d] / (float)_N;
    }
    uint step = 1;
    float cur = 0.f, last = 0.f;
    do {
      uint run = 1000;
      for (uint p = 0; p < P; ++p)  // augmentCentroids (:97-103)
        for (uint i = 0; i < step; ++i) for (uint d = 0; d < S; ++d) {
          cen[p][(size_t)(i + step) * S + d] = cen[p][(size_t)i * S + d] + 0.001f;
          cen[p][(size_t)i * S + d] = cen[p][(size_t)i * S + d] - 0.001f;
        }
      step *= 2;
      do {
        last = cur;
        for (uint p = 0; p < P; ++p) {
          SplitKMeans km{d_device, xDev, _A, D, S, p * S, {}, nullptr, _N};
          km.estep(cen[p], step, a[p], dd[p], dCen, dAsg, dDist);  // getAssignment (:40-66)
          // updateCentroids (:74-91): zero everything, sequential sums in vector order, divide non-empty
          std::fill(cen[p].begin(), cen[p].end(), 0.f);
          std::vector<float> cnt(C1, 0.f);
          for (uint i = 0; i < _N; ++i) {
            const uint c = a[p][i];
            for (uint d = 0; d < S; ++d) cen[p][(size_t)c * S + d] += _A[(size_t)i * D + p * S + d];
            cnt[c] += 1.f;
          }
          for (uint c = 0; c < C1; ++c) if (cnt[c] != 0) for (uint d = 0; d < S; ++d) cen[p][(size_t)c * S + d] /= cnt[c];
        }
        float sum = 0.f;  // loss (:113-124) over _distances[n*P + p] in index order
        for (uint i = 0; i < _N; ++i) for (uint p = 0; p < P; ++p) sum += dd[p][i];
        cur = sum;
        run--;
      } while ((std::fabs(last - cur) > 0.005f) && (run > 0));
    } while (step < C1);
    for (uint p = 0; p < P; ++p) {
      for (uint c = 0; c < C1; ++c) for (uint d = 0; d < S; ++d) cb1[(size_t)c * D + p * S + d] = cen[p][(size_t)c * S + d];
      // grouping uses the mapping of the LAST E step (treequantizer.hpp:140-147 reads _PQ->_mapping)
      assign1[p] = a[p];
    }
  }
  // ---- level 2: one vectorquantizer per (part, cell) on the segments grouped into that cell (treequantizer.hpp:163-172)
  for (uint p = 0; p < P; ++p)
    for (uint c = 0; c < C1; ++c) {
      SplitKMeans km{d_device, xDev, _A, D, S, p * S, {}, nullptr, 0};
      for (uint i = 0; i < _N; ++i) if (assign1[p][i] == c) km.rows.push_back(i);
      km.n = km.rows.size();
      h2d(dRows.p, km.rows.data(), km.n * 4);
      km.rowsDev = dRows.p;
      std::vector<float> cen((size_t)C2 * S, 0.f);
      std::vector<uint32_t> a(km.n, 0); std::vector<float> dd(km.n, 0.f);
      {
        std::vector<float> avg(S, 0.f);
        for (size_t i = 0; i < km.n; ++i) for (uint d = 0; d < S; ++d) avg[d] += km.row(i)[d];
        for (uint d = 0; d < S; ++d) cen[d] = avg[d] / (float)km.n;  // empty cell: NaN, zeroed by the first M step
      }
      auto mstep = [&](uint) {
        std::fill(cen.begin(), cen.end(), 0.f);
        std::vector<float> cnt(C2, 0.f);
        for (size_t i = 0; i < km.n; ++i) { const uint cc = a[i]; for (uint d = 0; d < S; ++d) cen[(size_t)cc * S + d] += km.row(i)[d]; cnt[cc] += 1.f; }
        for (uint cc = 0; cc < C2; ++cc) if (cnt[cc] != 0) for (uint d = 0; d < S; ++d) cen[(size_t)cc * S + d] /= cnt[cc];
      };
      uint step = 1;
      float cur = 0.f, last = 0.f;
      if (C2 == 1) { km.estep(cen, 1, a, dd, dCen, dAsg, dDist); mstep(1); }
      else do {
        for (uint i = 0; i < step; ++i) for (uint d = 0; d < S; ++d) {
          cen[(size_t)(i + step) * S + d] = cen[(size_t)i * S + d] + 0.001f;
          cen[(size_t)i * S + d] = cen[(size_t)i * S + d] - 0.001f;
        }
        step *= 2;
        uint guard = 100000;  // the reference has no cap here (vectorquantizer.hpp:137-143)
        do {
          last = cur;
          km.estep(cen, step, a, dd, dCen, dAsg, dDist);
          mstep(step);
          float sum = 0.f;
          for (size_t i = 0; i < km.n; ++i) sum += dd[i];
          cur = sum;
        } while ((std::fabs(last - cur) > 0.005f) && --guard);
      } while (step < C2);
      std::copy(cen.begin(), cen.end(), cb2.begin() + ((size_t)p * C1 + c) * C2 * S);
    }
  setTree(C1, C2, cb1.data(), cb2.data());
}

void PerturbationProTree::readTreeFromFile(const std::string& _name) {
  std::ifstream f(_name.c_str(), std::ifstream::in | std::ifstream::binary);
  if (!f.good()) throw std::runtime_error("cannot open file " + _name);
  uint dim, p, p2, c1, c2, ndbs;
  f >> dim >> p >> p2 >> c1 >> c2 >> ndbs;
  f.ignore(1);
  if (!f.good() || ndbs < 1) throw std::runtime_error("bad .ppqt header in " + _name);
  if (dim != d_dim || p != d_p) throw std::runtime_error("tree file does not match the constructor's dim/p");
  std::vector<float> cb1((size_t)ndbs * c1 * dim), cb2((size_t)ndbs * c1 * c2 * dim);
  f.read((char*)cb1.data(), cb1.size() * 4);
  f.read((char*)cb2.data(), cb2.size() * 4);
  if (!f.good()) throw std::runtime_error("short read in " + _name);
  d_p2 = p2;
  setTree(c1, c2, cb1.data(), cb2.data());  // perturbation 0 only (nDBs is hard-wired to 1, PerturbationProTree.cu:33)
}

void PerturbationProTree::writeTreeToFile(const std::string& _name) {
  handle();
  std::ofstream f(_name.c_str(), std::ofstream::out | std::ofstream::binary);
  if (!f.good()) throw std::runtime_error("cannot open file " + _name);
  f << d_dim << std::endl << d_p << std::endl << d_p2 << std::endl << d_nClusters << std::endl << d_nClusters2 << std::endl
    << 1 << std::endl;
  f.write((const char*)h_codeBook.data(), h_codeBook.size() * 4);
  f.write((const char*)h_codeBook2.data(), h_codeBook2.size() * 4);
}

void PerturbationProTree::loadTree(const std::string& _name) {
  std::ifstream f(_name.c_str(), std::ios_base::in | std::ios_base::binary);
  if (!f.good()) throw std::runtime_error("read error: cannot open file" + _name);
  uint hdr[5];
  f.read((char*)hdr, sizeof(hdr));
  if (hdr[0] != d_dim) throw std::runtime_error("D missmatch");
  if (hdr[3] != d_p) throw std::runtime_error("P missmatch");
  const uint c1 = hdr[1], c2 = hdr[2];
  std::vector<float> cb1((size_t)c1 * d_dim), cb2((size_t)c1 * c2 * d_dim);
  f.read((char*)cb1.data(), cb1.size() * 4);
  f.read((char*)cb2.data(), cb2.size() * 4);
  if (!f.good()) throw std::runtime_error("short read in " + _name);
  setTree(c1, c2, cb1.data(), cb2.data());
}

void PerturbationProTree::saveTree(const std::string& _name) {
  handle();
  std::ofstream f(_name.c_str(), std::ios_base::out | std::ios_base::binary);
  if (!f.good()) throw std::runtime_error("write error: cannot open file " + _name);
  const uint hdr[5] = {d_dim, d_nClusters, d_nClusters2, d_p, d_w};
  f.write((const char*)hdr, sizeof(hdr));
  f.write((const char*)h_codeBook.data(), h_codeBook.size() * 4);
  f.write((const char*)h_codeBook2.data(), h_codeBook2.size() * 4);
}

void PerturbationProTree::setBins(size_t _nbins, const uint* _ids, const uint* _sizes, const uint* _members) {
  size_t n = 0;
  for (size_t b = 0; b < _nbins; ++b) n += _sizes[b];
  h_binIds.assign(_ids, _ids + _nbins); h_binSizes.assign(_sizes, _sizes + _nbins); h_members.assign(_members, _members + n);
  d_N = n;
  d_hashSizeHeld = 0;  // the dense hashed getters are rebuilt on demand
  if (d_multi) { if (pqt_multi_set_bins(d_multi, _nbins, _ids, _sizes, _members, n) != PQT_OK) throw std::runtime_error(pqt_multi_last_error()); }
  else check(pqt_index_set_bins(handle(), _nbins, _ids, _sizes, _members), "pqt_index_set_bins");
}

void PerturbationProTree::setDB(uint _N, const uint* _prefix, const uint* _counts, const uint* _dbIdx, uint _hashSize) {
  singleDeviceOnly("setDB (hashed dump family)");
  d_N = _N;
  // the exact bin ids are not recoverable from the hashed form: the exact-bin state (and the dense getters cached from it) go
  h_binIds.clear(); h_binSizes.clear(); h_members.clear();
  d_hashSizeHeld = 0;
  check(pqt_index_set_db_hashed(handle(), _N, _prefix, _counts, _dbIdx, _hashSize), "pqt_index_set_db_hashed");
}

void PerturbationProTree::prepareEmptyLambda(uint _N, uint _lParts) {
  if (d_idx && _lParts != d_lineParts) throw std::runtime_error("line parts must be fixed before the tree is read");
  d_lineParts = _lParts;
  h_lines.assign((size_t)_N * _lParts, lineDescr{0, 0, 0});
}

void PerturbationProTree::setLines(const lineDescr* _lines, size_t _N) {
  h_lines.assign(_lines, _lines + _N * d_lineParts);
  uploadLines(_N);
}

void PerturbationProTree::loadBins(const std::string& _name) {
  std::ifstream f(_name.c_str(), std::ios_base::in | std::ios_base::binary);
  if (!f.good()) throw std::runtime_error("read error: cannot open file " + _name);
  uint nb = 0;
  f.read((char*)&nb, 4);
  std::vector<uint> ids(nb), sizes(nb), members;
  for (uint i = 0; i < nb; ++i) {
    f.read((char*)&ids[i], 4);
    f.read((char*)&sizes[i], 4);
    const size_t o = members.size();
    members.resize(o + sizes[i]);
    f.read((char*)(members.data() + o), (size_t)sizes[i] * 4);
  }
  uint len = 0, lp = 0;
  f.read((char*)&len, 4);
  f.read((char*)&lp, 4);
  if (!f.good()) throw std::runtime_error("short read in " + _name);
  if (lp != d_lineParts) throw std::runtime_error("LP missmatch");
  if (len != members.size()) throw std::runtime_error("#vectors missmatch ");
  std::vector<lineDescr> lines((size_t)len * lp);
  f.read((char*)lines.data(), lines.size() * 4);
  if (!f.good()) throw std::runtime_error("short read in " + _name);
  setBins(nb, ids.data(), sizes.data(), members.data());
  setLines(lines.data(), len);
}

void PerturbationProTree::saveBins(const std::string& _name) {
  std::ofstream f(_name.c_str(), std::ios_base::out | std::ios_base::binary);
  if (!f.good()) throw std::runtime_error("write error: cannot open file " + _name);
  // std::map order of the reference = ascending bin id
  std::vector<size_t> order(h_binIds.size()), start(h_binIds.size());
  size_t o = 0;
  for (size_t b = 0; b < order.size(); ++b) { order[b] = b; start[b] = o; o += h_binSizes[b]; }
  std::sort(order.begin(), order.end(), [&](size_t l, size_t r) { return h_binIds[l] < h_binIds[r]; });
  const uint nb = (uint)order.size();
  f.write((const char*)&nb, 4);
  for (size_t b : order) {
    f.write((const char*)&h_binIds[b], 4);
    f.write((const char*)&h_binSizes[b], 4);
    f.write((const char*)(h_members.data() + start[b]), (size_t)h_binSizes[b] * 4);
  }
  const uint len = (uint)d_N, lp = d_lineParts;
  f.write((const char*)&len, 4);
  f.write((const char*)&lp, 4);
  f.write((const char*)h_lines.data(), h_lines.size() * 4);
}

void PerturbationProTree::buildKBestDBChunk(const float* _A, uint _N, uint _idOffset, MemSpace _space) {
  pqt_index* h = handle();
  if (hipSetDevice(d_device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
  if (h_binOfVec.size() < (size_t)_idOffset + _N) h_binOfVec.resize((size_t)_idOffset + _N, 0);
  if (h_lines.size() < ((size_t)_idOffset + _N) * d_lineParts) h_lines.resize(((size_t)_idOffset + _N) * d_lineParts, lineDescr{0, 0, 0});
  if (!_N) return;
  DevBuf<float> dA(_space == HOST_PTR ? (size_t)_N * d_dim : 0);
  DevBuf<uint32_t> dBin(_N), dCodes((size_t)_N * d_lineParts);
  if (_space == HOST_PTR) h2d(dA.p, _A, (size_t)_N * d_dim * 4);
  check(pqt_build_assign_encode(h, _space == HOST_PTR ? dA.p : _A, _N, dBin.p, dCodes.p, nullptr), "pqt_build_assign_encode");
  d2h(h_binOfVec.data() + _idOffset, dBin.p, (size_t)_N * 4);
  d2h(h_lines.data() + (size_t)_idOffset * d_lineParts, dCodes.p, (size_t)_N * d_lineParts * 4);
}

void PerturbationProTree::finishDB() {
  handle();
  const size_t n = h_binOfVec.size();
  // bins in ascending id order, members in insertion (= id) order: exactly what the reference's std::map holds after
  // insert() over the whole dataset (treequantizer.hpp:212-217), whatever the chunking was
  std::map<uint, std::vector<uint> > bins;
  for (size_t i = 0; i < n; ++i) bins[h_binOfVec[i]].push_back((uint)i);
  std::vector<uint> ids, sizes, members;
  members.reserve(n);
  for (auto& kv : bins) { ids.push_back(kv.first); sizes.push_back((uint)kv.second.size()); members.insert(members.end(), kv.second.begin(), kv.second.end()); }
  setBins(ids.size(), ids.data(), sizes.data(), members.data());
  uploadLines(n);
  h_binOfVec.clear();
  h_binOfVec.shrink_to_fit();
}

void PerturbationProTree::buildKBestDB(const float* _A, uint _N, MemSpace _space) {
  h_binOfVec.clear();
  h_lines.clear();
  buildKBestDBChunk(_A, _N, 0, _space);
  finishDB();
}

// ---- the CUDA library's dump family --------------------------------------------------------------------------------
void PerturbationProTree::exportHashed(uint _hashSize, std::vector<uint>& _prefix, std::vector<uint>& _counts, std::vector<uint>& _dbIdx) const {
  if (!_hashSize) throw std::runtime_error("hash size must be > 0");
  _prefix.assign(_hashSize, 0); _counts.assign(_hashSize, 0); _dbIdx.clear(); _dbIdx.reserve(d_N);
  // slot = bin id % hashSize (PerturbationProTree.cu:3479-3484); bins sharing a slot are concatenated in ascending id order
  std::vector<size_t> start(h_binIds.size()), order(h_binIds.size());
  size_t o = 0;
  for (size_t b = 0; b < h_binIds.size(); ++b) { start[b] = o; o += h_binSizes[b]; order[b] = b; }
  std::sort(order.begin(), order.end(), [&](size_t l, size_t r) {
    const uint sl = h_binIds[l] % _hashSize, sr = h_binIds[r] % _hashSize;
    return sl != sr ? sl < sr : h_binIds[l] < h_binIds[r];
  });
  for (size_t b : order) {
    const uint slot = h_binIds[b] % _hashSize;
    if (_counts[slot] == 0) _prefix[slot] = (uint)_dbIdx.size();
    _counts[slot] += h_binSizes[b];
    _dbIdx.insert(_dbIdx.end(), h_members.begin() + start[b], h_members.begin() + start[b] + h_binSizes[b]);
  }
}

namespace {
void writeRaw(const std::string& name, const void* p, size_t bytes) {
  std::ofstream f(name.c_str(), std::ofstream::out | std::ofstream::binary);
  if (!f.good()) throw std::runtime_error("cannot open file " + name);
  f.write((const char*)p, bytes);
  if (!f.good()) throw std::runtime_error("write error in " + name);
}
void readRaw(const std::string& name, void* p, size_t bytes) {
  std::ifstream f(name.c_str(), std::ifstream::in | std::ifstream::binary);
  if (!f.good()) throw std::runtime_error("cannot open file " + name);
  f.read((char*)p, bytes);
  if (!f.good()) throw std::runtime_error("short read in " + name);
}
}  // namespace

void PerturbationProTree::saveHashedDB(const std::string& _pre, uint _hashSize) {
  if (h_binIds.empty() || h_lines.size() != d_N * d_lineParts) throw std::runtime_error("saveHashedDB: no exact bins / line codes held");
  std::vector<uint> prefix, counts, dbIdx;
  exportHashed(_hashSize, prefix, counts, dbIdx);
  writeRaw(_pre + "_" + std::to_string(d_lineParts) + ".lines", h_lines.data(), h_lines.size() * 4);
  writeRaw(_pre + ".prefix", prefix.data(), prefix.size() * 4);
  writeRaw(_pre + ".count", counts.data(), counts.size() * 4);
  writeRaw(_pre + ".dbIdx", dbIdx.data(), dbIdx.size() * 4);
}

void PerturbationProTree::loadHashedDB(const std::string& _pre, uint _N, uint _hashSize) {
  std::vector<uint> prefix(_hashSize), counts(_hashSize), dbIdx(_N);
  readRaw(_pre + ".prefix", prefix.data(), prefix.size() * 4);
  readRaw(_pre + ".count", counts.data(), counts.size() * 4);
  readRaw(_pre + ".dbIdx", dbIdx.data(), dbIdx.size() * 4);
  std::vector<lineDescr> lines((size_t)_N * d_lineParts);
  readRaw(_pre + "_" + std::to_string(d_lineParts) + ".lines", lines.data(), lines.size() * 4);
  h_binIds.clear(); h_binSizes.clear(); h_members.clear();  // the exact bin ids are not recoverable from the hashed form
  setDB(_N, prefix.data(), counts.data(), dbIdx.data(), _hashSize);
  setLines(lines.data(), _N);
}

const uint* PerturbationProTree::getDBIdx() {
  singleDeviceOnly("getDBIdx");
  const uint32_t* ids = nullptr;
  check(pqt_index_device_arrays(handle(), &ids, nullptr, nullptr), "pqt_index_device_arrays");
  return ids;
}

const lineDescr* PerturbationProTree::getLine() {
  handle();
  if (h_lines.empty()) throw std::runtime_error("getLine: no line codes held");
  if (!d_lineByIdValid) {
    if (hipSetDevice(d_device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
    if (d_lineById) (void)hipFree(d_lineById);
    d_lineById = nullptr;
    if (hipMalloc((void**)&d_lineById, h_lines.size() * sizeof(lineDescr)) != hipSuccess) throw std::runtime_error("device allocation failed");
    h2d(d_lineById, h_lines.data(), h_lines.size() * sizeof(lineDescr));
    d_lineByIdValid = true;
  }
  return d_lineById;
}

const lineDescr* PerturbationProTree::getLineBinOrder() {
  singleDeviceOnly("getLineBinOrder");
  const uint32_t* codes = nullptr;
  check(pqt_index_device_arrays(handle(), nullptr, &codes, nullptr), "pqt_index_device_arrays");
  return reinterpret_cast<const lineDescr*>(codes);
}

const uint* PerturbationProTree::getBinPrefix(uint _hashSize) {
  if (h_binIds.empty()) throw std::runtime_error("getBinPrefix/getBinCounts: no exact bins held (setBins / loadBins / buildKBestDB first; a hashed setDB cannot be re-hashed)");
  if (d_hashSizeHeld != _hashSize) {
    if (hipSetDevice(d_device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
    std::vector<uint> prefix, counts, dbIdx;
    exportHashed(_hashSize, prefix, counts, dbIdx);
    if (d_hashPrefix) (void)hipFree(d_hashPrefix);
    if (d_hashCounts) (void)hipFree(d_hashCounts);
    d_hashPrefix = d_hashCounts = nullptr; d_hashSizeHeld = 0;
    if (hipMalloc((void**)&d_hashPrefix, (size_t)_hashSize * 4) != hipSuccess || hipMalloc((void**)&d_hashCounts, (size_t)_hashSize * 4) != hipSuccess)
      throw std::runtime_error("device allocation failed");
    h2d(d_hashPrefix, prefix.data(), (size_t)_hashSize * 4);
    h2d(d_hashCounts, counts.data(), (size_t)_hashSize * 4);
    d_hashSizeHeld = _hashSize;
  }
  return d_hashPrefix;
}

const uint* PerturbationProTree::getBinCounts(uint _hashSize) {
  (void)getBinPrefix(_hashSize);
  return d_hashCounts;
}

void PerturbationProTree::ensureHeuristic(uint rows) {
  if (rows > d_heurRows) prepareDistSequence(rows);
}

namespace {
// n 32-bit words of value v at p with streaming (non-temporal) 16-byte stores: the padding is written once and not read back here
inline void fillStream32(void* p, size_t n, uint32_t v) {
  uint32_t* w = static_cast<uint32_t*>(p);
  size_t i = 0;
#if defined(__SSE2__)
  while (i < n && (reinterpret_cast<uintptr_t>(w + i) & 15)) w[i++] = v;
  const __m128i x = _mm_set1_epi32((int)v);
  for (; i + 4 <= n; i += 4) _mm_stream_si128(reinterpret_cast<__m128i*>(w + i), x);
#endif
  for (; i < n; ++i) w[i] = v;  // (no SSE2: plain stores)
}
inline void fillFence() {
#if defined(__SSE2__)
  _mm_sfence();
#endif
}
double msSince(const std::chrono::steady_clock::time_point& t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
}  // namespace

// Same contract as the reference (PerturbationProTree.cu:8179-8183, 8278-8281): the two vectors are resized to _QN * _nVec and every
// row is padded (here: id 0xffffffff, distance +inf).  Round 3 copied the whole padded arrays with two synchronous copies -- 134 MB per
// 4096-query batch at _nVec = 4096, ~82 % of it padding, 2.4 ms at the PCIe rate against 0.4 ms of kernels.  For large sparse results
// the filled prefixes are packed on the device (pqt_compact_results), only they cross PCIe (into pinned staging owned by the object)
// while a few host threads write the padding -- only where the previous hand-over into the same vectors left entries (setKeepPadding) --
// and the same threads then scatter the packed rows.  Small or dense results (every row full, e.g. _nVec = 100) are copied straight
// into the caller's vectors.  queryKNN = queryKNNAsync + queryKNNCollect.
void PerturbationProTree::queryKNN(std::vector<uint>& _resIdx, std::vector<float>& _resDist, const float* _Q, uint _QN, uint _nVec) {
  queryKNNCollect(queryKNNAsync(_Q, _QN, _nVec), _resIdx, _resDist);
}

int PerturbationProTree::queryKNNAsync(const float* _Q, uint _QN, uint _nVec) {
  if (d_issued - d_collected >= 2) throw std::runtime_error("queryKNNAsync: two batches are in flight already (collect one first)");
  const auto t0 = std::chrono::steady_clock::now();
  pqt_index* h = handle();
  ensureHeuristic(d_boundBins);
  // nothing in flight: slot 0 (a caller of the synchronous queryKNN never touches the view); one batch in flight: the other slot
  const int si = (d_issued == d_collected) ? 0 : 1 - d_lastSlot;
  KnnSlot& s = d_slots[si];
  if (s.busy) throw std::runtime_error("queryKNNAsync: slot still holds an uncollected batch");
  const size_t n = (size_t)_QN * _nVec;
  // The slot becomes a ticket only once everything has been enqueued: a call that throws on the way (bad bounds, out of memory, a view that
  // cannot be created) leaves the object exactly as it was -- no busy slot, no ticket counted -- so the next call can simply be made,
  // like after a failed call of the reference's queryKNN.
  s.QN = _QN; s.nVec = _nVec; s.compact = false; s.issueMs = 0;
  if (_QN && _nVec) {
    bool enqueued = false;
    try {
      if (hipSetDevice(d_device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
      // the packed hand-over pays for itself on large results only (two extra kernels, one extra round trip, thread wake-ups)
      s.compact = !d_legacyCopy && n * 8 >= d_packMin && n <= 0xffffffffull;
      ensureSlot(s, n, _QN, s.compact);
      uint* cnt = s.compact ? s.d_resCnt : nullptr;
      if (d_multi) {
        // several devices: slot si = lane si of the multi handle (lane 1: a view of every shard with its own streams and exchange buffers,
        // pqt_multi_query_lane); the batch is fanned out from the slot's stream and nothing is waited for here -- two batches in flight
        // like on one device (round 6; before, the batch ran to completion at issue time)
        s.h = h;
        if (pqt_multi_query_lane(d_multi, si, _Q, _QN, d_boundVectors, d_boundBins, _nVec, s.d_resIdx, s.d_resDist, cnt, s.stream, 0) != PQT_OK)
          throw std::runtime_error(std::string("queryKNN: ") + pqt_multi_last_error());
      } else {
        if (si == 0) s.h = h;
        else if (!s.h) check(pqt_index_create_view(h, &s.h), "pqt_index_create_view");
        check(pqt_query(s.h, _Q, _QN, d_boundVectors, d_boundBins, _nVec, s.d_resIdx, s.d_resDist, cnt, s.stream, 0), "queryKNN");
      }
      enqueued = true;
      if (s.compact) {
        // offsets + packed rows + the copy of the offsets behind the query on the slot's stream; nothing is waited for here
        check(pqt_compact_results(s.h, _QN, _nVec, s.d_resIdx, s.d_resDist, s.d_resCnt, s.d_offsets, s.d_packIdx, s.d_packDist, s.stream, 0), "pqt_compact_results");
        if (hipMemcpyAsync(s.h_offsets, s.d_offsets, ((size_t)_QN + 1) * 4, hipMemcpyDeviceToHost, s.stream) != hipSuccess || hipEventRecord(s.evOff, s.stream) != hipSuccess)
          throw std::runtime_error("D2H copy failed");
      }
    } catch (...) {
      // whatever part of the batch reached the slot's stream is drained before the slot is handed out again (its buffers are reused)
      if (enqueued && s.stream) (void)hipStreamSynchronize(s.stream);
      s.compact = false; s.QN = s.nVec = 0;
      throw;
    }
  }
  s.busy = true;
  ++d_issued;
  d_lastSlot = si;
  s.issueMs = msSince(t0);
  return si;
}

void PerturbationProTree::queryKNNCollect(int _ticket, std::vector<uint>& _resIdx, std::vector<float>& _resDist) {
  if (_ticket < 0 || _ticket > 1 || !d_slots[_ticket].busy) throw std::runtime_error("queryKNNCollect: no such batch in flight");
  if (d_issued - d_collected == 2 && _ticket == d_lastSlot) throw std::runtime_error("queryKNNCollect: batches are collected in the order they were issued");
  const auto tAll = std::chrono::steady_clock::now();
  KnnSlot& s = d_slots[_ticket];
  const uint _QN = s.QN, _nVec = s.nVec;
  const size_t n = (size_t)_QN * _nVec;
  s.busy = false;
  ++d_collected;
  double hostMs = 0;
  {
    const auto t = std::chrono::steady_clock::now();
    if (_resIdx.size() != n) { _resIdx.resize(n); d_padIdx = nullptr; }   // (value-initialises only what is new: a caller that reuses its vectors pays once)
    if (_resDist.size() != n) { _resDist.resize(n); d_padIdx = nullptr; }
    hostMs += msSince(t);
  }
  d_lastTiming = CallTiming();
  if (!_QN || !_nVec) return;
  if (hipSetDevice(d_device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
  auto t = std::chrono::steady_clock::now();
  size_t total = n;
  if (s.compact) {
    if (hipEventSynchronize(s.evOff) != hipSuccess) throw std::runtime_error("queryKNN: the batch failed on the device");
    d_lastTiming.kernels_ms = s.issueMs + msSince(t);  // issue + what was left of the kernels when the caller came to collect
    total = s.h_offsets[_QN];
  } else {
    if (hipStreamSynchronize(s.stream) != hipSuccess) throw std::runtime_error("queryKNN: the batch failed on the device");
    d_lastTiming.kernels_ms = s.issueMs + msSince(t);
  }
  if (!s.compact || total * 2 > n) {
    // dense (or small) result: the device arrays go straight into the caller's vectors
    t = std::chrono::steady_clock::now();
    d2h(_resIdx.data(), s.d_resIdx, n * 4);
    d2h(_resDist.data(), s.d_resDist, n * 4);
    d_padIdx = nullptr;  // (the row lengths of this hand-over are not known here)
    d_lastTiming.d2h_ms += msSince(t);
    d_lastTiming.host_ms = hostMs;
    d_lastTiming.total_ms = s.issueMs + msSince(tAll);
    d_lastTiming.d2h_bytes = 2 * n * 4 + (s.compact ? ((size_t)_QN + 1) * 4 : 0);
    d_lastTiming.columns = _nVec;
    d_lastTiming.packed = false;
    return;
  }
  t = std::chrono::steady_clock::now();
  if (total) {
    if (hipMemcpyAsync(s.h_stageIdx, s.d_packIdx, total * 4, hipMemcpyDeviceToHost, s.stream) != hipSuccess || hipEventRecord(s.evIdx, s.stream) != hipSuccess ||
        hipMemcpyAsync(s.h_stageDist, s.d_packDist, total * 4, hipMemcpyDeviceToHost, s.stream) != hipSuccess || hipEventRecord(s.evDist, s.stream) != hipSuccess)
      throw std::runtime_error("D2H copy failed");
  }
  if (!d_pool) d_pool = new HostPool(d_poolThreads);
  uint* const oi = _resIdx.data(); float* const od = _resDist.data();
  const uint* const off = s.h_offsets; const size_t nv = _nVec, qn = _QN;
  const uint32_t* const odBits = reinterpret_cast<const uint32_t*>(od);
  // the padding this storage already holds (left by the previous hand-over into the very same vectors): only what that batch filled
  // beyond this batch's prefix is written again
  const bool known = d_keepPadding && d_padIdx == oi && d_padDist == od && d_padQN == _QN && d_padNVec == _nVec && h_padCnt.size() == qn;
  if (h_padCnt.size() != qn) h_padCnt.assign(qn, 0u);
  uint* const prev = h_padCnt.data();
  // phase A (while the packed rows are in flight): the padding of both arrays
  d_pool->run_all([=](int tid, int nt) {
    for (size_t r = qn * tid / nt; r < qn * (tid + 1) / nt; ++r) {
      const size_t c = off[r + 1] - off[r];
      // the memory is trusted row by row only while the storage still shows it: the slot behind the previous prefix and the last slot
      // of the row must hold the sentinels the last hand-over left there (a vector that was reallocated at the same address, assigned
      // or filled in between does not) -- otherwise the whole row is padded like the reference does
      const size_t pr = prev[r];
      const bool rowKnown = known && (pr >= nv || (oi[r * nv + pr] == 0xffffffffu && oi[r * nv + nv - 1] == 0xffffffffu &&
                                                    odBits[r * nv + pr] == 0x7f800000u && odBits[r * nv + nv - 1] == 0x7f800000u));
      const size_t end = rowKnown ? std::max<size_t>(c, pr) : nv;
      if (end > c) { fillStream32(oi + r * nv + c, end - c, 0xffffffffu); fillStream32(od + r * nv + c, end - c, 0x7f800000u); }
      prev[r] = (uint)c;
    }
    fillFence();
  });
  d_padIdx = oi; d_padDist = od; d_padQN = _QN; d_padNVec = _nVec;
  hostMs += msSince(t);
  t = std::chrono::steady_clock::now();
  if (total && (hipEventSynchronize(s.evIdx) != hipSuccess || hipEventSynchronize(s.evDist) != hipSuccess)) throw std::runtime_error("D2H copy failed");
  d_lastTiming.d2h_ms += msSince(t);
  t = std::chrono::steady_clock::now();
  const uint* const si = s.h_stageIdx; const float* const sd = s.h_stageDist;
  // phase B: the packed rows into their places
  d_pool->run_all([=](int tid, int nt) {
    for (size_t r = qn * tid / nt; r < qn * (tid + 1) / nt; ++r) {
      const size_t o = off[r], c = off[r + 1] - o;
      if (c) { memcpy(oi + r * nv, si + o, c * 4); memcpy(od + r * nv, sd + o, c * 4); }
    }
  });
  hostMs += msSince(t);
  d_lastTiming.host_ms = hostMs;
  d_lastTiming.total_ms = s.issueMs + msSince(tAll);
  d_lastTiming.d2h_bytes = ((size_t)_QN + 1) * 4 + 2 * total * 4;
  d_lastTiming.columns = (uint)(total / _QN);
  d_lastTiming.packed = true;
}

void PerturbationProTree::queryBIGKNNRerank2(std::vector<uint>& _resIdx, std::vector<float>& _resDist, const float* _Q, uint _QN,
                                             uint _nVec, const float* /*_hlines*/) {
  queryKNN(_resIdx, _resDist, _Q, _QN, _nVec);
}

void PerturbationProTree::query(uint _boundVectors, uint _boundBins, const float* _vecHost, std::vector<std::pair<uint, float> >& _out) {
  pqt_index* h = handle();
  ensureHeuristic(_boundBins);
  // the whole sorted candidate list: k = upper bound of the list length
  pqt_stats st;
  uint k = 8192;
  std::vector<uint> idx; std::vector<float> dist; uint cnt = 0;
  for (;;) {
    idx.resize(k); dist.resize(k);
    if (d_multi) { if (pqt_multi_query_host(d_multi, _vecHost, 1, _boundVectors, _boundBins, k, idx.data(), dist.data(), &cnt) != PQT_OK) throw std::runtime_error(std::string("query: ") + pqt_multi_last_error()); }
    else check(pqt_query_host(h, _vecHost, 1, _boundVectors, _boundBins, k, idx.data(), dist.data(), &cnt), "pqt_query_host");
    if (cnt <= k) break;
    k = cnt;
  }
  (void)st;
  _out.clear();
  for (uint i = 0; i < cnt; ++i) _out.push_back(std::make_pair(idx[i], dist[i]));
}

pqt_stats PerturbationProTree::lastStats() {
  pqt_stats s;
  check(pqt_get_stats(handle(), &s), "pqt_get_stats");
  return s;
}

}  // namespace pqt
