// pqt/PerturbationProTree.hh -- host-side mirror of the reference's class surface for the query path, over the
// C-ABI of include/pqt_hip.h.  Same class names, method names, argument meaning and ownership rules as
//   pqt::ProQuantization      (pqt/ProQuantization.hh:21-94)
//   pqt::ProTree              (pqt/ProTree.hh:38-243)
//   pqt::PerturbationProTree  (pqt/PerturbationProTree.hh:28-235)
// restricted to what tool_query / tool_createdb / test1B call on the hot path; training and the
// dead experimental methods are not mirrored.  Results follow cpu_version semantics.
//
// Error behaviour mirrors the reference: file readers throw std::runtime_error (utils/filereader.hpp:52-56,
// treequantizer.hpp:702-705); device failures throw std::runtime_error carrying pqt_last_error() instead of the
// reference's exit(1)/abort (PerturbationProTree.cu:8229-8232).
#ifndef PQT_HOST_PERTURBATIONPROTREE_HH
#define PQT_HOST_PERTURBATIONPROTREE_HH

#include <stdint.h>
#include <string>
#include <vector>

#include "../../../include/pqt_hip.h"

typedef unsigned int uint;
// the two HIP handle types the class keeps (same declarations as hip_runtime_api.h, so that this header needs no HIP include)
typedef struct ihipStream_t* hipStream_t;
typedef struct ihipEvent_t* hipEvent_t;

namespace pqt {

class HostPool;  // PerturbationProTree.cpp

/** where a vector array lives: the reference's createTree / buildKBestDB take DEVICE pointers (ProTree.hh:82,
 *  PerturbationProTree.hh:53); the tools of this tree read files into host memory, so both are accepted */
enum MemSpace { HOST_PTR = 0, DEVICE_PTR = 1 };

/** 4-byte line code, layout of the reference's lineDescr (PerturbationProTree.hh:21-25) / code_t (helper.hpp:39-90) */
typedef struct {
  unsigned char p1;
  unsigned char p2;
  unsigned short lambda;
} lineDescr;

class ProQuantization {
 public:
  ProQuantization(uint _dim, uint _p);
  virtual ~ProQuantization();
  /** first-level codebook, HOST copy (the reference returns its device copy; the device copy here is owned by the handle) */
  const float* getCodeBook() const { return h_codeBook.data(); }
  uint getDim() const { return d_dim; }
  uint getP() const { return d_p; }

 protected:
  uint d_dim, d_p, d_vl, d_nClusters;
  std::vector<float> h_codeBook;
};

class ProTree : public ProQuantization {
 public:
  ProTree(uint _dim, uint _p, uint _p2);
  uint getClusters2() const { return d_nClusters2; }
  uint getNClusters() const { return d_nClusters; }
  const float* getCodebook1() const { return h_codeBook.data(); }
  /** traversal heuristic: the first _rows tuples of prepareHeuristic (treequantizer.hpp:75-127); the CUDA
   *  prepareDistSequence(int,int) (ProTree.hh:66) is its counterpart */
  void prepareDistSequence(uint _rows);
  /** the CUDA library's own heuristic with its signature (ProTree.hh:66, ProTree.cu:128-207): sum-of-sqrt order over digits
   *  < min(16, _maxCluster); _groupParts must be p.  Optional mode: changes the enumeration order only. */
  void prepareDistSequence(int _maxCluster, int _groupParts);
  /** the 1B path's 2-D anisotropic sequences with the reference's signature (ProTree.hh:69, ProTree.cu:50-126; test/test1B.cpp:941
   *  calls it with 512): 10 orders of the _maxCluster^2 grid; queries then pick their rows through the pairwise merges of
   *  PerturbationProTree.cu:2914-3100 (pqt_index_build_heuristic_2d).  p = 4.  Optional mode: changes which rows are enumerated. */
  void prepare2DDistSequence(int _maxCluster);

 protected:
  virtual pqt_index* handle() = 0;
  uint d_p2, d_nClusters2;
  std::vector<float> h_codeBook2;
};

class PerturbationProTree : public ProTree {
 public:
  PerturbationProTree(uint _dim, uint _p, uint _p2);
  ~PerturbationProTree();

  /** device selection (reference: cudaSetDevice(FLAGS_device), tool_query.cpp:74); call before reading a tree */
  void setDevice(int _device) { d_device = _device; d_devices.clear(); }
  /** several devices (no reference counterpart: the reference drives one device): the database is RANGE-SHARDED by vector id
   *  over _devices (the same device may appear more than once), the object keeps ONE class surface -- loadTree / loadBins /
   *  setBins / setLines / buildKBestDB / queryKNN / query behave as with one device and return the same results; a batch is
   *  fanned out on the devices' streams (pqt_multi_*, include/pqt_hip.h).  _Q of queryKNN and its results live on _devices[0].
   *  Call before reading a tree.  The hashed dump family (setDB / loadHashedDB) and the device-array getters stay
   *  single-device. */
  void setDevices(const std::vector<int>& _devices) { d_devices = _devices; if (!_devices.empty()) d_device = _devices[0]; }
  size_t getNDevices() const { return d_devices.size() > 1 ? d_devices.size() : 1; }
  /** shadows ProTree's: with several devices every shard gets the table */
  void prepareDistSequence(uint _rows);
  void prepareDistSequence(int _maxCluster, int _groupParts);
  void prepare2DDistSequence(int _maxCluster);
  /** W of treequantizer<..,W,..> / k1 of queryKNN (PerturbationProTree.cu:8187); call before reading a tree */
  void setW(uint _w) { d_w = _w; }
  /** query bounds of treequantizer::query(boundVectors, boundBins, ..) (cpu_version/tools/query.cpp:42) */
  void setBounds(uint _boundVectors, uint _boundBins) { d_boundVectors = _boundVectors; d_boundBins = _boundBins; }

  /** GPU .ppqt (ASCII header dim,p,p2,C1,C2,nDBs + cb1 + cb2; PerturbationProTree.cu:60-220) */
  void writeTreeToFile(const std::string& _name);
  void readTreeFromFile(const std::string& _name);
  /** CPU .tree (5 x u32 D,C1,C2,P,W + cb1 + cb2; treequantizer.hpp:699-737 / 782-837) */
  void saveTree(const std::string& _name);
  void loadTree(const std::string& _name);
  /** CPU .bins (treequantizer.hpp:745-774 / 845-893): bins + line codes */
  void loadBins(const std::string& _name);
  void saveBins(const std::string& _name);
  /** produces a two-layer product quantization tree with _k / _k2 centroids per level from _N training vectors
   *  (ProTree.hh:82 createTree; algorithm = cpu_version: productquantizer::generate productquantizer.hpp:131-158,
   *  vectorquantizer::generate vectorquantizer.hpp:117-146, treequantizer::generate treequantizer.hpp:155-177:
   *  k-means by centroid splitting).  _A: host pointer, or device pointer like the reference with _space = DEVICE_PTR.
   *  The E step (nearest centroid of every vector) runs on the GPU (pqt_kmeans_assign), the M step (sequential sums) on
   *  the host, so the result does not depend on any parallel reduction order. */
  void createTree(uint _k, uint _k2, const float* _A, uint _N, MemSpace _space = HOST_PTR);
  /** set the tree from host arrays cb1[C1][dim], cb2[p][C1][C2][dim/p] */
  void setTree(uint _c1, uint _c2, const float* _cb1, const float* _cb2);

  /** upload a previously stored db; host pointers, copied, caller keeps ownership (PerturbationProTree.hh:66).
   *  slot = bin id % hashSize like the CUDA library (HASH_SIZE, PerturbationProTree.hh:12). */
  void setDB(uint _N, const uint* _prefix, const uint* _counts, const uint* _dbIdx, uint _hashSize = 400000000u);
  /** exact (un-hashed) bins: ids[nbins], sizes[nbins], members[N] */
  void setBins(size_t _nbins, const uint* _ids, const uint* _sizes, const uint* _members);

  /** line store: the reference exposes only prepareEmptyLambda/getLine (PerturbationProTree.hh:101-103) and
   *  passes `_hlines` per call; here the codes are uploaded once. */
  void prepareEmptyLambda(uint _N, uint _lParts = 16);
  void setLines(const lineDescr* _lines, size_t _N);
  uint getLineParts() const { return d_lineParts; }

  /** insert = id() + prepareReranking for _N vectors (treequantizer.hpp:212-217; CUDA buildKBestDB + lineDist):
   *  fills the bin store and the line store of this object.  _A: host pointer, or device pointer like the reference
   *  (PerturbationProTree.hh:53) with _space = DEVICE_PTR. */
  void buildKBestDB(const float* _A, uint _N, MemSpace _space = HOST_PTR);
  /** chunked build (test/test1B.cpp:783-871: per-chunk buildKBestDB + lineDist, then the host-side CSR merge): vectors
   *  [_idOffset, _idOffset + _N) are assigned and line-encoded; finishDB() merges all chunks into one bin store (bins in
   *  ascending id order, members in id order = what one insert() pass over the whole dataset produces) and uploads it. */
  void buildKBestDBChunk(const float* _A, uint _N, uint _idOffset, MemSpace _space = HOST_PTR);
  void finishDB();
  void lineDist(const float* /*_DB*/, uint /*_N*/) {}  // line codes are produced by buildKBestDB in one pass

  /** _Q is a DEVICE pointer (like the reference), results are resized to _QN*_nVec (PerturbationProTree.cu:8182-8183).
   *  Unused slots: id 0xffffffff, distance +inf (the reference pads with 1e7). */
  void queryKNN(std::vector<uint>& _resIdx, std::vector<float>& _resDist, const float* _Q, uint _QN, uint _nVec);
  /** Two batches in flight behind the same contract (no reference counterpart: its loop, tool_query.cpp:152-160, answers one batch at a
   *  time).  queryKNNAsync enqueues the whole batch -- traversal, rerank, packing of the filled prefixes, the copy of the row lengths -- on
   *  one of TWO slots (the index and a view of it, pqt_index_create_view: own scratch, own stream, own result and staging buffers) and
   *  returns a ticket without waiting; queryKNNCollect(ticket, ..) waits for that batch and hands it over exactly like queryKNN (which
   *  is queryKNNAsync + queryKNNCollect).  A caller that issues batch i + 1 before it collects batch i has batch i + 1's kernels running
   *  under batch i's copies and host-side scatter.  At most two tickets are outstanding; tickets are collected in the order they were
   *  issued.  _Q must stay valid until the ticket is collected.  With several devices (setDevices) slot i is lane i of the multi handle
   *  (pqt_multi_query_lane: lane 1 = a view of every shard), so two batches are in flight there as well. */
  int queryKNNAsync(const float* _Q, uint _QN, uint _nVec);
  void queryKNNCollect(int _ticket, std::vector<uint>& _resIdx, std::vector<float>& _resDist);
  /** The padding of a row (id 0xffffffff, distance +inf behind its filled prefix) is remembered per result storage: a caller that hands
   *  the SAME two vectors back with the same shape -- the reference's loop does, tool_query.cpp:149-154 -- gets only the slots re-padded
   *  that the previous batch filled beyond the new one's prefix (110 MB of padding per 4096 x 4096 call otherwise, 1.1 of the call's
   *  1.6 ms).  false: always write the whole padding, like the reference.  Default FALSE: pointer identity and shape alone do not prove
   *  that the storage still holds the last hand-over's padding (a vector freed and reallocated at the same address, assign / fill
   *  between calls); callers that own their vectors for the whole loop opt in (host/tool_query.cpp, host/frontend_cabi.cpp).  With the
   *  memory on a row is still trusted only while its sentinels (the slot behind the previous prefix, the last slot) are in place. */
  void setKeepPadding(bool _on) { d_keepPadding = _on; d_padIdx = nullptr; }
  /** measurement only (bench.py's "legacy_copy" leg): always hand over with two whole-array copies (round 3's form; = PQT_FRONTEND_LEGACY_COPY at construction) */
  void setLegacyCopy(bool _on) { d_legacyCopy = _on; }
  /** same with line codes "in host memory": the codes live in HBM here, _hlines is ignored (kept for signature parity) */
  void queryBIGKNNRerank2(std::vector<uint>& _resIdx, std::vector<float>& _resDist, const float* _Q, uint _QN,
                          uint _nVec, const float* _hlines);
  /** host-pointer variant of treequantizer::query for one vector: sorted (id, distance) pairs */
  void query(uint _boundVectors, uint _boundBins, const float* _vecHost, std::vector<std::pair<uint, float> >& _out);

  /** the CUDA library's dump family (tool_createdb.cpp:111-138, test/test1B.cpp:865-892): dense hashed CSR
   *  <pre>.prefix / <pre>.count (hashSize u32 each, slot = bin id % hashSize; bins that share a slot are concatenated in
   *  ascending id order), <pre>.dbIdx (N u32) and <pre>_<lineparts>.lines (N x lineparts 4-byte lineDescr, vector-id
   *  order).  This is the .bins <-> triple translation: saveHashedDB writes it from the exact bins held, loadHashedDB
   *  reads it back through setDB (bins that shared a slot stay merged, exactly like in the CUDA library). */
  void saveHashedDB(const std::string& _pre, uint _hashSize = 400000000u);
  void loadHashedDB(const std::string& _pre, uint _N, uint _hashSize = 400000000u);
  void exportHashed(uint _hashSize, std::vector<uint>& _prefix, std::vector<uint>& _counts, std::vector<uint>& _dbIdx) const;

  /** DEVICE pointers like the reference's getters (PerturbationProTree.hh:97-103).  getDBIdx(): vector ids grouped by
   *  bin.  getLine(): the line codes indexed by VECTOR ID like the reference's d_lineLambda (PerturbationProTree.cu:5134;
   *  row i = code of vector i, the order of the .lines dump) -- an id-ordered device copy materialised from the host
   *  copy on the first call after the lines changed (the engine itself reads a bin-ordered store).  getLineBinOrder(): that
   *  bin-ordered store (row i belongs to getDBIdx()[i]).  getBinPrefix()/getBinCounts(): the dense hashed arrays of
   *  _hashSize entries, materialised on the first call after the bins changed (the engine keeps a compact two-choice
   *  table of the non-empty bins instead of these 2 x 1.6 GB); they need the exact bins (setBins / loadBins /
   *  buildKBestDB) and throw after setDB / loadHashedDB, whose bin ids are not recoverable. */
  const uint* getDBIdx();
  const lineDescr* getLine();
  const lineDescr* getLineBinOrder();
  const uint* getBinPrefix(uint _hashSize = 400000000u);
  const uint* getBinCounts(uint _hashSize = 400000000u);

  /** wall-clock split of the last queryKNN call (no reference counterpart; bench.py's frontend_queryKNN leg reads it) */
  struct CallTiming {
    double kernels_ms = 0;   // pqt_query / pqt_multi_query including its final synchronisation
    double d2h_ms = 0;       // waiting for the device-to-host copies (list lengths + the used columns of both arrays)
    double host_ms = 0;      // resizing the result vectors, scattering the staged rows into them, writing the padding
    double total_ms = 0;
    size_t d2h_bytes = 0;    // bytes that crossed PCIe
    uint columns = 0;        // mean entries copied per row (packed hand-over) or _nVec (whole-array copy)
    bool packed = false;     // the packed hand-over was taken (large result, at least half of it padding)
  };
  const CallTiming& lastCallTiming() const { return d_lastTiming; }

  uint getNPerturbations() const { return 1; }
  pqt_stats lastStats();
  const std::vector<uint>& binIds() const { return h_binIds; }

 protected:
  pqt_index* handle();
  void ensureHeuristic(uint rows);
  void check(int rc, const char* what);

  void releaseDeviceScratch();

  void uploadLines(size_t _N);
  void singleDeviceOnly(const char* what) const;

  pqt_index* d_idx;
  pqt_multi* d_multi;           // several devices: the shards live here and d_idx stays null
  std::vector<int> d_devices;
  // one batch in flight: the handle that runs it, its stream, its device result arrays and host staging (grown on demand, freed in the
  // destructor), and what queryKNNCollect needs to know about the batch
  struct KnnSlot {
    pqt_index* h = nullptr;                                   // slot 0: the index (or the multi handle's first shard); slot 1: a view of the index
    uint* d_resIdx = nullptr; float* d_resDist = nullptr; size_t resCap = 0;
    uint* d_resCnt = nullptr; uint* d_offsets = nullptr; uint* d_packIdx = nullptr; float* d_packDist = nullptr;  // list lengths, row offsets, packed rows (device)
    uint* h_stageIdx = nullptr; float* h_stageDist = nullptr; size_t stageCap = 0;                               // pinned host staging of the packed rows
    uint* h_offsets = nullptr; size_t cntCap = 0;
    hipStream_t stream = nullptr; hipEvent_t evOff = nullptr, evIdx = nullptr, evDist = nullptr;
    bool busy = false; uint QN = 0, nVec = 0; bool compact = false; double issueMs = 0;
  };
  KnnSlot d_slots[2]; unsigned d_issued, d_collected; int d_lastSlot;
  void ensureSlot(KnnSlot& s, size_t _n, size_t _qn, bool staging);
  void releaseSlot(KnnSlot& s);
  // padding memory (setKeepPadding): the storage the last hand-over padded, its shape and the filled prefix of every row
  bool d_keepPadding; const uint* d_padIdx; const float* d_padDist; uint d_padQN, d_padNVec; std::vector<uint> h_padCnt;
  // environment switches, read ONCE at construction: PQT_FRONTEND_LEGACY_COPY (any value: always the whole-array copy -- bench.py's
  // "legacy_copy" leg), PQT_FRONTEND_PACK_MIN_BYTES (smallest result, in bytes, that takes the packed hand-over; default 8 MiB; tests lower it),
  // PQT_FRONTEND_THREADS (host threads of the scatter / padding pool, default min(8, hardware threads))
  bool d_legacyCopy; size_t d_packMin; int d_poolThreads;
  HostPool* d_pool;                                                      // host threads that write the padding and scatter the packed rows
  CallTiming d_lastTiming;
  uint* d_hashPrefix; uint* d_hashCounts; uint d_hashSizeHeld;
  lineDescr* d_lineById; bool d_lineByIdValid;  // id-ordered device copy behind getLine()
  std::vector<uint> h_binOfVec;  // chunked build: bin id of every vector seen so far
  int d_device;
  uint d_w, d_lineParts, d_boundVectors, d_boundBins, d_heurRows;
  std::vector<uint> h_binIds, h_binSizes, h_members;
  std::vector<lineDescr> h_lines;
  size_t d_N;
};

}  // namespace pqt
#endif
