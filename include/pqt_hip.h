/* =====================================================================================
 * pqt_hip.h -- C-ABI of libpqt_hip.so: the MI355X (gfx950) Product-Quantization-Tree
 * query engine.  This is the drop-in boundary for the reference's query hot path
 * (per-query distance tables -> two-level traversal -> bin enumeration -> ADC line
 * rerank -> top-k).  Plain pointers and sizes only; no C++/torch types cross it.
 *
 * Semantics follow the reference's cpu_version (treequantizer<T,D,C1,C2,P,W,LP>,
 * cpu_version/quantizer/treequantizer.hpp); the class surface it sits under is the CUDA
 * library's pqt::PerturbationProTree (pqt/PerturbationProTree.hh).  Each entry point
 * cites the reference interface it replaces.  All citations are relative to the
 * reference repository root.
 *
 * Conventions
 *   - every function returns 0 (PQT_OK) or a negative pqt_status; pqt_last_error()
 *     returns a thread-local message for the last failure.  Nothing throws across
 *     the ABI and nothing calls exit() (the reference aborts: PerturbationProTree.cu:8229-8232).
 *   - "host" / "dev" in a parameter name says where the pointer must live.
 *   - a handle owns its device copies and a persistent scratch arena; the query
 *     path performs no allocation once the arena has been sized by a first call
 *     with the same (QN, Bv, Bb) or larger.
 *   - a handle is thread-compatible: one query batch in flight per handle.
 *   - there is NO CPU fallback: if no gfx950 device is usable the calls fail.
 * ===================================================================================== */
#ifndef PQT_HIP_H
#define PQT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pqt_index pqt_index; /* opaque */

typedef enum {
  PQT_OK = 0,
  PQT_ERR_INVALID = -1,   /* bad argument / parameter combination            */
  PQT_ERR_DEVICE = -2,    /* HIP runtime error (no device, OOM, launch fail) */
  PQT_ERR_STATE = -3,     /* index not fully populated for this call         */
  PQT_ERR_LIMIT = -4,     /* request exceeds an implementation limit         */
  PQT_ERR_IO = -5         /* file error (host-side helpers)                  */
} pqt_status;

/* Template parameters of treequantizer<T,D,C1,C2,P,W,LP> (treequantizer.hpp:15-26) /
 * constructor + file values of PerturbationProTree(dim,p,p2) (PerturbationProTree.hh:37).
 * Constraints: dim % p == 0, dim % lp == 0, lp % p == 0, 1 <= w <= c1 <= 256, c2 <= 256, p <= 8. */
typedef struct {
  uint32_t dim; /* D  */
  uint32_t p;   /* P  : parts of the product quantizer                      */
  uint32_t c1;  /* C1 : first-level centroids per part                      */
  uint32_t c2;  /* C2 : second-level centroids per first-level cell         */
  uint32_t w;   /* W  : first-level cells expanded per part (CUDA: k1)      */
  uint32_t lp;  /* LP : line parts of the rerank code (CUDA: lineparts)     */
} pqt_params;

/* Per-batch statistics of the last pqt_query* call on a handle. */
typedef struct {
  uint64_t queries;          /* QN of the last call                                      */
  uint64_t candidates;       /* sum over queries of reranked candidates (nCand)          */
  uint64_t bins_visited;     /* sum over queries of heuristic rows enumerated            */
  uint64_t bins_nonempty;    /* sum over queries of included non-empty bins              */
  uint64_t ties_l1;          /* exact float ties seen while ordering L1 cells            */
  uint64_t ties_l2;          /* ... while ordering the W*C2 second-level entries         */
  uint64_t ties_bins;        /* ... between adjacent bins of the sorted bin order        */
  uint64_t ties_final;       /* ... between adjacent candidates of the sorted result     */
  float ms_tables;           /* device time of the distance-table kernel (stage a1+a2)   */
  float ms_bins;             /* bin enumeration + probe + cut + candidate gather (a4-a6) */
  float ms_rerank;           /* ADC line rerank (a7)                                     */
  float ms_select;           /* final sort / top-k (a8)                                  */
  float ms_total;            /* whole batch on the device                                */
  uint32_t max_bin;          /* largest bin population in the index                      */
  uint32_t filter_fallbacks; /* queries of the last call (its last chunk) that the first rerank kernel handed to its back-up: band-filtered exact rerank (big coarse tables) -> plain exact kernel; short-list sort kernel (128 < k <= 4096) -> block-wide select kernel */
} pqt_stats;

const char* pqt_last_error(void);
/* number of visible HIP devices with gcnArchName gfx950 (0 => nothing will work) */
int pqt_device_count(void);

/* ---- lifetime ---------------------------------------------------------------------
 * replaces: treequantizer() ctor (treequantizer.hpp:38-50) / PerturbationProTree(dim,p,p2)
 * (PerturbationProTree.hh:37) + cudaSetDevice(FLAGS_device) (tool_query.cpp:74). */
int pqt_index_create(const pqt_params* prm, int device, pqt_index** out);
void pqt_index_destroy(pqt_index* idx);
int pqt_index_params(const pqt_index* idx, pqt_params* out);
/* A second handle on the SAME loaded index for a second batch in flight (a handle serves one batch at a time): the view shares
 * every array of `owner` (tree, heuristic, bins, line store and what the owner derives from them) and owns only scratch, a
 * stream and statistics, so two batches -- or the two halves of one (sharding.py: half B's kernels run under half A's
 * collectives) -- can be enqueued on two streams.  A view re-reads the owner's state at every call; it never builds shared data
 * itself: the owner must have served one call of the same kind before (PQT_ERR_STATE otherwise).  Query entry points only
 * (pqt_query*, pqt_traverse_bins, pqt_query_shard_bins, statistics, "stage_timing"); loading into a view is an error.
 * Destroyed by pqt_index_destroy(view) or with the owner.  The reference has no counterpart (one batch at a time on the
 * default stream, PerturbationProTree.cu:8179-8321). */
int pqt_index_create_view(pqt_index* owner, pqt_index** out);
/* tuning/debug options: "fused" = 1 (default) use the wave-per-query fused kernels (traversal; rerank+select) when
 * the request fits them, 0 = always use the workgroup-per-query staged kernels (which also keep the stage
 * intermediates readable by pqt_debug_read). Results are identical either way.
 * "small_lists" = 0: a 128 < k <= 4096 call sends every query through the block-wide select kernel (default 1: candidate lists
 * of <= 1024 entries are evaluated and sorted by one wavefront each, pqt_k_rerank_sort_small; same results).
 * "overlap" = 1 (2..4: that many pieces): pqt_query / pqt_query_shard / pqt_query_shard_bins run a batch as pieces on their own
 * streams (the later ones on view handles that share the index arrays and own their scratch), each piece's rerank launch on its
 * share of the workgroup slots.  Off by default (0 / -1): since the statistics atomics of the rerank were reduced to one per
 * workgroup the one-piece call is faster at every measured shape ("Round 3").  Calls that carry stage events
 * ("stage_timing") always run in one piece.  Same results; pqt_get_stats / pqt_debug_read cover all pieces.
 * "one_launch" = 1 (SIFT1M shape, bound_bins <= 512, k <= 128, unsharded): traversal and rerank/select of a query by the same
 * wavefront in one launch (pqt_k_query_fused).  Off by default: measured 0.191 against 0.167 ms per 10 k queries.  Same results.
 * "wg_rerank" = 0 disables the workgroup-per-query rerank kernel for large first-level codebooks (tuning).
 * "balance" = schedule of the wave-per-query rerank: -1 (default) = 2 for line stores beyond the 256 MiB Infinity Cache and
 * for the filtered rerank (coarse table too large for the LDS), 1 otherwise; 2 = per-XCD query pools in longest-first order (the traversal
 * registers every query under its size class), a share dealt out statically and the rest drawn in shrinking chunks, other
 * pools' leftovers when the own is empty; 1 = a fixed share per workgroup whose wavefronts draw it longest-first through an
 * LDS ticket; 0 = static round-robin.  It only changes the schedule, never a result.
 * "stage_timing" = N: the per-kernel start/stop events behind pqt_get_stage_ms_history / pqt_get_stats ride on every N-th
 * call (1 = every call, the default; 0 = never); they cost about 5 us per kernel launch.
 * "order_all_rows" = 1 makes the fused traversal order all enumerated rows instead of only the populated ones (the
 * fallback it takes by itself when more than 128 rows are populated); results are identical.
 * "exact_part_sorts" = 1 sends every query's second-level part lists through the one-list-at-a-time sort that settles near-ties of
 * the distances exactly (compile-time shapes: normally only a query in whose row-parallel sort two neighbours agree in the upper 26
 * bits of their distance keys takes it); results are identical (test switch).
 * "debug_bits" = ablation switches of the fused kernels (measurement only: results are WRONG for non-zero values;
 * scripts/ablate*.sh, PQT_DBG).
 * "enumerate_beyond_wrap" = 1: enumerable heuristic rows = the true (W*C2)^P, not the reference's uint32 product (which wraps to 0 at
 * BASELINE configs[4], where the reference therefore enumerates nothing): throughput-only mode WITHOUT a reference counterpart, used with
 * a supplied prefix (pqt_index_set_heuristic); bin ids keep the uint32 wrap-around.  Default 0.
 * "scratch_mb" = budget of the candidate arena in MiB (default 1/8 of device memory, at most 24 GiB): batches whose
 * candidate lists exceed it are processed in several chunks of queries. */
int pqt_index_set_option(pqt_index* idx, const char* name, int64_t value);

/* ---- tree ---------------------------------------------------------------------------
 * replaces: loadTree payload (treequantizer.hpp:782-837) / readTreeFromFile payload
 * (PerturbationProTree.cu:118-220): cb1[C1][D], cb2[P][C1][C2][D/P], f32 host pointers,
 * copied.  Also runs the coarse-table kernel = computeLookupTable (treequantizer.hpp:183-203)
 * / computeCBL1L1Dist (PerturbationProTree.cu:1902-1917). */
int pqt_index_set_codebooks(pqt_index* idx, const float* cb1_host, const float* cb2_host);
/* coarse[LP][C1][C1] (cpu layout (lp*C1 + A)*C1 + B) copied back to the host */
int pqt_index_get_coarse(const pqt_index* idx, float* out_host);

/* ---- traversal heuristic ----------------------------------------------------------------
 * replaces: prepareHeuristic (treequantizer.hpp:75-127) / prepareDistSequence (ProTree.cu:128-207).
 * build: enumerates all (W*C2)^P tuples and sorts them by squared norm on the host exactly as the
 * reference does (same comparator, std::sort), keeping the first `rows` rows.
 * set: takes a caller-supplied prefix tuples[rows][P] (e.g. dumped from a reference run). */
int pqt_index_build_heuristic(pqt_index* idx, uint64_t rows);
int pqt_index_set_heuristic(pqt_index* idx, const uint32_t* tuples_host, uint64_t rows);
/* Optional mode: the CUDA library's traversal heuristic, ProTree::prepareDistSequence(maxCluster, groupParts)
 * (pqt/ProTree.cu:128-207; called with (C2*k1, p) by queryKNN, PerturbationProTree.cu:8191): digits in base
 * min(16, max_cluster), tuples ordered by sum_p sqrt(digit) (f32, ties by tuple index), at most 65536 rows.  Changes the
 * ORDER in which bins are enumerated, nothing else (bin ids, cut and rerank stay cpu_version). */
int pqt_index_build_heuristic_cuda(pqt_index* idx, uint32_t max_cluster, uint64_t rows);
/* Optional mode: the CUDA library's 2-D anisotropic sequences of its 1B path, ProTree::prepare2DDistSequence(maxCluster)
 * (pqt/ProTree.cu:50-126; test/test1B.cpp:941 passes 512) and their per-query use (pqt/PerturbationProTree.cu:2839-3100:
 * computeSlopeIdx, generate2DBins, selectBinKernel2D2Parts, selectBinKernel2DFinal).  p = 4 only.  Builds the 10 cell orders
 * (key x^0.8 + s*y^0.8, 65536 cells each); from then on every query picks ITS rows: parts (0,1) and (2,3) are merged into two
 * 256-long pair lists through the order chosen by the slope of their sorted distances, the pair lists the same way, and row r of
 * the query is the tuple of four part ranks behind cell r of that order (cells outside the lists name no bin).  Changes the SET of
 * enumerated rows (at most min(65536, max_cluster^2), bound_bins <= 8192 as always; 16 <= max_cluster <= 4096), nothing else: bin ids, the exact sort of the
 * rows by distance, the cut and the rerank stay cpu_version; the CUDA kernels' sorting inside 1024-row chunks, 2-vectors-per-bin
 * cap and stop at k vectors are not reproduced.  Any other heuristic entry switches the mode off again. */
int pqt_index_build_heuristic_2d(pqt_index* idx, uint32_t max_cluster);
int pqt_index_get_heuristic(const pqt_index* idx, uint32_t* out_host, uint64_t rows);

/* ---- bin store ----------------------------------------------------------------------------
 * replaces: loadBins bin section (treequantizer.hpp:845-872): nbins records {bin id, size, members}.
 * bin_ids[nbins] (any order, unique), bin_sizes[nbins], members[sum sizes] (vector ids, bin by bin,
 * insertion order).  Host pointers, copied. */
int pqt_index_set_bins(pqt_index* idx, uint64_t nbins, const uint32_t* bin_ids_host,
                       const uint32_t* bin_sizes_host, const uint32_t* members_host);
/* Range shard for multi-GPU: same full description of the global bins, but only members with
 * id in [id_lo, id_hi) are kept on this device; every bin keeps its GLOBAL population so that all
 * shards apply the identical "finish the bin, then stop" cut (treequantizer.hpp:468-476). */
int pqt_index_set_bins_shard(pqt_index* idx, uint64_t nbins, const uint32_t* bin_ids_host,
                             const uint32_t* bin_sizes_host, const uint32_t* members_host,
                             uint32_t id_lo, uint32_t id_hi);
/* The same range shard described from the shard's side, for indices built shard by shard (each device encodes only its own
 * id range and never sees the other members): for every bin of the WHOLE database its id, its global population,
 * the number of its members held by lower-ranked shards (ids below this shard's range) and by this shard, and the
 * concatenated local member lists (ids ascending inside a bin = the reference's insertion order).  n_total = database
 * size over all shards.  Counterpart of the chunked build + CSR merge of test/test1B.cpp:783-871, with the merge
 * reduced to the per-bin counts (one all-gather at build time, see bench.py / sharding.py). */
int pqt_index_set_bins_local(pqt_index* idx, uint64_t nbins, const uint32_t* bin_ids_host, const uint32_t* global_sizes_host,
                             const uint32_t* lower_sizes_host, const uint32_t* local_sizes_host,
                             const uint32_t* local_members_host, uint64_t n_total);
/* replaces: PerturbationProTree::setDB(N, prefix, counts, dbIdx) (PerturbationProTree.hh:66,
 * .cu:1184-1229): the CUDA library's dense hashed CSR, slot = bin id % hash_size.  Host pointers. */
int pqt_index_set_db_hashed(pqt_index* idx, uint32_t n, const uint32_t* prefix_host,
                            const uint32_t* counts_host, const uint32_t* dbidx_host, uint32_t hash_size);

/* ---- line codes -------------------------------------------------------------------------------
 * replaces: loadBins code section (treequantizer.hpp:874-889) / the `_hlines` argument of
 * queryBIGKNNRerank2 (PerturbationProTree.hh:81) + prepareEmptyLambda/getLine (:101-103).
 * codes[nvec][LP], 4 bytes each = code_t {u8 p1; u8 p2; u16 lambda} (cpu_version/helper.hpp:39-90).
 * Row r holds the code of vector id (id_base + r).  _host copies; _dev adopts a device buffer
 * (caller keeps ownership and must keep it alive). */
int pqt_index_set_lines_host(pqt_index* idx, const uint32_t* codes_host, uint64_t nvec, uint64_t id_base);
int pqt_index_set_lines_dev(pqt_index* idx, const uint32_t* codes_dev, uint64_t nvec, uint64_t id_base);

/* ---- offline build on the device ("next" row: insert / prepareReranking) ------------------------
 * replaces: treequantizer::insert (treequantizer.hpp:212-217) = id() (:640-688) + prepareReranking
 * (:356-412) for n vectors; CUDA counterparts buildKBestDB / lineDist (PerturbationProTree.cu:1231,7663).
 * vecs_dev[n][D] f32 device pointer.  out_bin_dev[n] u32 bin ids, out_codes_dev[n][LP] line codes. */
int pqt_build_assign_encode(pqt_index* idx, const float* vecs_dev, uint64_t n,
                            uint32_t* out_bin_dev, uint32_t* out_codes_dev, void* hip_stream);

/* Exact re-rank of the first k results against the raw vectors ("next" row 8f-4; CUDA queryBIGKNNRerankPerfect
 * PerturbationProTree.hh:84, rerankBIGKernelPerfect .cu:5532): in_idx_dev[QN][k] (0xffffffff = empty) -> the same ids
 * ordered by exact squared L2 (f32, summed left to right; ties keep the previous order), with the distances.
 * raw_dev: row (id - raw_id_base) holds the vector, f32 or uint8 (raw_is_u8); 1 <= k <= 512.  Device pointers. */
int pqt_rerank_exact(pqt_index* idx, const float* q_dev, uint32_t qn, uint32_t k, const uint32_t* in_idx_dev,
                     const void* raw_dev, int raw_is_u8, uint64_t raw_id_base, uint64_t raw_rows,
                     uint32_t* out_idx_dev, float* out_dist_dev, void* hip_stream, int sync);

/* E step of the reference's k-means (productquantizer.hpp:40-66 getAssignment, vectorquantizer.hpp:33-53): nearest of
 * `ncen` centroids (rows of cen_dev, stride cen_ld) for n rows of x_dev (stride ld, optionally gathered through
 * rows_dev[n]) over `dim` dims; squared distances summed left to right, first minimum wins.  The M step (sequential
 * sums) stays on the host, which makes the whole training bit-reproducible (host/pqt/PerturbationProTree.cpp createTree). */
int pqt_kmeans_assign(int device, const float* x_dev, uint64_t n, uint32_t dim, uint32_t ld, const uint32_t* rows_dev,
                      const float* cen_dev, uint32_t ncen, uint32_t cen_ld, uint32_t* out_assign_dev, float* out_dist_dev,
                      void* hip_stream);

/* ---- query ----------------------------------------------------------------------------------------
 * replaces: treequantizer::query(boundVectors, boundBins, vec, out) (treequantizer.hpp:323-350) for a
 * batch, / PerturbationProTree::queryKNN(resIdx,resDist,Q,QN,nVec) (PerturbationProTree.hh:72).
 *   q_dev[QN][D] f32 device pointer; out_idx_dev[QN][k] u32; out_dist_dev[QN][k] f32;
 *   out_count_dev[QN] u32 (may be NULL) = candidate-list length nCand of each query.
 * Row q holds the first min(k, nCand) entries of the reference's sorted candidate list; unused slots
 * are 0xffffffff / +inf.  Equal distances are ordered by candidate visiting position (stable).
 * hip_stream: a hipStream_t to enqueue on; NULL = the handle's own NON-BLOCKING stream, which is not ordered with the
 * legacy default stream -- callers that produce the inputs asynchronously must pass the producing stream (or
 * synchronise first).  The call returns after the work is enqueued unless `sync` != 0. */
int pqt_query(pqt_index* idx, const float* q_dev, uint32_t qn, uint32_t bound_vectors, uint32_t bound_bins,
              uint32_t k, uint32_t* out_idx_dev, float* out_dist_dev, uint32_t* out_count_dev,
              void* hip_stream, int sync);
/* Host-pointer convenience wrapper (copies in/out, synchronous); the rate through this entry point
 * includes PCIe and is never the benchmark value. */
int pqt_query_host(pqt_index* idx, const float* q_host, uint32_t qn, uint32_t bound_vectors,
                   uint32_t bound_bins, uint32_t k, uint32_t* out_idx_host, float* out_dist_host,
                   uint32_t* out_count_host);
/* Hand-over helper for padded result arrays (the front-end's queryKNN(.., 4096) fills a few hundred of the 4096 slots of a row:
 * PerturbationProTree.cu:8278-8281 copies the padded arrays whole): offsets_dev[q] = sum_{i<q} min(count[i], k), offsets_dev[qn] = total,
 * and the first min(count, k) entries of every row of idx_dev / dist_dev [qn][k] packed back to back in packed_*_dev (capacity
 * qn * k).  Enqueued on hip_stream (NULL: the handle's own stream, i.e. behind a pqt_query issued with NULL).  Device pointers. */
int pqt_compact_results(pqt_index* idx, uint32_t qn, uint32_t k, const uint32_t* idx_dev, const float* dist_dev, const uint32_t* count_dev,
                        uint32_t* offsets_dev, uint32_t* packed_idx_dev, float* packed_dist_dev, void* hip_stream, int sync);
/* Oracle-parity entry: the reference's WHOLE sorted candidate list, treequantizer::query(boundVectors,
 * boundBins, vec, out) (treequantizer.hpp:323-350) for a batch -- row q of out_idx_dev / out_dist_dev [QN][cap] holds the
 * list of query q, out_count_dev[QN] (required) its true length; a list longer than `cap` is cut after its first cap
 * entries (call again with a larger cap).  Device pointers. */
int pqt_query_candidates(pqt_index* idx, const float* q_dev, uint32_t qn, uint32_t bound_vectors, uint32_t bound_bins,
                         uint32_t cap, uint32_t* out_idx_dev, float* out_dist_dev, uint32_t* out_count_dev,
                         void* hip_stream, int sync);
/* replaces: PerturbationProTree::getDBIdx() / getLine() (PerturbationProTree.hh:99-103), which hand out the DEVICE arrays
 * of the loaded database: ids_dev[n_local] = vector ids grouped by bin (the order the bins were handed over), and
 * codes_bin_dev[n_local][LP] = the line codes in the same order (row i = code of ids_dev[i]).  Owned by the handle, valid
 * until the bins / lines are replaced.  Each out pointer may be NULL. */
int pqt_index_device_arrays(const pqt_index* idx, const uint32_t** ids_dev, const uint32_t** codes_bin_dev, uint64_t* n_local);
/* Device memory held by a handle, in bytes: out8[0] id-ordered line store (owned copy, dropped once the bin-ordered one exists), [1] bin-ordered
 * line store, [2] its group-major copy (coarse table beyond the LDS), [3] its X-code copy (LDS-table rerank at C1 = 32), [4] row bias + member
 * ids, [5] bin table + presence bitmap, [6] codebooks, coarse table, heuristic prefix, [7] the scratch arena as sized by the calls so far.
 * (bench.py reports them as config.device_bytes: the copies of the line store are what an index costs per GPU.) */
int pqt_index_device_bytes(const pqt_index* idx, uint64_t* out8);
/* Multi-GPU merge helper: out of `nshards` per-shard results (as gathered by an all-gather of pqt_query_shard
 * outputs) produce the global first-k per query.  Each of idx/dist/pos points at shard 0's [QN][k] block; the block of
 * shard s starts shard_stride 32-bit words later (0 = QN*k, i.e. [shard][QN][k]; 3*QN*k when the three arrays of a
 * shard travel as one [3][QN][k] message).  Device pointers. */
int pqt_merge_topk(pqt_index* idx, uint32_t nshards, uint32_t qn, uint32_t k,
                   const uint32_t* idx_dev, const float* dist_dev, const uint32_t* pos_dev, uint64_t shard_stride,
                   uint32_t* out_idx_dev, float* out_dist_dev, void* hip_stream, int sync);
/* Shard-local query: like pqt_query but also returns each result's global visiting position
 * (out_pos_dev[QN][k]) so that merges break ties exactly like the unsharded engine. */
int pqt_query_shard(pqt_index* idx, const float* q_dev, uint32_t qn, uint32_t bound_vectors,
                    uint32_t bound_bins, uint32_t k, uint32_t* out_idx_dev, float* out_dist_dev,
                    uint32_t* out_pos_dev, uint32_t* out_count_dev, void* hip_stream, int sync);

/* ---- query-sharded traversal (multi-GPU) -------------------------------------------------------------------
 * The traversal (stages a1..a6 up to the cut) does not depend on which vectors a shard holds, so W shards need not all run it
 * for all queries: shard r runs pqt_traverse_bins for QN/W queries, the results are exchanged (one all-gather of <= (cap + 1)
 * x 8 bytes per query) and every shard reranks its own slice of the database from them with pqt_query_shard_bins.  The
 * reference has no counterpart (one device: cudaSetDevice(FLAGS_device), tool_query.cpp:74); the stages are those of
 * treequantizer::query (treequantizer.hpp:323-350): id + segmentInfo + orderBins + the cut of rerankVectors (:450-477).
 *   out_bins_dev[qn][cap + 1] u64, shard independent: entry i of a row = bin id | (global visiting position of the bin's first
 *   member << 32) of the i-th included populated bin in visiting order; the trailer word [cap] = number of entries | (global
 *   candidate count << 32).  A count of 0xffffffff marks a query whose list does not fit `cap` entries (1 <= cap <= 256), or
 *   that the fused traversal could not finish: pqt_query_shard_bins traverses such queries itself.  Sharded indices only. */
int pqt_traverse_bins(pqt_index* idx, const float* q_dev, uint32_t qn, uint32_t bound_vectors, uint32_t bound_bins, uint32_t cap,
                      unsigned long long* out_bins_dev, void* hip_stream, int sync);
/* pqt_query_shard with the traversal results handed in (bins_dev[qn][cap + 1] as written by pqt_traverse_bins on any shard of
 * the same database): distance tables for all queries, the listed bins resolved against this shard's bin table, ADC rerank of
 * the local members, top-k with global visiting positions.  Identical output to pqt_query_shard. */
int pqt_query_shard_bins(pqt_index* idx, const float* q_dev, uint32_t qn, uint32_t bound_vectors, uint32_t bound_bins, uint32_t k,
                         const unsigned long long* bins_dev, uint32_t cap, uint32_t* out_idx_dev, float* out_dist_dev,
                         uint32_t* out_pos_dev, uint32_t* out_count_dev, void* hip_stream, int sync);

/* ---- one handle over a range-sharded database (SURVEY 8b "multi-GPU handle fans out internally", 8e) ---------------------
 * N shard indices on N devices of one node inside ONE process (csrc/pqt_multi.cpp).  The reference has no counterpart: its
 * classes drive one device (cudaSetDevice(FLAGS_device), tool_query.cpp:74; queryKNN PerturbationProTree.hh:72).  Shard s owns
 * the vector ids [s*N/S, (s+1)*N/S): its rows of the line store and its members of every bin; tree, heuristic and every bin's
 * global population are replicated.  pqt_multi_query fans a batch out on the shards' streams -- shard s traverses query slice
 * s (pqt_traverse_bins), the bin lists are exchanged by peer copies (xGMI), every shard reranks its slice of the database
 * (pqt_query_shard_bins), the per-shard top-k lists are copied to the first device and merged by (distance, global visiting
 * position) (pqt_merge_topk) -- ordered by events, no host synchronisation inside a batch.  The result is the unsharded
 * engine's, bit for bit.  devices[s] = HIP device of shard s (NULL: 0, 1, ...; the same device may appear more than once).
 * Errors: as above, text in pqt_multi_last_error(). */
typedef struct pqt_multi pqt_multi;
const char* pqt_multi_last_error(void);
int pqt_multi_create(const pqt_params* prm, int nshards, const int* devices, pqt_multi** out);
void pqt_multi_destroy(pqt_multi* m);
int pqt_multi_shards(const pqt_multi* m);
pqt_index* pqt_multi_shard(pqt_multi* m, int s);  /* the shard's own handle (statistics, options, debug read-back) */
int pqt_multi_shard_range(const pqt_multi* m, int s, uint64_t* id_lo, uint64_t* id_hi);
/* pqt_index_set_option on every shard; "replicated_traversal" = 1 makes every shard traverse the whole batch itself */
int pqt_multi_set_option(pqt_multi* m, const char* name, int64_t value);
int pqt_multi_set_codebooks(pqt_multi* m, const float* cb1_host, const float* cb2_host);   /* = pqt_index_set_codebooks */
int pqt_multi_build_heuristic(pqt_multi* m, uint64_t rows);                                /* = pqt_index_build_heuristic */
int pqt_multi_build_heuristic_cuda(pqt_multi* m, uint32_t max_cluster, uint64_t rows);     /* = pqt_index_build_heuristic_cuda, every shard gets the table */
int pqt_multi_build_heuristic_2d(pqt_multi* m, uint32_t max_cluster);                        /* = pqt_index_build_heuristic_2d on every shard (each traverses all queries itself in this mode) */
int pqt_multi_set_heuristic(pqt_multi* m, const uint32_t* tuples_host, uint64_t rows);
/* the WHOLE database as for pqt_index_set_bins (members of a bin in ascending id order = the reference's insertion order);
 * every shard keeps its id range.  n_total = number of vectors (0: the sum of the bin sizes). */
int pqt_multi_set_bins(pqt_multi* m, uint64_t nbins, const uint32_t* bin_ids_host, const uint32_t* bin_sizes_host,
                       const uint32_t* members_host, uint64_t n_total);
/* codes[nvec][LP] of the whole database (row r = vector id r); every shard copies its rows */
int pqt_multi_set_lines_host(pqt_multi* m, const uint32_t* codes_host, uint64_t nvec);
/* = pqt_query for the sharded database: q / out_* live on the FIRST shard's device; hip_stream: a stream of that device or
 * NULL (the handle's own) */
int pqt_multi_query(pqt_multi* m, const float* q_dev0, uint32_t qn, uint32_t bound_vectors, uint32_t bound_bins, uint32_t k,
                    uint32_t* out_idx_dev0, float* out_dist_dev0, uint32_t* out_count_dev0, void* hip_stream, int sync);
int pqt_multi_query_host(pqt_multi* m, const float* q_host, uint32_t qn, uint32_t bound_vectors, uint32_t bound_bins, uint32_t k,
                         uint32_t* out_idx_host, float* out_dist_host, uint32_t* out_count_host);
/* Two batches in flight behind one multi handle (no reference counterpart: its loop answers one batch at a time, tool_query.cpp:152-160;
 * host/pqt/PerturbationProTree::queryKNNAsync with setDevices uses it).  lane 0 = the shard indices (pqt_multi_query is lane 0), lane 1 = a
 * view of every shard (pqt_index_create_view: own scratch, same loaded shard) with streams, events and exchange buffers of its own, created on
 * first use.  Two calls on different lanes with sync = 0 (and different hip_stream arguments, or NULL) overlap; a lane answers its batches in
 * the order they were issued.  Outputs and the query array must stay valid until the lane's batch has completed on its stream. */
int pqt_multi_query_lane(pqt_multi* m, int lane, const float* q_dev0, uint32_t qn, uint32_t bound_vectors, uint32_t bound_bins, uint32_t k,
                         uint32_t* out_idx_dev0, float* out_dist_dev0, uint32_t* out_count_dev0, void* hip_stream, int sync);

/* ---- stage-level read-back (parity tests; not a fast path) ------------------------------------------
 * After a pqt_query* call the handle still holds the intermediates of that batch:
 *   l1virt[QN][LP][C1]                      = _L1distancesVirtual   (treequantizer.hpp:914)
 *   seg_d2[QN][P][W*C2], seg_bin[...]       = segmentInfo output sorted by d2 (:597-630); seg_bin = l1*C2+l2
 *   cand_idx[QN][stride], cand_dist[...]    = candidates in visiting order (rerankVectors before its sort)
 * Each pointer may be NULL.  Host pointers.  stride is returned by pqt_debug_stride(). */
uint64_t pqt_debug_stride(const pqt_index* idx);
/* per-query phase timestamps of the fused traversal kernel (only when the process runs with PQT_TSTAMP=1):
 * out[qn][24] shader-clock ticks (0-8 traversal phase stamps, 9-14 rerank: start, row wait, ADC + filter, flushes, total | wall
 * clock, slot; 15 traversal hw id; 16-19 rerank detail: set-up, exact re-evaluation of the band, result write-out, candidates) */
int pqt_debug_tstamps(const pqt_index* idx, unsigned long long* out_host, uint32_t qn);
int pqt_debug_read(const pqt_index* idx, uint32_t qn, float* l1virt_host, float* seg_d2_host,
                   uint32_t* seg_bin_host, uint32_t* cand_idx_host, float* cand_dist_host,
                   uint32_t* ncand_host);

/* PMC calibration probe: `gathers` random reads of row_bytes (64|128) rows, one lane per row, 16 bytes per load, from a
 * table of 2^log2_rows rows (choose it far above the 256 MiB Infinity Cache); known bytes = gathers*row_bytes.  Run it
 * under `rocprofv3 --pmc FETCH_SIZE` (scripts/pmc_calibrate.sh) to get the counter-to-bytes ratio of this access shape. */
int pqt_debug_calibrate_gather(int device, uint32_t log2_rows, uint32_t row_bytes, uint64_t gathers, float* out_ms);
/* Read-only streaming probe: `reps` launches of a kernel that reads every 16-byte piece of a `bytes`-sized buffer once
 * (coalesced, four loads per lane in flight); *out_ms = mean launch time.  bench.py reports bytes / time beside the nominal
 * HBM peak (choose bytes far above the 256 MiB Infinity Cache). */
int pqt_debug_stream_read(int device, uint64_t bytes, int reps, float* out_ms);
/* The reference's own self-checks of its sort / scan primitives (pqt/bitonicSort.cuh:213-252: sortTestLarge sorts the values N - tid
 * carrying tid and expects payload[tid] == N - tid - 1; scanTestLarge expects the exclusive scan of ones == tid; N = 1024, 2048,
 * 4096), run through THIS library's primitives by a one-workgroup kernel: mode 0 = in-register wave sorting network
 * (pqt_wave_sort_u64, n <= 2048), 1 = block-wide bitonic network in LDS, 2 = wave radix select (out[j] = payload of the (j+1)-th
 * smallest key for every j; n = 512 | 1024), 3 = block radix select (64 ranks n/64 apart, the other entries stay 0xffffffff),
 * 4 = wave scan, 5 = block scan (out[n] = the total).  out_host[n + 1].  n a power of two, 64..8192.
 * Two more modes check the cross-lane primitives themselves (n is ignored): 6 = the lane exchanges of the sorting networks,
 * out_host[i * 64 + lane] = 0x9e3779b9 * (partner + 1) with partner = lane ^ (1 << i), i = 0..5 (DPP moves, gfx950 row / half swaps), i = 6: partner = lane + 1 (63: itself),
 * i = 7: the 64 keys (0x9e3779b9 * (lane + 1)) | 1 sorted by the u32 network;
 * 7 = inclusive wave scan of v(lane) = (lane * 2654435761 mod 2^32) >> 24 in out_host[0..63] (out_host must hold 513 words);
 * 8 = the four-lists-at-once row network of the traversal's part sorts (pqt_row_sort64_u32): out_host[64 p + e] = e-th smallest of the 64
 * keys (0x9e3779b9 * (64 p + j + 1)) | 1, j = 0..63, for each of the four 16-lane rows p. */
int pqt_debug_sort_scan(int device, uint32_t mode, uint32_t n, uint32_t* out_host);
int pqt_get_stats(const pqt_index* idx, pqt_stats* out);
/* Which kernels the last pqt_query* call launched, as text (tests assert the path taken, not only the result):
 * "traverse=<fused|fused-wide|staged>[-shape1|-shape2|-p2|-generic][-f1] rerank=<lds-table|l2-table|mode1-nwN|mode2-nwN|wg-gG|
 * big-*|staged-*>[-runs] chunks=<n>".  Returns the length written (<= cap - 1, NUL terminated). */
int pqt_get_last_path(const pqt_index* idx, char* out, int cap);
/* What the shared-row pass of the filtered rerank (csrc/pqt_shared_rows.h; no reference counterpart: the reference reads a line code per
 * candidate, cpu_version/quantizer/treequantizer.hpp:423-439,462-476) read and wrote for the last batch, counted on the device when
 * pqt_index_set_option("sr_stats", 1) is on: out8 = {bins in the per-batch table, (query, bin) pairs, distinct rows, rows the evaluating
 * kernel reads (a bin's rows once per chunk of 8 queries), items, queries with candidates the pass did not cover (handed back to the exact
 * kernels), filter distances written for the covered queries, capacity flag (non-zero: items or lists did not fit, every query handed
 * back)}.  bench.py prices the evaluating kernel's roofline with these.  PQT_ERR_STATE when the last call did not take the pass. */
int pqt_get_shared_rows_stats(const pqt_index* idx, uint64_t* out8);
/* duration (ms) of each launch of the dominant kernel (rerank) in the most recent query call that carried per-kernel events
 * (with "stage_timing" = N > 1 not every call does; 0 launches are reported when none of the last 32 calls did), via HIP
 * events on the stream it ran on; returns the number of launches written (<= cap).  pqt_get_stats reports its ms_* fields
 * for the same call. */
int pqt_get_rerank_launch_ms(const pqt_index* idx, float* out_ms, int cap);
/* per-stage device times of the most recent query calls (ring of 32), oldest first: out[n][5] = {tables, traversal/bins,
 * gap, rerank(+select when fused), select} in ms; returns n (<= cap).  On the fused path the start/stop timestamps ride
 * on the two kernel dispatches themselves (hipExtLaunchKernel; no event packets on the stream): "traversal" and "rerank"
 * are the pure durations of pqt_k_traverse and of the rerank+select kernel, "gap" is the time between them, "tables" and
 * "select" are 0 (those stages are inside the two fused kernels).  The staged path records one HIP event per stage
 * (gap = 0). */
int pqt_get_stage_ms_history(const pqt_index* idx, float* out_ms, int cap);

/* ---- scalar helpers (line-quantisation arithmetic; known-answer tests of run.cu:33-113) ----------------
 * computed ON THE DEVICE by a one-thread kernel, so they pin the kernels' arithmetic, not the host's. */
int pqt_dev_triangle(const float* a_host, const float* b_host, const float* c_host, const float* l_host,
                     uint32_t n, float* out_dist_host, float* out_ratio_host, uint16_t* out_lambda_u16_host,
                     float* out_lambda_roundtrip_host, int device);

#ifdef __cplusplus
}
#endif
#endif /* PQT_HIP_H */
