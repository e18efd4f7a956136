// pqt_fused_launch.hip -- the one-launch query kernel (pqt_k_query_fused: traversal + rerank/select of a query by the same wavefront),
// SIFT1M shape.  A translation unit of its own for build time, see pqt_internal.h.
#include "pqt_internal.h"

// LDS of a launch: coarse table + per wavefront max(traversal arena, rerank key slots + L1virt copy) + ticket words
size_t queryFusedPerWave(const pqt_index* idx, const TravPlan& tp) {
  const PqtDevParams& d = idx->dp;
  const size_t rs = (size_t)(PQT_RS_BEST + PQT_RS_PEND) * 8 + (size_t)d.LP * d.C1 * 4;
  return (std::max<size_t>(tp.perWave, rs) + 15) & ~(size_t)15;
}
bool queryFusedShape(const pqt_index* idx) {
  const PqtDevParams& d = idx->dp;
  return pqt_shape_of(d) == 1 && d.LP == 16 && d.C1 == 32;
}
int launchQueryFused(pqt_index* idx, const PqtTravArgs& targs, const TravPlan& tp, uint32_t grid, hipStream_t st, const float* qL1virt,
                     const uint32_t* nLocal, uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, hipEvent_t ev0, hipEvent_t ev1) {
  const PqtDevParams& d = idx->dp;
  const size_t perWave = queryFusedPerWave(idx, tp);
  static const int envNW = getenv("PQT_FUSED_NW") ? atoi(getenv("PQT_FUSED_NW")) : kFusedWaves;  // experiment: 8 wavefronts per workgroup (no spills, slower)
  const int nw = envNW == 8 ? 8 : kFusedWaves;
  const size_t lds = (size_t)d.LP * d.C1 * d.C1 * 4 + (size_t)nw * perWave + 16;
  if (lds > kMaxLds) return pqtFail(PQT_ERR_LIMIT, "one-launch query kernel does not fit the LDS");
#ifdef PQT_DEV_SIFT1M_ONLY
  (void)nw; (void)qL1virt; (void)nLocal; (void)stride; (void)k; (void)nq; (void)oI; (void)oD; (void)ev0; (void)ev1; (void)grid; (void)st; (void)targs;
  return pqtFail(PQT_ERR_LIMIT, "development build");
#else
  auto kern = nw == 8 ? pqt_k_query_fused<8, 4, 4, 5, 1> : pqt_k_query_fused<kFusedWaves, 4, 4, 5, 1>;
  int rc = allowLds(kern, lds);
  if (rc) return rc;
  const PqtRsArgs rargs{idx->d_codesBin, idx->d_ids, qL1virt, idx->d_coarse, idx->d_cand, idx->d_candPos, nLocal, stride, k, nq, idx->dp, oI, oD, nullptr,
                        idx->ctr, idx->dbg, nullptr, 0u, idx->curZero8,
                        nullptr, 0, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr,
                        nullptr, nullptr, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, 0u};
  hipExtLaunchKernelGGL(kern, dim3(grid), dim3(nw * 64), (uint32_t)lds, st, ev0, ev1, 0u, targs, rargs, (uint32_t)perWave);
  return PQT_OK;
#endif
}
