#!/bin/bash
# round 5: the 100 M configuration at (20000, 500) with the shared-row pass (the automatic choice) and without it, same box:
# kernel statistics, FETCH_SIZE / WRITE_SIZE / TCC hit-miss per kernel
cd $GRAFT_REPO_ROOT
export PQT_PMC_TIMEOUT=200
bash scripts/r05_profile.sh r05_cfg3_100m_20000_500_shared 2.0 synth100m 20000 500 100 > gpurun_out/prof_s.log 2>&1 < /dev/null
bash scripts/r05_profile.sh r05_cfg3_100m_20000_500_noshared 2.0 synth100m 20000 500 100 --option shared_rows=0 > gpurun_out/prof_n.log 2>&1 < /dev/null
tail -12 gpurun_out/prof_s.log | cut -c1-260
tail -8 gpurun_out/prof_n.log | cut -c1-260
cat gpurun_out/prof/r05_cfg3_100m_20000_500_shared_pmc.json | head -60
