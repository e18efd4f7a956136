export PQT_PROFILE_PMC_ONLY=1
bash scripts/r02_profile.sh r02_cfg3_100m_20000_500 2.0 synth100m 20000 500 100 2>&1 | tail -14
bash scripts/r02_profile.sh r02_cfg3_100m_4096_4096 2.0 synth100m 4096 4096 100 2>&1 | tail -14
