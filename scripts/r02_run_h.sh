bash scripts/r02_profile.sh r02_cfg2_sift1m 1.0 sift1m 20000 500 100 2>&1 | tail -25
