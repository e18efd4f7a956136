# final profile refresh of the round: headline + 100 M at both knob sets (kernel stats, FETCH/WRITE passes)
bash scripts/r02_profile.sh r02_cfg2_sift1m 1.0 sift1m 20000 500 100 > gpurun_out/prof_a.log 2>&1
bash scripts/r02_profile.sh r02_cfg3_100m_20000_500 2.0 synth100m 20000 500 100 > gpurun_out/prof_b.log 2>&1
bash scripts/r02_profile.sh r02_cfg3_100m_4096_4096 2.0 synth100m 4096 4096 100 > gpurun_out/prof_c.log 2>&1
tail -3 gpurun_out/prof_a.log gpurun_out/prof_b.log gpurun_out/prof_c.log | cut -c1-400
