unset PQT_PROFILE_PMC_ONLY
bash scripts/r02_profile.sh r02_cfg2_sift1m 1.0 sift1m 20000 500 100 2>&1 | tail -4
bash scripts/r02_profile.sh r02_cfg3_100m_20000_500 2.0 synth100m 20000 500 100 2>&1 | tail -4
bash scripts/r02_profile.sh r02_cfg3_100m_4096_4096 2.0 synth100m 4096 4096 100 2>&1 | tail -4
PQT_BENCH_BACKEND=gloo PQT_BENCH_SAME_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --workload synth10m --steps 5 --warmup 2 > gpurun_out/prof/r02_shard2_gloo_same_device_synth10m.json 2> gpurun_out/prof/r02_shard2_gloo_same_device_synth10m.log
python bench.py --steps 20 --warmup 5 --extras > gpurun_out/prof/r02_bench_default_extras.json 2> gpurun_out/prof/r02_bench_default_extras.log
python bench.py --steps 20 --warmup 5 > gpurun_out/prof/r02_bench_default.json 2> gpurun_out/prof/r02_bench_default.log
python bench.py --workload synth100m --steps 10 --warmup 3 --no-cpu > gpurun_out/prof/r02_bench_synth100m.json 2> gpurun_out/prof/r02_bench_synth100m.log
python bench.py --workload synth100m --steps 10 --warmup 3 --no-cpu --option adc_bias=1 > gpurun_out/prof/r02_bench_synth100m_adc_bias.json 2> gpurun_out/prof/r02_bench_synth100m_adc_bias.log
python bench.py --workload synth10m --steps 10 --warmup 3 --cpu-seconds 10 > gpurun_out/prof/r02_bench_synth10m.json 2> gpurun_out/prof/r02_bench_synth10m.log
