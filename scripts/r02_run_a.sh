mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_a_bench_default.json 2> gpurun_out/r02_a_bench_default.log; tail -3 gpurun_out/r02_a_bench_default.log
PQT_BENCH_BACKEND=gloo PQT_BENCH_SAME_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --workload synth1m --steps 5 --warmup 2 > gpurun_out/r02_a_shard2_synth1m.json 2> gpurun_out/r02_a_shard2_synth1m.log; tail -5 gpurun_out/r02_a_shard2_synth1m.log; cat gpurun_out/r02_a_shard2_synth1m.json | head -c 3000
python bench.py --workload synth10m --steps 5 --warmup 2 --no-cpu > gpurun_out/r02_a_bench_synth10m.json 2> gpurun_out/r02_a_bench_synth10m.log; tail -3 gpurun_out/r02_a_bench_synth10m.log
