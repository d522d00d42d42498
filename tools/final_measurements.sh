mkdir -p gpurun_out
timeout 400 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
timeout 500 python tools/bench_configs.py --reps 8 > gpurun_out/r03_configs.jsonl 2> gpurun_out/r03_configs.err
timeout 300 python tools/host_call_latency.py --threads 1,4,16,64 > gpurun_out/r03_host_calls.jsonl 2> gpurun_out/r03_host_calls.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --same-device --no-cpu-baseline > gpurun_out/r03_bench_2rank.json 2> gpurun_out/r03_bench_2rank.err
timeout 300 python bench.py --inprocess --gpus 2 --same-device --steps 10 --warmup 2 > gpurun_out/r03_bench_inprocess.json 2> gpurun_out/r03_bench_inprocess.err
tail -c 600 gpurun_out/r03_bench.json; tail -2 gpurun_out/r03_bench_2rank.json | cut -c1-400; tail -1 gpurun_out/r03_bench_inprocess.json | cut -c1-400; wc -l gpurun_out/r03_configs.jsonl gpurun_out/r03_host_calls.jsonl
