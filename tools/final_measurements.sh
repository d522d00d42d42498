# Round-end measurements, part B (part A = tools/round_end.sh: tests + bench + tools/profile_round.sh): bash tools/final_measurements.sh <tag> [fuzz seed] [fuzz shapes]
TAG=${1:-r06}; SEED=${2:-601}; SHAPES=${3:-60000}
mkdir -p gpurun_out
timeout 500 python tools/bench_configs.py --reps 8 > gpurun_out/${TAG}_configs.jsonl 2> gpurun_out/${TAG}_configs.err
timeout 300 python tools/host_call_latency.py --threads 1,4,16,64 --leaders 4 > gpurun_out/${TAG}_host_calls.jsonl 2> gpurun_out/${TAG}_host_calls.err
timeout 200 python tools/host_call_latency.py --threads 1,4 --leaders 0 >> gpurun_out/${TAG}_host_calls.jsonl 2>> gpurun_out/${TAG}_host_calls.err
MSDFHIP_NO_FUSED_SINGLE=1 timeout 200 python tools/host_call_latency.py --threads 1,64 --leaders 4 > gpurun_out/${TAG}_host_calls_batched_path.jsonl 2>> gpurun_out/${TAG}_host_calls.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --same-device --no-cpu-baseline > gpurun_out/${TAG}_bench_2rank.json 2> gpurun_out/${TAG}_bench_2rank.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --strong --steps 10 --warmup 2 --same-device --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_2rank_strong.json 2> gpurun_out/${TAG}_bench_2rank_strong.err
timeout 300 python bench.py --inprocess --gpus 2 --same-device --steps 10 --warmup 2 > gpurun_out/${TAG}_bench_inprocess.json 2> gpurun_out/${TAG}_bench_inprocess.err
timeout 600 python tools/fuzz_parity.py --shapes $SHAPES --seed $SEED > gpurun_out/${TAG}_fuzz_$SEED.json 2> gpurun_out/${TAG}_fuzz_$SEED.err
timeout 400 python tools/fuzz_parity.py --shapes 12000 --seed $((SEED+1)) --single > gpurun_out/${TAG}_fuzz_single_$((SEED+1)).json 2> gpurun_out/${TAG}_fuzz_single_$((SEED+1)).err
bash tests/sanitize/run.sh both > gpurun_out/${TAG}_sanitizers.txt 2>&1
tail -2 gpurun_out/${TAG}_bench_2rank.json | cut -c1-300; tail -1 gpurun_out/${TAG}_bench_2rank_strong.json | cut -c1-300; tail -1 gpurun_out/${TAG}_bench_inprocess.json | cut -c1-300
cat gpurun_out/${TAG}_fuzz_$SEED.json gpurun_out/${TAG}_fuzz_single_$((SEED+1)).json; cat gpurun_out/${TAG}_sanitizers.txt; wc -l gpurun_out/${TAG}_configs.jsonl gpurun_out/${TAG}_host_calls.jsonl
