# Round 5 (session 2), call 6: alternating work queues (no memset at the head of the persistent launch) + overflow count mirrored per pipeline chunk (no k_ec_slow at the
# end of every chunk's chain): full GPU suite, A/B of the bench step / CJK-like config 4 / end-to-end path against the round-4 forms, 2-rank rehearsal (gloo, one GPU).
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 400 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_e.log 2>&1; tail -3 gpurun_out/r05_gputests_e.log
for v in "A=1" "MSDFHIP_QUEUE_MEMSET=1" "MSDFHIP_PIPELINE_OVERFLOW_PASS=1" "A=2" "MSDFHIP_QUEUE_MEMSET=1 MSDFHIP_PIPELINE_OVERFLOW_PASS=1"; do
  echo "== $v"; env $v python bench.py --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms'])"
  python tools/e2e_stream.py 9 $v 2>/dev/null | cut -c1-330
  env $v python tools/bench_configs.py --reps 6 --only "cfg4: 8192 CJK" 2>/dev/null | cut -c1-300
done > gpurun_out/r05_queue_ab2.txt 2>&1
cat gpurun_out/r05_queue_ab2.txt
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --same-device --no-cpu-baseline --no-extras > gpurun_out/r05_bench_2rank.json 2> gpurun_out/r05_bench_2rank.err; tail -c 700 gpurun_out/r05_bench_2rank.json
