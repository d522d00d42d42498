# Round 5 (session 2), call 5: the persistent launch zeroes its own work queue (no memset kernel at the head of the class's chain) -- full GPU suite (incl. the RCCL
# self-tests), A/B of the bench step and of the end-to-end path against MSDFHIP_QUEUE_MEMSET=1, the bench line.
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 400 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_d.log 2>&1; tail -3 gpurun_out/r05_gputests_d.log
for v in "A=1" "MSDFHIP_QUEUE_MEMSET=1" "A=2" "MSDFHIP_QUEUE_MEMSET=1"; do
  echo "== $v"; env $v python bench.py --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms'])"
  python tools/e2e_stream.py 9 $v 2>/dev/null | cut -c1-330
  env $v python tools/bench_configs.py --reps 6 --only "cfg4: 8192 CJK" 2>/dev/null | cut -c1-300
done > gpurun_out/r05_queue_ab.txt 2>&1
cat gpurun_out/r05_queue_ab.txt
