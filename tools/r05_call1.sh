# Round 5 (session 2), call 1: GPU suite on the uncommitted pipeline work + the bench line with the streamed end-to-end section.
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 300 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_c.log 2>&1; tail -3 gpurun_out/r05_gputests_c.log
timeout 200 python bench.py > gpurun_out/r05_bench_c.json 2> gpurun_out/r05_bench_c.err; tail -c 600 gpurun_out/r05_bench_c.json; tail -3 gpurun_out/r05_bench_c.err
