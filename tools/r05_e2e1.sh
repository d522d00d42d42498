mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
python tools/e2e_stream.py 9 > gpurun_out/r05_e2e_default.json 2> gpurun_out/r05_e2e_default.err; cat gpurun_out/r05_e2e_default.json
for t in 1 2 4 8 16 32; do python tools/e2e_stream.py 7 MSDFHIP_HOST_THREADS=$t 2>/dev/null; done > gpurun_out/r05_e2e_threads.jsonl; cat gpurun_out/r05_e2e_threads.jsonl | cut -c1-330
python tools/e2e_stream.py 2 MSDFHIP_PIPELINE_TRACE=1 > /dev/null 2> gpurun_out/r05_e2e_trace.txt
timeout 120 python tools/pipeline_chunks.py 0 > gpurun_out/r05_pipe_resident.jsonl 2>/dev/null; cat gpurun_out/r05_pipe_resident.jsonl
timeout 60 tests/shim/shim_check flatten /dev/null 1 1 >/dev/null 2>&1
