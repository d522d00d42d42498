# Round 5 (session 2), call 4: the new default chunk schedules (tests + end-to-end figures), the sanitizer driver incl. the streamed generator (libraries prebuilt in the authoring container).
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 300 python -m pytest tests -x -q -m gpu -k "pipeline or streamed or batch_entry or sharded or bytes" > gpurun_out/r05_tests_call4.log 2>&1; tail -3 gpurun_out/r05_tests_call4.log
for i in 1 2 3; do python tools/e2e_stream.py 9 A=$i 2>/dev/null | cut -c1-420; done | tee gpurun_out/r05_e2e_newdefault.jsonl
timeout 900 bash tests/sanitize/run.sh both > gpurun_out/r05_sanitizers.txt 2>&1; cat gpurun_out/r05_sanitizers.txt
tail -5 gpurun_out/sanitize_asan.log | cut -c1-300; tail -5 gpurun_out/sanitize_tsan.log | cut -c1-300
