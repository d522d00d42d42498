# Round 5 (session 2), call 2: colouring-kernel LDS tiers (tests), pipeline gate (distance pass done vs all kernels done) and chunk schedules on the streamed end-to-end path.
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 300 python -m pytest tests -x -q -m gpu -k "preparation or streamed or pipeline or batch_entry" > gpurun_out/r05_tests_call2.log 2>&1; tail -3 gpurun_out/r05_tests_call2.log
run() { python tools/e2e_stream.py 9 "$@" 2>/dev/null | cut -c1-400; }
(
run A=1
run MSDFHIP_PIPELINE_GATE=distance
run MSDFHIP_PIPELINE_LENGTHS=1024,2048,2048,2048,512,512
run MSDFHIP_PIPELINE_GATE=distance MSDFHIP_PIPELINE_LENGTHS=1024,2048,2048,2048,512,512
run MSDFHIP_PIPELINE_GATE=distance MSDFHIP_PIPELINE_LENGTHS=1024,1536,1536,1536,1536,1024
run MSDFHIP_PIPELINE_GATE=distance MSDFHIP_PIPELINE_DEPTH=1
run MSDFHIP_PIPELINE_GATE=distance MSDFHIP_PIPELINE_DEPTH=1 MSDFHIP_PIPELINE_LENGTHS=1024,1024,1024,1024,1024,1024,1024,1024
run MSDFHIP_PIPELINE_LENGTHS=768,2048,2048,2048,768,512
run A=2
) > gpurun_out/r05_e2e_gate.jsonl 2>&1
cat gpurun_out/r05_e2e_gate.jsonl | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    print(d['env'], 'float', d['float_tiles_ms'], d['float_tiles_ms_min'], 'u8', d['uint8_atlas_ms'], d['uint8_atlas_ms_min'])
"
