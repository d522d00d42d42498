mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
run() { echo "== $*"; env "$@" timeout 100 python tools/pipeline_chunks.py 0 2>/dev/null; }
(
run A=1
run MSDFHIP_PIPELINE_DEPTH=3
run MSDFHIP_PIPELINE_DEPTH=1
run MSDFHIP_PIPELINE_CLASSES=concurrent
run MSDFHIP_PIPELINE_CLASSES=concurrent MSDFHIP_PIPELINE_DEPTH=3
run MSDFHIP_PIPELINE_CLASSES=concurrent GPU_MAX_HW_QUEUES=16
run MSDFHIP_PIPELINE_DEPTH=3 GPU_MAX_HW_QUEUES=16
run MSDFHIP_PIPELINE_LENGTHS=512,1024,2048,2048,2048,512
run MSDFHIP_PIPELINE_LENGTHS=512,1536,2048,2048,1536,512
run MSDFHIP_PIPELINE_LENGTHS=1024,2048,2048,2048,512,512
run MSDFHIP_PIPELINE_LENGTHS=512,1024,1536,2048,2048,1024 MSDFHIP_PIPELINE_DEPTH=3
run MSDFHIP_PIPELINE_LENGTHS=1024,1024,1024,1024,1024,1024,1024,1024 MSDFHIP_PIPELINE_DEPTH=3
) > gpurun_out/r05_pipe2.txt 2>&1
cat gpurun_out/r05_pipe2.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/pt -o pt -- python $GRAFT_REPO_ROOT/tools/pipeline_timeline.py run bytes > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python tools/pipeline_timeline.py report /tmp/pt > gpurun_out/r05_timeline_bytes.txt 2>&1; tail -5 gpurun_out/r05_timeline_bytes.txt
