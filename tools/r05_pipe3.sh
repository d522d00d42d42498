mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
run() { echo "== $*"; env "$@" timeout 100 python tools/pipeline_chunks.py 0 2>/dev/null; }
(
run A=1
run A=2
run MSDFHIP_PIPELINE_CLASSES=concurrent MSDFHIP_SHARE_GRID=0
run MSDFHIP_PIPELINE_CLASSES=concurrent MSDFHIP_SHARE_GRID=0 MSDFHIP_SIDE_PRIORITY=none
run MSDFHIP_PIPELINE_CLASSES=concurrent MSDFHIP_SHARE_GRID=0 MSDFHIP_PIPELINE_DEPTH=1
run MSDFHIP_PIPELINE_CLASSES=concurrent MSDFHIP_SHARE_GRID=0 MSDFHIP_PIPELINE_DEPTH=3
run MSDFHIP_PIPELINE_LENGTHS=1024,2048,2048,2048,512,512
run MSDFHIP_PIPELINE_LENGTHS=512,2048,2048,2048,1024,512
) > gpurun_out/r05_pipe3.txt 2>&1
cat gpurun_out/r05_pipe3.txt
for e in A=1 MSDFHIP_STREAM_UPLOAD=copy "MSDFHIP_PIPELINE_CLASSES=concurrent MSDFHIP_SHARE_GRID=0"; do python tools/e2e_stream.py 9 $e 2>/dev/null | cut -c1-300; done | tee gpurun_out/r05_e2e_3.jsonl
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/pt -o pt -- python $GRAFT_REPO_ROOT/tools/pipeline_timeline.py run bytes > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python tools/pipeline_timeline.py report /tmp/pt > gpurun_out/r05_timeline_bytes2.txt 2>&1; grep -c . gpurun_out/r05_timeline_bytes2.txt
