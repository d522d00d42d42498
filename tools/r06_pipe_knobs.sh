# Knobs of the streamed pipeline on the final kernels (8 192 glyphs from msdfgen::Shape objects): bash tools/r06_call.sh <tag> r06_pipe_knobs.sh
TAG=$1
for K in "X=1" "MSDFHIP_PIPELINE_DEPTH=3" "MSDFHIP_PIPELINE_GATE=distance" "MSDFHIP_PIPELINE_CLASSES=concurrent" "MSDFHIP_PIPELINE_DEPTH=3 MSDFHIP_PIPELINE_GATE=distance" "MSDFHIP_HOST_THREADS=8" "MSDFHIP_HOST_THREADS=32" "X=1"; do
  timeout 120 python tools/e2e_stream.py 9 $K 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('%-60s u8 %.3f (min %.3f)  float %.3f (min %.3f)' % (' '.join('%s=%s'%kv for kv in d['env'].items()), d['uint8_atlas_ms'], d['uint8_atlas_ms_min'], d['float_tiles_ms'], d['float_tiles_ms_min']))"
done | tee gpurun_out/${TAG}_pipe_knobs.txt
