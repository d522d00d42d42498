# Schedules of the streamed end-to-end pipeline (8 192 glyphs at 64x64 from msdfgen::Shape objects): bash tools/r06_pipe_sweep.sh <tag>
TAG=$1
for L in "" "256,512,2048,2048,2048,768,512" "384,1024,2048,2048,1536,768,384" "512,1024,2048,2048,1536,768,256" "256,768,2048,2048,2048,1024" "512,2048,2048,2048,1024,512" "1024,2048,2048,2048,1024"; do
  python tools/e2e_stream.py 9 MSDFHIP_PIPELINE_LENGTHS=$L 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('%-44s u8 %.3f (min %.3f)  float %.3f (min %.3f)' % (d['env'].get('MSDFHIP_PIPELINE_LENGTHS') or 'default', d['uint8_atlas_ms'], d['uint8_atlas_ms_min'], d['float_tiles_ms'], d['float_tiles_ms_min']))"
done | tee gpurun_out/${TAG}_pipe_sweep.txt
