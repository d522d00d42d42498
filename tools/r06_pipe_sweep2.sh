TAG=$1
for L in "" "768,2048,2048,2048,1024,256" "512,1536,2048,2048,1536,512" "768,1536,2048,2048,1280,512" "1024,2048,2048,2048,768,256" "768,2048,2560,2048,768" ""; do
  python tools/e2e_stream.py 9 MSDFHIP_PIPELINE_LENGTHS=$L 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('%-44s u8 %.3f (min %.3f)  float %.3f (min %.3f)' % (d['env'].get('MSDFHIP_PIPELINE_LENGTHS') or 'default', d['uint8_atlas_ms'], d['uint8_atlas_ms_min'], d['float_tiles_ms'], d['float_tiles_ms_min']))"
done | tee gpurun_out/${TAG}_pipe_sweep.txt
