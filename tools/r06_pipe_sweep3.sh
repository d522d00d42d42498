for L in "" "384,1024,1024,1024,1024,1024,1024,1024,640" "256,768,1024,1024,1024,1024,1024,1024,1024" "512,1024,1024,1024,1024,1024,1024,1024,256,256" "640,1280,1280,1280,1280,1280,1152" ""; do
  timeout 120 python tools/e2e_stream.py 9 MSDFHIP_PIPELINE_LENGTHS=$L 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('%-60s u8 %.3f (min %.3f)  float %.3f (min %.3f)' % (d['env'].get('MSDFHIP_PIPELINE_LENGTHS') or 'default', d['uint8_atlas_ms'], d['uint8_atlas_ms_min'], d['float_tiles_ms'], d['float_tiles_ms_min']))"
done
