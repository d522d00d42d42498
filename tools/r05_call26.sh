for lib in msdfgen_amd/lib/libmsdfgen_hip.so variants/s6.so msdfgen_amd/lib/libmsdfgen_hip.so variants/s6.so; do echo "== $lib"; MSDFGEN_HIP_LIB=$PWD/$lib python tools/bench_configs.py --reps 8 --only "bench workload,headline,DejaVu glyphs msdf 64x64, simple" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-60s %.3f ms  %s' % (d['config'][:60], d['ms_per_step'], d.get('kernel_ms_distance_and_post')))"; done
