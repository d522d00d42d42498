# Round 5 (session 2), call 9: k_distance's overlapping-combiner instantiations at 4 / 5 / 6 wavefronts per SIMD x LDS budgets that let a CU hold that many.
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
one() { # lib budget
  env MSDFGEN_HIP_LIB=$1 MSDFHIP_RES_LDS_BUDGET=$2 python tools/bench_configs.py --reps 6 --only "bench workload,cfg4: 8192 CJK,cfg4 real,cfg5,mtsdf" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-50s %.3f ms  %s' % (d['config'][:50], d['ms_per_step'], d.get('kernel_ms_distance_and_post')))"
}
(
echo "== main budget 13312"; one $PWD/msdfgen_amd/lib/libmsdfgen_hip.so 13312
for b in 11008 9984 9216; do echo "== w4 budget $b"; one $PWD/variants/w4.so $b; done
for b in 9984 7936 7168 6400; do echo "== w5 budget $b"; one $PWD/variants/w5.so $b; done
for b in 7936 6656 5888; do echo "== w6 budget $b"; one $PWD/variants/w6.so $b; done
) > gpurun_out/r05_waves456.txt 2>&1
cat gpurun_out/r05_waves456.txt
