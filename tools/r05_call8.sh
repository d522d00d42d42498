# Round 5 (session 2), call 8: k_distance's overlapping-combiner instantiations at FOUR wavefronts per SIMD (128 VGPRs; the spills land outside the edge loop:
# tools/isa_loop_depth.py) together with an LDS budget that lets a CU actually hold 16 of them (10 KB per wavefront; at the default 13 KB the LDS caps the CU at 12
# whatever the registers allow -- which is what round 2's "4 waves: 3.9 vs 3.7 ms" measured).
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
one() { # lib budget
  env MSDFGEN_HIP_LIB=$1 MSDFHIP_RES_LDS_BUDGET=$2 python tools/bench_configs.py --reps 6 --only "bench workload,cfg4: 8192 CJK,cfg4 real" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-50s %.3f ms  %s' % (d['config'][:50], d['ms_per_step'], d.get('kernel_ms_distance_and_post')))"
}
(
for b in 13312 9984 8448 6912; do echo "== main budget $b"; one $PWD/msdfgen_amd/lib/libmsdfgen_hip.so $b; done
for b in 13312 9984 8448 6912 5376; do echo "== w4 budget $b"; one $PWD/variants/w4.so $b; done
) > gpurun_out/r05_waves4.txt 2>&1
cat gpurun_out/r05_waves4.txt
