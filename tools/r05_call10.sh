# Round 5 (session 2), call 10: four wavefronts per SIMD as the default -- full GPU suite, bench line, end-to-end path.
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
timeout 400 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_f.log 2>&1; tail -3 gpurun_out/r05_gputests_f.log
timeout 200 python bench.py > gpurun_out/r05_bench_f.json 2> gpurun_out/r05_bench_f.err; python -c "
import json; d=json.loads(open('gpurun_out/r05_bench_f.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms']); e=d['end_to_end_metric']; print(e['uint8_atlas_glyphs_per_s'], e['float_tiles_glyphs_per_s']); s=d['strong_scaling']; print({k: [s[k][x]['efficiency'] for x in ('x2','x4','x8')] for k in ('cjk_like','dejavu')}, s['cjk_like']['ms_whole_set'], s['dejavu']['ms_whole_set'])"
timeout 120 python tools/host_call_latency.py --threads 1,4,64 --leaders 4 2>/dev/null | cut -c1-200
