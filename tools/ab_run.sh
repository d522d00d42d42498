# scratch script for one gpurun call: parity tests, then A/B of kernel build variants on the BASELINE configs
ONLY="headline,Basic-Latin msdf 64x64, simple,bench workload,DejaVu glyphs msdf 64x64, simple,cfg4: 8192 CJK,cfg4 real,cfg5,mtsdf,Basic-Latin sdf,Basic-Latin psdf"
python tools/bench_configs.py --reps 6 --only "$ONLY" > gpurun_out/ab_main.jsonl 2> gpurun_out/ab_main.err
for v in "$@"; do
  MSDFGEN_HIP_LIB=$PWD/variants/$v.so MSDFHIP_RES_LDS_BUDGET=${BUDGET:-13312} python tools/bench_configs.py --reps 6 --only "$ONLY" > gpurun_out/ab_$v.jsonl 2> gpurun_out/ab_$v.err
done
python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullsize.py::test_config5_logo_1024_every_texel_and_stencil 2>&1 | tail -12 > gpurun_out/r02_gputests.log
tail -5 gpurun_out/r02_gputests.log
