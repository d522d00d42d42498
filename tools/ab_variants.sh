# A/B of build variants (variants/<name>.so) on selected configs: bash tools/ab_variants.sh "<only>" name1 name2 ...
ONLY=$1; shift
python tools/bench_configs.py --reps 6 --only "$ONLY" > gpurun_out/abv_main.jsonl 2> gpurun_out/abv_main.err
for v in "$@"; do
  MSDFGEN_HIP_LIB=$PWD/variants/$v.so python tools/bench_configs.py --reps 6 --only "$ONLY" > gpurun_out/abv_$v.jsonl 2> gpurun_out/abv_$v.err
done
