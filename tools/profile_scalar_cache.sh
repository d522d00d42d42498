# scalar-data-cache behaviour of the bench step: bash tools/profile_scalar_cache.sh  -> gpurun_out/sc_*.txt
REPO=$PWD; export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "SQC_DCACHE|SQ_INST_CYCLES_SMEM|SQC_TC_|SQ_WAIT_INST_LDS|SQ_INSTS_SMEM" | head -60 > $REPO/gpurun_out/sc_avail.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_ANY -d $REPO/gpurun_out/sc_a -o sc -- $CMD > $REPO/gpurun_out/sc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_TC_REQ SQC_TC_DATA_READ_REQ SQC_DCACHE_REQ_READ_1 SQC_DCACHE_REQ_READ_2 SQC_DCACHE_REQ_READ_4 SQC_DCACHE_REQ_READ_8 SQC_DCACHE_REQ_READ_16 -d $REPO/gpurun_out/sc_b -o sc -- $CMD > $REPO/gpurun_out/sc_b.log 2>&1
cd $REPO
python tools/rocpd_summary.py $(find gpurun_out/sc_a gpurun_out/sc_b -name "*.db") > gpurun_out/sc_counters.txt 2>&1
grep -E "k_distance" gpurun_out/sc_counters.txt | cut -c1-44,73-150
