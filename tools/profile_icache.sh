# instruction-cache behaviour of the bench step per kernel: bash tools/profile_icache.sh -> gpurun_out/ic_counters.txt
REPO=$PWD; export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|SQ_IFETCH|SQ_WAIT_IFETCH|INST_FETCH|SQ_INST_LEVEL" | head -40 > $REPO/gpurun_out/ic_avail.txt
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_IFETCH -d $REPO/gpurun_out/ic_a -o ic -- $CMD > $REPO/gpurun_out/ic_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_MISSES_DUPLICATE SQ_WAIT_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES -d $REPO/gpurun_out/ic_b -o ic -- $CMD > $REPO/gpurun_out/ic_b.log 2>&1
cd $REPO
python tools/rocpd_summary.py $(find gpurun_out/ic_a gpurun_out/ic_b -name "*.db") > gpurun_out/ic_counters.txt 2>&1
grep -E "k_ec_query|k_distance|k_ec_fast|counter|kernel" gpurun_out/ic_counters.txt | cut -c1-200
tail -3 gpurun_out/ic_a.log gpurun_out/ic_b.log
find gpurun_out/ic_a gpurun_out/ic_b -name "*.db" -delete
cat gpurun_out/ic_avail.txt | cut -c1-150 | head -30
