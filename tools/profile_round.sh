# One gpurun call: every profile the bench line and DESIGN.md quote, for the bench command (counters in their own passes, --kernel-trace only).
# Usage: bash tools/profile_round.sh <tag> [commit]     -> gpurun_out/<tag>_*  and  profiles/<tag>_kernel_stats.txt, <tag>_pmc_bench.json, pmc_traffic.json
TAG=${1:-r04}; COMMIT=${2:-?}
REPO=$PWD; export TMPDIR=/tmp
P1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64"
P2="SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU"
P3="SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P4="SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
BENCH="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
cd /tmp
pass() { local d=$1 c=$2; shift 2; rocprofv3 --kernel-trace --pmc $c -d $REPO/gpurun_out/${TAG}_$d -o $d -- "$@" > $REPO/gpurun_out/${TAG}_$d.log 2>&1; }
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/${TAG}_stats -o $TAG -- $BENCH > $REPO/gpurun_out/${TAG}_stats.log 2>&1
pass fetch "FETCH_SIZE" $BENCH
pass write "WRITE_SIZE" $BENCH
for p in 1 2 3 4; do eval c=\$P$p; pass bench_p$p "$c" $BENCH; done
if [ -n "$WITH_CONFIGS" ]; then
for cfgname in "cfg4: 8192 CJK" "cfg5"; do
    short=$(echo $cfgname | cut -c1-4)
    for p in 1 2 3 4; do eval c=\$P$p; pass ${short}_p$p "$c" python $REPO/tools/bench_configs.py --reps 2 --only "$cfgname"; done
done
fi
cd $REPO
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_stats -name "*.db") > profiles/${TAG}_kernel_stats.txt
python tools/pmc_report.py $TAG bench > gpurun_out/${TAG}_pmc_bench.txt
if [ -n "$WITH_CONFIGS" ]; then python tools/pmc_report.py $TAG cfg4 k_distance k_ec >> gpurun_out/${TAG}_pmc_bench.txt; python tools/pmc_report.py $TAG cfg5 k_distance k_ec >> gpurun_out/${TAG}_pmc_bench.txt; fi
python tools/pmc_traffic.py $TAG $COMMIT
cat gpurun_out/${TAG}_pmc_bench.txt | head -30
find gpurun_out -name "*.db" -path "*${TAG}_*" -size +8M -delete
mkdir -p gpurun_out/${TAG}_profiles_out; cp profiles/${TAG}_kernel_stats.txt profiles/${TAG}_pmc_*.json profiles/pmc_traffic.json gpurun_out/${TAG}_profiles_out/ 2>/dev/null   # (only gpurun_out/ travels back)
