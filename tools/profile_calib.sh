# One gpurun call: instruction-class calibration (tools/valu_calib) + class-mix and busy counters of the bench step and of configs 4 / 5.
# Usage: bash tools/profile_calib.sh <tag>    -> gpurun_out/<tag>_calib*.{jsonl,txt}
# Counters are collected in their own passes with --kernel-trace only (never with API tracing).
TAG=${1:-r03}
REPO=$PWD; export TMPDIR=/tmp
P1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64"
P2="SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU"
P3="SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P4="SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
$REPO/tools/valu_calib 1 2 4 8 > gpurun_out/${TAG}_calib.jsonl 2> gpurun_out/${TAG}_calib.err
$REPO/tools/valu_calib --fetch > gpurun_out/${TAG}_calib_fetch.jsonl 2>> gpurun_out/${TAG}_calib.err
cd /tmp
pass() {   # pass <dir-tag> "<counters>" <command...>
    local d=$1 c=$2; shift 2
    rocprofv3 --kernel-trace --pmc $c -d $REPO/gpurun_out/${TAG}_$d -o $d -- "$@" > $REPO/gpurun_out/${TAG}_$d.log 2>&1
}
pass calib_p1 "$P1" $REPO/tools/valu_calib 4
pass calib_p2 "$P2" $REPO/tools/valu_calib 4
pass calib_fetch "FETCH_SIZE" $REPO/tools/valu_calib --fetch
pass calib_write "WRITE_SIZE" $REPO/tools/valu_calib --fetch
BENCH="python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
for p in 1 2 3 4; do eval c=\$P$p; pass bench_p$p "$c" $BENCH; done
if [ -z "$SKIP_CONFIGS" ]; then
for cfgname in "cfg4: 8192 CJK" "cfg5"; do
    short=$(echo $cfgname | cut -c1-4)
    for p in 1 2 3 4; do eval c=\$P$p; pass ${short}_p$p "$c" python $REPO/tools/bench_configs.py --reps 2 --only "$cfgname"; done
    rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/${TAG}_${short}_stats -o st -- python $REPO/tools/bench_configs.py --reps 3 --only "$cfgname" > $REPO/gpurun_out/${TAG}_${short}_stats.log 2>&1
done
fi
cd $REPO
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_calib_p1 gpurun_out/${TAG}_calib_p2 gpurun_out/${TAG}_calib_fetch gpurun_out/${TAG}_calib_write -name "*.db") > gpurun_out/${TAG}_calib_counters.txt 2>&1
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_bench_p1 gpurun_out/${TAG}_bench_p2 gpurun_out/${TAG}_bench_p3 gpurun_out/${TAG}_bench_p4 -name "*.db") > gpurun_out/${TAG}_bench_classmix.txt 2>&1
if [ -z "$SKIP_CONFIGS" ]; then
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_cfg4_p1 gpurun_out/${TAG}_cfg4_p2 gpurun_out/${TAG}_cfg4_p3 gpurun_out/${TAG}_cfg4_p4 gpurun_out/${TAG}_cfg4_stats gpurun_out/${TAG}_cfg5_p1 gpurun_out/${TAG}_cfg5_p2 gpurun_out/${TAG}_cfg5_p3 gpurun_out/${TAG}_cfg5_p4 gpurun_out/${TAG}_cfg5_stats -name "*.db") > gpurun_out/${TAG}_config_counters.txt 2>&1
fi
# the .db files are large: keep the summaries only
find gpurun_out -name "*.db" -path "*${TAG}_*" -size +20M -delete
head -5 gpurun_out/${TAG}_calib.jsonl
