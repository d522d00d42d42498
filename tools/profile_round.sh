# one gpurun call: rocprofv3 kernel statistics + the PMC passes of the bench command (counters never combined with API tracing).
# Usage: bash tools/profile_round.sh <tag>     -> gpurun_out/<tag>_{stats,fetch,write,sq,sq2}/
TAG=${1:-r02}
REPO=$PWD
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/${TAG}_stats -o $TAG -- $CMD > $REPO/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $REPO/gpurun_out/${TAG}_fetch -o $TAG -- $CMD > $REPO/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $REPO/gpurun_out/${TAG}_write -o $TAG -- $CMD > $REPO/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $REPO/gpurun_out/${TAG}_sq -o $TAG -- $CMD > $REPO/gpurun_out/${TAG}_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $REPO/gpurun_out/${TAG}_sq2 -o $TAG -- $CMD > $REPO/gpurun_out/${TAG}_sq2.log 2>&1
cd $REPO
find gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_sq gpurun_out/${TAG}_sq2 -name "*.db" | head
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_stats -name "*.db") > gpurun_out/${TAG}_kernel_stats.txt
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_sq gpurun_out/${TAG}_sq2 -name "*.db") > gpurun_out/${TAG}_pmc.txt
head -20 gpurun_out/${TAG}_kernel_stats.txt
