# SQ counters of the bench step for one build (main or variants/<name>.so), kernels matching a pattern: bash tools/pmc_kernel.sh <name|main> <tag> [grep-pattern]
REPO=$PWD; export TMPDIR=/tmp
if [ "$1" != main ]; then export MSDFGEN_HIP_LIB=$REPO/variants/$1.so; fi
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $REPO/gpurun_out/pk_$2_a -o pk -- $CMD > $REPO/gpurun_out/pk_$2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH -d $REPO/gpurun_out/pk_$2_b -o pk -- $CMD >> $REPO/gpurun_out/pk_$2.log 2>&1
cd $REPO
python tools/rocpd_summary.py $(find gpurun_out/pk_$2_a gpurun_out/pk_$2_b -name "*.db") > gpurun_out/pk_$2.txt
grep -E "${3:-k_ec_fast}" gpurun_out/pk_$2.txt | cut -c1-40,73-150
find gpurun_out/pk_$2_a gpurun_out/pk_$2_b -name "*.db" -delete
