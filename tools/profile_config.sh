# SQ counters + kernel times of one tools/bench_configs.py configuration: bash tools/profile_config.sh <name-substring> <tag>
REPO=$PWD; export TMPDIR=/tmp
CMD="python $REPO/tools/bench_configs.py --reps 3 --only $1"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $REPO/gpurun_out/pc_$2 -o pc -- $CMD > $REPO/gpurun_out/pc_$2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_BRANCH -d $REPO/gpurun_out/pc2_$2 -o pc -- $CMD > $REPO/gpurun_out/pc2_$2.log 2>&1
cd $REPO
python tools/rocpd_summary.py $(find gpurun_out/pc_$2 gpurun_out/pc2_$2 -name "*.db") > gpurun_out/pc_$2.txt
grep -E "k_distance" gpurun_out/pc_$2.txt | cut -c1-40,73-150
