# per-kernel times of the bench step for a build variant: bash tools/profile_variant.sh <variant.so|main> <tag>
REPO=$PWD; export TMPDIR=/tmp
if [ "$1" != main ]; then export MSDFGEN_HIP_LIB=$REPO/variants/$1.so; fi
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/pv_$2 -o pv -- python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $REPO/gpurun_out/pv_$2.log 2>&1
cd $REPO
python tools/rocpd_summary.py $(find gpurun_out/pv_$2 -name "*.db") | head -12 | cut -c1-60,73-130
