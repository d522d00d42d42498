# per-kernel times of the bench step (or any command): bash tools/kernel_trace.sh <tag> [env assignments...]
REPO=$PWD; export TMPDIR=/tmp; TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/kt_$TAG -o kt -- python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $REPO/gpurun_out/kt_$TAG.log 2>&1
cd $REPO
python tools/rocpd_summary.py $(find gpurun_out/kt_$TAG -name "*.db") 2>&1 | grep -E "^void|^msdfhip|kernel " | cut -c1-60,73-140 | head -14
find gpurun_out/kt_$TAG -name "*.db" -delete
