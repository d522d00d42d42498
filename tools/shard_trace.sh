# bash tools/shard_trace.sh <tag> <set> <parts> <rank>: per-kernel durations of one config-4 shard's step
REPO=$PWD; export TMPDIR=/tmp; TAG=$1; shift
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/st_$TAG -o st -- python $REPO/tools/shard_trace.py "$@" > $REPO/gpurun_out/st_$TAG.log 2>&1
cd $REPO
tail -1 gpurun_out/st_$TAG.log
python tools/rocpd_summary.py $(find gpurun_out/st_$TAG -name "*.db") 2>&1 | grep -E "^void|^msdfhip|kernel " | cut -c1-60,73-140 | head -14
find gpurun_out/st_$TAG -name "*.db" -delete
