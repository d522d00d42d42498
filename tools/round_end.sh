# Round-end call, part A in ONE gpurun call: GPU tests, the profile round (bench command + configs 4 / 5), the bench line, the prepare call, host calls.
# Usage: bash tools/round_end.sh <tag> <commit>     (part B = tools/final_measurements.sh)
TAG=${1:-r04}; COMMIT=${2:-?}
mkdir -p gpurun_out
timeout 240 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gputests_final.log 2>&1; tail -3 gpurun_out/${TAG}_gputests_final.log
WITH_CONFIGS=1 timeout 260 bash tools/profile_round.sh $TAG $COMMIT > gpurun_out/${TAG}_profile_round.log 2>&1; tail -2 gpurun_out/${TAG}_profile_round.log | cut -c1-200
timeout 120 python bench.py > gpurun_out/${TAG}_bench_final.json 2> gpurun_out/${TAG}_bench_final.err; tail -1 gpurun_out/${TAG}_bench_final.json | cut -c1-300
timeout 60 python tools/bench_configs.py --only prep > gpurun_out/${TAG}_prep.jsonl 2> gpurun_out/${TAG}_prep.err; cat gpurun_out/${TAG}_prep.jsonl | cut -c1-400
timeout 90 python tools/host_call_latency.py --threads 1,4,64 --leaders 4 > gpurun_out/${TAG}_host_calls_final.jsonl 2> gpurun_out/${TAG}_host_calls_final.err; cut -c1-420 gpurun_out/${TAG}_host_calls_final.jsonl
