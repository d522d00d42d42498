# one gpurun call at the end of a work block: parity tests, bench line, smoke, profiles (kernel stats + PMC passes), configs, host-call latency
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r02_gputests.log
tail -3 gpurun_out/r02_gputests.log
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1; tail -1 gpurun_out/r02_smoke.log
bash tools/profile_round.sh r02 > gpurun_out/r02_profile.log 2>&1
python tools/bench_configs.py --reps 8 > gpurun_out/r02_configs.jsonl 2> gpurun_out/r02_configs.err
python tools/host_call_latency.py --threads 1,4,16,64 > gpurun_out/r02_host_calls.jsonl 2> gpurun_out/r02_host_calls.err
