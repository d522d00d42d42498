# one gpurun call: parity tests, the bench line, smoke
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r02_gputests.log
tail -6 gpurun_out/r02_gputests.log
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
tail -c 600 gpurun_out/r02_bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1; tail -2 gpurun_out/r02_smoke.log
