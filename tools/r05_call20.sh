# Round 5 (session 2), call 20: parity sweep of the final kernels -- 200 000 random shapes in batches (every mode x combiner x correction kind) + 12 000 through single calls.
mkdir -p gpurun_out
timeout 900 python tools/fuzz_parity.py --shapes 200000 --seed 504 > gpurun_out/r05_fuzz_504.json 2> gpurun_out/r05_fuzz_504.err; cat gpurun_out/r05_fuzz_504.json
timeout 300 python tools/fuzz_parity.py --shapes 12000 --seed 505 --single > gpurun_out/r05_fuzz_505.json 2> gpurun_out/r05_fuzz_505.err; cat gpurun_out/r05_fuzz_505.json
