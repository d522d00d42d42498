# Round 5, GPU call 1: what the box's PCIe link delivers, where the host-output pipeline's time goes, and the LDS-class sweep on the CJK-like set.
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
( nproc; lscpu | grep -i "numa\|model name\|socket" ; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -4; taskset -p $$ ) > gpurun_out/r05_box.txt 2>&1
timeout 150 tools/pcie_probe > gpurun_out/r05_pcie_probe.jsonl 2> gpurun_out/r05_pcie_probe.err; tail -3 gpurun_out/r05_pcie_probe.jsonl
MSDFHIP_PIPELINE_TRACE=1 timeout 120 python tools/pipeline_chunks.py 0 > gpurun_out/r05_pipe_default.jsonl 2> gpurun_out/r05_pipe_default.trace; cat gpurun_out/r05_pipe_default.jsonl
timeout 120 python tools/pipeline_chunks.py 0 1024 2048 4096 8192 > gpurun_out/r05_pipe_chunks.jsonl 2> gpurun_out/r05_pipe_chunks.err; cat gpurun_out/r05_pipe_chunks.jsonl
timeout 300 python tools/lds_class_sweep.py --reps 4 > gpurun_out/r05_lds_sweep.jsonl 2> gpurun_out/r05_lds_sweep.err; wc -l gpurun_out/r05_lds_sweep.jsonl; tail -2 gpurun_out/r05_lds_sweep.err
