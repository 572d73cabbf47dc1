#!/bin/bash
# round 6, session j: block-aggregated list-builder counters (block_alloc4) -- the GPU suite, the cold shard probe (host wall per call), the bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6j
export GPU_MAX_HW_QUEUES=16
S=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -6 | tee gpurun_out/r6j/pytest.txt; echo "pytest wall $(( $(date +%s) - S )) s"
LAMD_CACHE=0 PROBE_REPS=5 timeout 300 python tools/call_trace_probe.py gossip 2>&1 | tail -4 | tee gpurun_out/r6j/probe_cold.txt
PROBE_REPS=5 timeout 300 python tools/call_trace_probe.py gossip 2>&1 | tail -4 | tee gpurun_out/r6j/probe_warm.txt
cd /tmp && export TMPDIR=/tmp
LAMD_CACHE=0 PROBE_REPS=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r6j/trace -- python $R/tools/call_trace_probe.py gossip > /dev/null 2> $R/gpurun_out/r6j/trace.err
cd $R
F=$(find gpurun_out/r6j/trace -name "*_kernel_trace.csv" | head -1)
python tools/trace_calls.py $F 1 2 2>&1 | grep -E "k_partition|k_cache_lookup|k_dedupe|k_group|after marker" | tee gpurun_out/r6j/builders.txt
find gpurun_out/r6j -name "*.csv" -delete
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r6j/bench20.json 2> gpurun_out/r6j/bench20.err; cp bench_details.json gpurun_out/r6j/details20.json
grep real gpurun_out/r6j/bench20.err; cut -c1-900 gpurun_out/r6j/bench20.json
