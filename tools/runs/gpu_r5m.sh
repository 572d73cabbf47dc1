#!/bin/bash
# round 5, GPU session m: traces -- one commitment_signed as one call; a 1/8 shard of the commit storm through the streaming queue; the service probe (in-process
# leg fixed); then the full GPU suite and the bench line on the final tree
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r5m_trace_commit -- python $R/tools/commit_trace_probe.py 2>&1 | grep -E "sight|cached" | tee $R/gpurun_out/r5m_commit_probe.txt)
(cd /tmp && export TMPDIR=/tmp && PROBE_REPS=4 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r5m_trace_storm -- python $R/tools/call_trace_probe.py storm 2>&1 | grep -E "storm" | tee $R/gpurun_out/r5m_storm_probe.txt)
find gpurun_out/r5m_trace_commit gpurun_out/r5m_trace_storm -name "*.csv" | xargs gzip -9
timeout 600 python tools/served_probe.py 2>&1 | grep -E "in-process|service|server" | tee gpurun_out/r5m_served_probe.txt
S=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error|Error|assert|FAILED" | tail -8 | tee gpurun_out/r5m_pytest.log; echo "pytest wall $(( $(date +%s) - S )) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r5m_bench.json 2> gpurun_out/r5m_bench.err; echo "bench.py rc=$? wall $(( $(date +%s) - S )) s"; tail -2 gpurun_out/r5m_bench.err | cut -c1-300
python tools/bench_summary.py gpurun_out/r5m_bench.json
