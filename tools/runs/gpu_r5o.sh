#!/bin/bash
# round 5, GPU session o: lamd_check_commitment_signed after the second fix (rows with long output lists hashed on the host while the blob is packed): trace + latency; commitment tests; bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 300 python tools/commit_trace_probe.py 2>&1 | grep -E "sight|cached" | tee gpurun_out/r5o_commit_probe.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r5o_trace_commit -- python $R/tools/commit_trace_probe.py 2>&1 | grep -E "sight|cached" | tee -a $R/gpurun_out/r5o_commit_probe.txt)
find gpurun_out/r5o_trace_commit -name "*.csv" | xargs gzip -9
timeout 900 python -m pytest tests/test_gpu_commitment.py tests/test_served.py "tests/test_gpu_parity.py::test_reference_held_transactions_bolt3_htlc_and_second_grind_kat" -m gpu -q -x 2>&1 | tail -3
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r5o_bench.json 2> gpurun_out/r5o_bench.err; echo "bench.py rc=$? wall $(( $(date +%s) - S )) s"; tail -2 gpurun_out/r5o_bench.err | cut -c1-300
python tools/bench_summary.py gpurun_out/r5o_bench.json | grep -E "^value|latency|config h2h"
