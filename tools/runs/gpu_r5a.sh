#!/bin/bash
# round 5, GPU session a: the new parity tests (mirror arguments, commitment_signed as one call, lamd_multi), the multi H2D A/B, the full bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
S=$(date +%s); timeout 900 python -m pytest tests/test_gpu_commitment.py tests/test_cln_shim.py tests/test_gpu_multi.py tests/test_abi.py -m gpu -q -x 2>&1 | tail -25 | tee gpurun_out/r5a_pytest.log; echo "pytest wall $(( $(date +%s) - S )) s"
timeout 300 python tools/multi_h2d_probe.py 2>&1 | tail -2 | tee gpurun_out/r5a_multi_h2d.txt
LAMD_MULTI_PINNED=0 timeout 300 python tools/multi_h2d_probe.py 2>&1 | tail -2 | tee -a gpurun_out/r5a_multi_h2d.txt
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err; echo "bench.py rc=$? wall $(( $(date +%s) - S )) s"; tail -5 gpurun_out/r5a_bench.err
python tools/bench_summary.py gpurun_out/r5a_bench.json
