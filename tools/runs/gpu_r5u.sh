#!/bin/bash
# round 5, GPU session u: host-side SHA-256 with the x86 SHA extensions + whole-block feeding (the commitment transaction's 20 KB row is hashed in front of the launch):
# lamd_check_commitment_signed latency, the template / commitment / served / gossip-latency tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
grep -m1 "model name" /proc/cpuinfo | tee gpurun_out/r5u_commit_probe.txt; grep -o -m1 "sha_ni" /proc/cpuinfo | tee -a gpurun_out/r5u_commit_probe.txt
timeout 300 python tools/commit_trace_probe.py 2>&1 | grep -E "sight|cached" | tee -a gpurun_out/r5u_commit_probe.txt
timeout 300 python tools/commit_trace_probe.py 2>&1 | grep -E "sight|cached" | tee -a gpurun_out/r5u_commit_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_commitment.py tests/test_served.py tests/test_cln_shim.py -m gpu -q -x -k "tx_sig or templates or transactions or commitment or served or native or gossip or shim or mirror" 2>&1 | tail -3 | tee gpurun_out/r5u_tests.txt
