#!/bin/bash
# round 5, GPU session l: the service with merged commitment validations (tests), check_tx_sig batches hashing on the device, the service under load
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 900 python -X faulthandler -m pytest tests/test_served.py tests/test_gpu_commitment.py tests/test_cln_shim.py "tests/test_gpu_parity.py::test_reference_held_transactions_bolt3_htlc_and_second_grind_kat" -m gpu -v -x > gpurun_out/r5l_tests.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|error|PASSED|FAILED|Fatal|fault|Abort|assert" gpurun_out/r5l_tests.log | tail -24
timeout 900 python tools/served_probe.py 2>&1 | grep -E "in-process|service|server" | tee gpurun_out/r5l_served_probe.txt
