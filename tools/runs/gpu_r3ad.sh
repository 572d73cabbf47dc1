#!/bin/bash
# round 3, GPU session ad: the new reference-held goldens (checkmessage vectors, BOLT #3 transactions, second grind KAT) on the device and through the mirror
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export GPU_MAX_HW_QUEUES=16
python -m pytest tests/test_gpu_parity.py tests/test_cln_shim.py -m gpu -x -q -k "reference_held or recover or golden or reference_unit or shim or grind" 2>&1 | tail -4
