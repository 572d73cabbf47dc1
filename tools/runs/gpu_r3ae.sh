#!/bin/bash
# round 3, GPU session ae: the build after the move of the shared per-row front ends into verify_core.h -- gossip / check_tx_sig / shim / small-path tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export GPU_MAX_HW_QUEUES=16
python -m pytest tests/test_gpu_parity.py tests/test_cln_shim.py tests/test_gossip_ingest.py -m gpu -x -q -k "gossip or tx_sig or shim or golden or reference_held or small or cfg4 or ingest or store" 2>&1 | tail -4
