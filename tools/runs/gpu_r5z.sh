#!/bin/bash
# round 5, GPU session z: whole waves of degenerate rows through the ladder's hot form (new test)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "degenerate_rows or native" 2>&1 | tail -5 | tee gpurun_out/r5z_tests.txt
