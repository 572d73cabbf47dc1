#!/bin/bash
# round 4, GPU session f: (1) is the ingest's host logic slower inside a process that holds a GPU context?  (2) streaming queue with 17 staging sets
# (16 flushes in flight) against 9; (3) the full GPU suite + smoke on the build with 24-bit G windows
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
echo "== host bench, plain process"; LAMD_INGEST_PROFILE=1 timeout 300 python tools/ingest_host_bench.py 100000 4 2>&1 | grep -E "sub-batch 1/|host logic" | tail -3
echo "== host bench, process with an engine (GPU context, pinned memory)"; LAMD_INGEST_PROFILE=1 timeout 300 python tools/ingest_host_bench.py 100000 4 --with-engine 2>&1 | grep -E "sub-batch 1/|host logic" | tail -3
echo "== the same with MALLOC_ARENA_MAX=1 / glibc tunables"; MALLOC_ARENA_MAX=1 LAMD_INGEST_PROFILE=1 timeout 300 python tools/ingest_host_bench.py 100000 4 --with-engine 2>&1 | grep -E "sub-batch 1/|host logic" | tail -3
MALLOC_TOP_PAD_=1073741824 MALLOC_TRIM_THRESHOLD_=4294967296 MALLOC_MMAP_THRESHOLD_=4294967296 LAMD_INGEST_PROFILE=1 timeout 300 python tools/ingest_host_bench.py 100000 4 --with-engine 2>&1 | grep -E "sub-batch 1/|host logic" | tail -3
echo "== streaming queue: 9 staging sets (shipped) against 17"
PROBE_INFLIGHT=8,8 PROBE_RESIDENT=4,8,16,100 timeout 300 python tools/stream_probe.py 2>&1 | grep -E "in flight|resident" | tee gpurun_out/r4f_stream_q9.txt
LAMD_LIB_PATH=$PWD/tools/variants/liblightning_amd_q17.so PROBE_INFLIGHT=8,12,16,16 PROBE_RESIDENT=100 timeout 300 python tools/stream_probe.py 2>&1 | grep -E "in flight|resident" | tee gpurun_out/r4f_stream_q17.txt
python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r4f_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
