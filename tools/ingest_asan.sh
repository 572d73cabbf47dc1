#!/bin/bash
# Host-side C++ (cln_shim.cpp, gossip_ingest.cpp) under AddressSanitizer + UBSan: builds an instrumented copy of liblightning_amd_cln.so,
# runs the CPU tests of the mirror and of the batched gossip ingest plus a 20 k-channel flood through it, then puts the normal build back.
# The ingest's verdict maps hold keys that refer to message bytes they do not own: this is the check of those lifetimes.  No GPU needed.
set -eu
cd "$(dirname "$0")/.."
SO=lightning_amd/liblightning_amd_cln.so
cp $SO /tmp/lamd_cln_plain.so
trap 'cp /tmp/lamd_cln_plain.so $SO; touch $SO' EXIT
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -fPIC -shared -Wall -Wno-unknown-pragmas -Wno-unused-function \
  -o $SO lightning_amd/csrc/cln_shim.cpp lightning_amd/csrc/gossip_ingest.cpp -Llightning_amd -llightning_amd -Wl,-rpath,"$PWD/lightning_amd"
touch $SO
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
python -m pytest tests/test_gossip_ingest.py tests/test_cln_shim.py -x -q -m "not gpu"
python tools/ingest_host_bench.py 20000 4
