#!/bin/bash
# Host-side C++ (gossip_ingest.cpp) under ThreadSanitizer: the ingest's parallel passes (planning, update runs, announcement runs, sharded txout
# replies, the three-stage pipeline of a drained queue) share maps by construction rules -- a worker owns whole shards, the planning stage reads
# only what no apply pass writes.  This is the check of those rules.  No GPU needed.
set -eu
cd "$(dirname "$0")/.."
SO=lightning_amd/liblightning_amd_cln.so
cp $SO /tmp/lamd_cln_plain.so
trap 'cp /tmp/lamd_cln_plain.so $SO; touch $SO' EXIT
g++ -O1 -g -fsanitize=thread -fno-omit-frame-pointer -std=c++17 -fPIC -shared -Wall -Wno-unknown-pragmas -Wno-unused-function \
  -o $SO lightning_amd/csrc/cln_shim.cpp lightning_amd/csrc/gossip_ingest.cpp -Llightning_amd -llightning_amd -Wl,-rpath,"$PWD/lightning_amd"
touch $SO
export TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:exitcode=0:log_path=/tmp/lamd_tsan"
rm -f /tmp/lamd_tsan.*
export LD_PRELOAD="$(gcc -print-file-name=libtsan.so)"
# (the Python test suite under ThreadSanitizer takes tens of minutes on a small VM: the flood below drives every parallel pass)
LAMD_INGEST_SUB=1500 LAMD_INGEST_RUN_MIN=64 LAMD_INGEST_THREADS=4 timeout 1200 python tools/ingest_host_bench.py 6000 4
unset LD_PRELOAD
if ls /tmp/lamd_tsan.* >/dev/null 2>&1; then echo "ThreadSanitizer reports:"; grep -h "WARNING: ThreadSanitizer" /tmp/lamd_tsan.* | sort | uniq -c; grep -h -A 12 "WARNING: ThreadSanitizer: data race" /tmp/lamd_tsan.* | grep -E "gossip_ingest|#0|#1" | head -40; exit 1; fi
echo "ThreadSanitizer: no reports"
