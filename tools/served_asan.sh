#!/bin/bash
# The service under AddressSanitizer / UBSan on the CPU: lamd_served and liblightning_amd_client.so built with -fsanitize=address,undefined, the CPU part
# of tests/test_served.py (stub engine: every operation, merging, streaming, trust) run against them.  Round 6: clean.
set -eu
cd "$(dirname "$0")/.."
F="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -pthread -Wall -Wno-unknown-pragmas -Wno-unused-function"
g++ $F -o /tmp/lamd_served_asan lightning_amd/csrc/lamd_served.cpp -ldl
g++ $F -fPIC -shared -o /tmp/libclient_asan.so lightning_amd/csrc/lamd_client.cpp
cat > /tmp/run_served_asan.py <<'PY'
import os, sys
root = os.environ["LAMD_ROOT"]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from lightning_amd import _build
real = _build.build_served
_build.build_served = lambda force=False: ("/tmp/lamd_served_asan", "/tmp/libclient_asan.so", real()[2])
import pytest
sys.exit(pytest.main(["-x", "-q", "-m", "not gpu", os.path.join(root, "tests", "test_served.py"), "-p", "no:cacheprovider"]))
PY
LAMD_ROOT=$PWD ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" python /tmp/run_served_asan.py
