#!/bin/bash
# tools/build_variant.sh NAME FLAGS...: the engine built with extra compile flags into tools/variants/liblightning_amd_NAME.so
# (run the bench against it with LAMD_LIB_PATH; the shipped library and its stamps are not touched)
set -eu
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -Wno-unknown-pragmas -mllvm -amdgpu-codegenprepare-mul24=false"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o tools/variants/engine_$name.o lightning_amd/csrc/lamd_engine.hip
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -pthread -c -o tools/variants/multi_$name.o lightning_amd/csrc/lamd_multi.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o tools/variants/liblightning_amd_$name.so tools/variants/engine_$name.o tools/variants/multi_$name.o
rm -f tools/variants/engine_$name.o tools/variants/multi_$name.o
ls -la tools/variants/liblightning_amd_$name.so
