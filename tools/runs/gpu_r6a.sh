#!/bin/bash
# round 6, session a: the re-cut bench line under the driver's own command, then the default run, then the GPU suite
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r6a_bench20.json 2> gpurun_out/r6a_bench20.err
cp bench_details.json gpurun_out/r6a_details20.json
wc -c gpurun_out/r6a_bench20.json
( time python bench.py ) > gpurun_out/r6a_bench250.json 2> gpurun_out/r6a_bench250.err
cp bench_details.json gpurun_out/r6a_details250.json
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r6a_pytest.txt
cat gpurun_out/r6a_pytest.txt
grep real gpurun_out/r6a_bench20.err gpurun_out/r6a_bench250.err
cat gpurun_out/r6a_bench20.json
