#!/bin/bash
# round 6, session c: counters of the build with the XYZZ G run (tools/pmc_run.sh: trace + stats, FETCH, WRITE, two SQ groups), their summary,
# the GPU suite, and the bench lines (the driver's command and the default run)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6c
export GPU_MAX_HW_QUEUES=16
(cd tools && [ -x gather_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o gather_calib gather_calib.hip) 2>&1 | tail -2
bash tools/pmc_run.sh r06 > gpurun_out/r6c/pmc_run.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r06 gpurun_out/r6c/r06 > /dev/null 2> gpurun_out/r6c/pmc_summary.err; tail -3 gpurun_out/r6c/pmc_summary.err
cp gpurun_out/pmc_r06/trace/runc/*_kernel_stats.csv gpurun_out/r6c/r06_roofline_only_kernel_stats.csv 2>/dev/null
cp gpurun_out/pmc_r06/trace.json gpurun_out/r6c/r06_roofline_only_bench_under_rocprof.json 2>/dev/null
find gpurun_out/pmc_r06 -name "*.csv" -size +2M -delete; find gpurun_out/pmc_r06 -name "*.db" -delete
cp gpurun_out/r6c/r06_pmc_latest.json profiles/pmc_latest.json
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r6c/bench20.json 2> gpurun_out/r6c/bench20.err; cp bench_details.json gpurun_out/r6c/details20.json
( time python bench.py ) > gpurun_out/r6c/bench250.json 2> gpurun_out/r6c/bench250.err; cp bench_details.json gpurun_out/r6c/details250.json
grep real gpurun_out/r6c/bench20.err gpurun_out/r6c/bench250.err
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r6c/pytest.txt
cat gpurun_out/r6c/bench20.json; sed -n '/VALU wave-instructions per STEP/,/^$/p' gpurun_out/r6c/r06_pmc_summary.txt; sed -n '/derived:/,/^$/p' gpurun_out/r6c/r06_pmc_summary.txt
du -sh gpurun_out
