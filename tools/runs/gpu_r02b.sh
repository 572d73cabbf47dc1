set -x
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_r02b.log 2>&1; tail -15 gpurun_out/pytest_gpu_r02b.log
timeout 900 python bench.py > gpurun_out/bench_r02b.json 2> gpurun_out/bench_r02b.err; tail -c 1500 gpurun_out/bench_r02b.json; tail -5 gpurun_out/bench_r02b.err
LAMD_KEYED_WAVES=3 timeout 600 python bench.py --skip-extra --cpu-sample 0 > gpurun_out/bench_r02b_w3.json 2> gpurun_out/bench_r02b_w3.err; head -c 400 gpurun_out/bench_r02b_w3.json
bash tools/pmc_run.sh r02b > gpurun_out/pmc_r02b.log 2>&1; tail -3 gpurun_out/pmc_r02b.log
