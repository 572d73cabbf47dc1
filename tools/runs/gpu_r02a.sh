set -x
mkdir -p gpurun_out
for q in 4 16 32; do GPU_MAX_HW_QUEUES=$q timeout 300 python tools/repro_gen_race.py 20 2>&1 | tail -1; done > gpurun_out/repro_race.txt
cat gpurun_out/repro_race.txt
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_r02a.log 2>&1; tail -5 gpurun_out/pytest_gpu_r02a.log
tools/queue_sweep.sh 4 1
timeout 600 python bench.py > gpurun_out/bench_r02a.json 2> gpurun_out/bench_r02a.err; tail -c 600 gpurun_out/bench_r02a.json
