set -x
mkdir -p gpurun_out
timeout 300 python tools/diag_txsig.py > gpurun_out/diag_txsig.txt 2>&1; tail -25 gpurun_out/diag_txsig.txt
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_cln_shim.py::test_reference_unit_test_expectations > gpurun_out/pytest_gpu_r02d.log 2>&1; tail -25 gpurun_out/pytest_gpu_r02d.log
timeout 900 python bench.py > gpurun_out/bench_r02d.json 2> gpurun_out/bench_r02d.err; tail -3 gpurun_out/bench_r02d.err; cut -c1-1500 gpurun_out/bench_r02d.json
