mkdir -p gpurun_out
for v in base v_cs v_ms v_cs_ms v_cs_ms_al v_cs_ms_il; do echo "== $v"; timeout 120 tools/variants/fe_bench_$v; done > gpurun_out/fe_bench.txt 2>&1
timeout 200 tools/variants/fe_bench_base ops >> gpurun_out/fe_bench.txt 2>&1
cat gpurun_out/fe_bench.txt
