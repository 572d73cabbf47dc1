#!/bin/bash
# round 3, GPU session f: tests; occupancy of the dominant kernel in the pipelined loop (LAMD_KEYED_WAVES 3 vs 4); ingest flood
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r3f_pytest.log
tail -2 gpurun_out/r3f_pytest.log
for w in 3 4 3 4; do
  LAMD_KEYED_WAVES=$w timeout 300 python bench.py --roofline-only > gpurun_out/r3f_w$w.json 2> /dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r3f_w$w.json').read().strip().splitlines()[-1]); r=d['roofline']
print('LAMD_KEYED_WAVES=$w: cold %.1f M/s, launch in the loop %.3f ms, isolated %.3f ms, mismatches %d' % (d['value']/1e6, r['avg_launch_ms'], r['isolated']['launch_ms'], d['parity']['mismatches']))"
done | tee gpurun_out/r3f_waves.txt
timeout 300 python tools/ingest_host_bench.py 2>&1 | tail -1 | tee gpurun_out/r3f_ingest_host.txt
