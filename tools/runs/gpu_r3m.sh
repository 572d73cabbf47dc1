#!/bin/bash
# round 3, GPU session m: the fused front end without the device-scope fences -- keyed / cache / chunk parity tests, A/B in the cold loop
# against LAMD_FUSED_FRONT=0, lanes 4 / 5 / 6 with two streams per lane (LAMD_MERGE_SIDE=1)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "keyed or cache or chunk or ragged or repeated or lanes or cfg4_gossip_replay_small" 2>&1 | tail -4 > gpurun_out/r3m_pytest.log
tail -3 gpurun_out/r3m_pytest.log
line() {
  python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; k=d['rates']['kernel_ms_ecdsa_isolated']
print('$2: cold %.1f M/s, step %.2f ms, launch in the loop %.3f ms, isolated %.3f ms, isolated front %.2f tables %.2f, mismatches %d' % (d['value']/1e6, d['ms_per_step'], r['avg_launch_ms'], r['isolated']['launch_ms'], k['prep'], k['keys_and_tables'], d['parity']['mismatches']))"
}
for v in 1 0 1 0; do
  LAMD_FUSED_FRONT=$v timeout 300 python bench.py --roofline-only > gpurun_out/r3m_f$v.json 2> gpurun_out/r3m_f$v.err
  line gpurun_out/r3m_f$v.json "LAMD_FUSED_FRONT=$v"
done | tee gpurun_out/r3m_fused.txt
for l in 4 5 6; do
  LAMD_MERGE_SIDE=1 LAMD_LANES=$l timeout 300 python bench.py --roofline-only > gpurun_out/r3m_m$l.json 2> gpurun_out/r3m_m$l.err
  line gpurun_out/r3m_m$l.json "LAMD_MERGE_SIDE=1 LAMD_LANES=$l"
done | tee -a gpurun_out/r3m_fused.txt
