#!/bin/bash
# round 6, session aa: the cold rows' ladder started right behind the key classification (k_partition_cold, LAMD_EARLY_COLD=1, the new default) against the
# list made by k_partition after the table kernels (=0): parity first, then the bench's strong-scaling sweep of configs[3] alternating on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6aa
export GPU_MAX_HW_QUEUES=16
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gossip_ingest.py tests/test_gpu_commitment.py -m gpu -q -x 2>&1 | tail -2 | tee gpurun_out/r6aa/parity.txt
for rep in 1 2 3; do
  for ec in 1 0; do
    LAMD_EARLY_COLD=$ec timeout 300 python bench.py --cpu-sample 0 --no-h2h --details gpurun_out/r6aa/d_${ec}_$rep.json > gpurun_out/r6aa/l_${ec}_$rep.json 2> gpurun_out/r6aa/e_${ec}_$rep.err
    python - $ec $rep <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6aa/d_%s_%s.json" % (sys.argv[1], sys.argv[2])))
c = d["strong_scaling_1gpu"]["cfg4_gossip_replay"]
print("early_cold=%s  value %.1f M/s  cfg4 T1 %.2f ms  W=4 %.2f  W=8 %.2f ms  predicted x%.2f (one cut x%.2f)  mismatches %d %d" % (
    sys.argv[1], d["value"] / 1e6, c["1"]["slowest_ms"], c["4"]["slowest_ms"], c["8"]["slowest_ms"], c["predicted_speedup_8"], c["one_cut"]["predicted_speedup_8"],
    d["parity"]["mismatches"], c["mismatches"]))
PY
  done
done 2>&1 | tee gpurun_out/r6aa/ab.txt
