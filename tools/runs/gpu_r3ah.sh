#!/bin/bash
# round 3, GPU session ah: small_path(): more than one block of the one-launch path only with a key-table cache -- small-path tests, a commitment on a cache-off engine
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export GPU_MAX_HW_QUEUES=16
python -m pytest tests/test_gpu_parity.py tests/test_cln_shim.py -m gpu -x -q -k "small or learn or boundaries or veneers or keyed or shim or reference_held" 2>&1 | tail -3
LAMD_CACHE=0 python - <<'PY' 2>&1 | grep -v "^W\|amdgpu.ids"
import time, numpy as np
from lightning_amd import Engine, workload
with Engine(0) as e:
    st = workload.make_commit_storm(e, 2, device="cuda:0")["ecdsa"]
    hh, ss, pp = [np.ascontiguousarray(x[:484]) for x in st.cols]
    for _ in range(3):
        assert (e.verify_ecdsa(hh, ss, pp) == st.expect[:484]).all()
    ts = []
    for _ in range(50):
        t = time.perf_counter(); e.verify_ecdsa(hh, ss, pp); ts.append(time.perf_counter() - t)
    print("cache-off engine, one commitment (general path, a per-call table): p50 %.3f ms, new tables %d" % (sorted(ts)[25] * 1e3, e.info()["last_new_tables"]))
    assert (e.verify_ecdsa(hh[:1], ss[:1], pp[:1]) == st.expect[:1]).all()
PY
