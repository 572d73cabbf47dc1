mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -x -k "bolt12 or tx_sig or reference_unit" > gpurun_out/pytest_gpu_r02h.log 2>&1; tail -5 gpurun_out/pytest_gpu_r02h.log
LAMD_PREP_BATCH=64 LAMD_PREP_MIN_THREADS=16384 timeout 300 python tools/diag_mismatch.py > gpurun_out/diag_pb64.txt 2>gpurun_out/diag_pb64.err; grep -c . gpurun_out/diag_pb64.txt; python - <<'PY'
import json
for l in open("gpurun_out/diag_pb64.txt"):
    d=json.loads(l)
    if d["mismatches"]: print({k:d[k] for k in ("waves","cache","call","kind","mismatches","values","classes","first_rows","same_as_prev_call") if k in d})
print("done")
PY
