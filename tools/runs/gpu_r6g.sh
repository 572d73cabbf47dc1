#!/bin/bash
# round 6, session g: both additions of a comb column as straight-line code (-DLAMD_UNROLL_HALF=1) against the shipped loop: timing (bench.py --ab,
# alternating) and executed VALU instructions per verification (one SQ counter pass each)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6g
export GPU_MAX_HW_QUEUES=16
one() {  # label lib
  unset LAMD_LIB_PATH; [ "$2" != base ] && export LAMD_LIB_PATH=$R/tools/variants/liblightning_amd_$2.so
  timeout 300 python bench.py --ab --cpu-sample 0 --details gpurun_out/r6g/$1.json > gpurun_out/r6g/$1.line 2> gpurun_out/r6g/$1.err
  python - "$1" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6g/%s.json" % sys.argv[1])); r = d["roofline"]
print("%-8s cold %.1f M/s step %.3f ms | chained launch %.3f ms frac %.3f | isolated %.3f ms | peak %.2f T | mismatches %d" % (
    sys.argv[1], d["value"] / 1e6, d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["isolated"]["launch_ms"], r["peak"], d["parity"]["mismatches"]))
PY
}
for rep in 1 2 3; do one base$rep base; one uh$rep uh; done 2>&1 | tee gpurun_out/r6g/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in base uh; do
  unset LAMD_LIB_PATH; [ $v != base ] && export LAMD_LIB_PATH=$R/tools/variants/liblightning_amd_$v.so
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r6g/${v}_pmc -- python $R/bench.py --roofline-only --steps 4 --warmup 1 --cpu-sample 0 --details $R/gpurun_out/r6g/pmc_$v.json > /dev/null 2> $R/gpurun_out/r6g/${v}_pmc.err
done
cd $R
python - <<'PY' | tee gpurun_out/r6g/valu.txt
import csv, glob, collections
for v in ("base", "uh"):
    f = glob.glob("gpurun_out/r6g/%s_pmc/**/*_counter_collection.csv" % v, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        if "k_ecmult_keyed<false" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 500000:
            acc[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    per = [a["SQ_INSTS_VALU"] / a["SQ_WAVES"] for a in acc.values() if a.get("SQ_WAVES")]
    print("%-5s VALU instructions per verification: %.0f (over %d launches)" % (v, sum(per) / len(per), len(per)))
PY
find gpurun_out/r6g -name "*.csv" -size +2M -delete
