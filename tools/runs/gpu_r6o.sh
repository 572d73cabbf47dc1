#!/bin/bash
# round 6, session o: executed VALU instructions of the table kernels, chains against tree (one SQ counter pass each over bench.py --roofline-only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6o
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
for v in chains tree; do
  [ $v = tree ] && export LAMD_KC_TREE=1 || export LAMD_KC_TREE=0
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r6o/${v}_pmc -- python $R/bench.py --roofline-only --steps 4 --warmup 1 --cpu-sample 0 --details $R/gpurun_out/r6o/pmc_$v.json > /dev/null 2> $R/gpurun_out/r6o/${v}_pmc.err
done
cd $R
python - <<'PY' | tee gpurun_out/r6o/valu.txt
import csv, glob, collections
for v in ("chains", "tree"):
    f = glob.glob("gpurun_out/r6o/%s_pmc/**/*_counter_collection.csv" % v, recursive=True)
    tot, n, dur = collections.defaultdict(float), collections.defaultdict(int), collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            tot[k] += float(r["Counter_Value"]); n[k] += 1
    steps = max(1, n.get("k_ecmult_keyed<false, 3>", 2) // 2)
    keep = [k for k in tot if k.startswith("k_kc_") or k.startswith("k_keys_bases") or k.startswith("k_ecmult_keyed<false")]
    print(v, "steps", steps, {k: "%.4g per step" % (tot[k] / steps) for k in sorted(keep)}, "all kernels %.4g per step" % (sum(t for k, t in tot.items() if k.startswith("k_") and not k.startswith("k_gen") and not k.startswith("k_gtable") and not k.startswith("k_mul32")) / steps))
PY
find gpurun_out/r6o -name "*.csv" -size +2M -delete
