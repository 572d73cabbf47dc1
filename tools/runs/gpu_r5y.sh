#!/bin/bash
# round 5, GPU session y: why the ladder's hot form buys 1-2 % instead of 5-6 %: SQ counters of k_ecmult<3> (VALU instructions, active / wait cycles, memory) for the
# complete form (tools/variants/r5q_base.so) and the hot form (the tree's library), all-distinct rows
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r5y
for lib in tools/variants/r5q_base.so lightning_amd/liblightning_amd.so; do
  tag=$(basename $lib .so)
  (cd /tmp && export TMPDIR=/tmp
   LAMD_LIB_PATH=$R/$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r5y/sq_$tag -- python $R/tools/cold_rows_probe.py 786432 3 > /dev/null 2>&1
   LAMD_LIB_PATH=$R/$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/r5y/sq2_$tag -- python $R/tools/cold_rows_probe.py 786432 3 > /dev/null 2>&1
   LAMD_LIB_PATH=$R/$lib timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r5y/fetch_$tag -- python $R/tools/cold_rows_probe.py 786432 3 > /dev/null 2>&1
   LAMD_LIB_PATH=$R/$lib timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r5y/write_$tag -- python $R/tools/cold_rows_probe.py 786432 3 > /dev/null 2>&1)
done
python3 - <<'PY'
import csv, glob, collections, os
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "r5y")
for d in sorted(glob.glob(root + "/*")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void k_ecmult<3>"):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d), {k: (len(v), sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
find gpurun_out/r5y -name "*.csv" ! -name "*counter_collection.csv" -delete; find gpurun_out/r5y -name "*counter_collection.csv" | xargs gzip -9
