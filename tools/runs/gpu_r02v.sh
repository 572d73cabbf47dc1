for p in 0 1 0 1; do LAMD_SIDE_PRIORITY=$p timeout 300 python bench.py --skip-extra --cpu-sample 0 > gpurun_out/bench_prio_$p.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_prio_$p.json')); print('side priority $p: cold %.1f warm %.1f mism %d lat1 %.3f lat484 %.3f commit %.3f' % (d['value']/1e6, d['warm_cache']['value']/1e6, d['parity']['mismatches'], d['latency']['ecdsa65_batch_1']['p50_ms'], d['latency']['ecdsa65_batch_484']['p50_ms'], d['latency']['commitment_484_one_htlc_key']['p50_ms']))"; done
cd /tmp && export TMPDIR=/tmp && LAMD_CACHE=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_iso2 -- python $GRAFT_REPO_ROOT/tools/prof_calls.py > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python - <<"PY"
import csv,glob
f=glob.glob("gpurun_out/prof_iso2/*/*_kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
sel=[r for r in rows if int(r["Grid_Size_X"])>=60000 and "gen" not in r["Kernel_Name"] and "gtable" not in r["Kernel_Name"]]
tail=sel[-22:]
t0=int(tail[0]["Start_Timestamp"])
for r in tail:
    print("%9.1f +%8.1f  q%-3s %-36s" % ((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ","")[:36]))
PY
