#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel stats plus every
dispatch of the hot kernel.  usage: rocpd_summary.py results.db > profiles/rNN_kernel_trace_stats.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("# rocprofv3 --kernel-trace --stats summary (from %s)" % sys.argv[1].split("/")[-1])
print("%-60s %6s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-60s %6d %14.1f %12.1f %6.2f%%" % (name.split("(")[0][:60], calls, total, avg, pct))
print()
print("# dispatches of k_ecmult / k_ecdsa_prep / k_schnorr_prep / k_keys in launch order (duration in us)")
print("%-16s %10s %10s %6s %6s %6s %8s %8s" % ("kernel", "dur_us", "grid_x", "wg_x", "vgpr", "sgpr", "lds", "scratch"))
q = ("select name,duration,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,scratch_size from kernels "
     "where name like 'k_ecmult%' or name like 'k_ecdsa_prep%' or name like 'k_schnorr_prep%' or name like 'k_keys%' order by start")
for name, dur, gx, wx, vg, sg, lds, scr in cur.execute(q):
    print("%-16s %10.1f %10d %6d %6d %6d %8d %8d" % (name.split("(")[0], dur / 1000.0, gx, wx, vg, sg, lds, scr))
