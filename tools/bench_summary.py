#!/usr/bin/env python3
"""one screen of a bench.py JSON line"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f M/s step %.3f ms (%d steps) | roofline %s frac %.3f (boost %.3f) frac_step %.3f launch %.3f ms sum %.3f <= %.3f | iso %.3f ms | peak_sustained %.2f T (boost %.1f)" % (
    d["value"] / 1e6, d["ms_per_step"], d["steps"], r["mode"], r["frac"], r.get("frac_vs_boost_peak", float("nan")), r.get("frac_step", float("nan")), r["avg_launch_ms"],
    r["sum_of_launch_ms_per_step"], r["ms_per_step"], (r.get("isolated") or {}).get("launch_ms", float("nan")), r.get("peak_sustained", float("nan")), r.get("peak_boost", float("nan"))))
print("peak detail", json.dumps(r.get("peak_sustained_detail")))
if d.get("steady_state"):
    print("steady %.1f M/s" % (d["steady_state"]["value"] / 1e6))
c = d["config"]
print("config h2h", c.get("value_host_to_host"), c.get("host_to_host_over_value"), "predicted_speedup_8", c.get("predicted_speedup_8"))
o = d.get("other_configs_1gpu") or {}
if "gossip_ingest_flood" in o:
    print("ingest", {k: (round(v / 1e6, 2) if isinstance(v, float) and v > 1000 else v) for k, v in o["gossip_ingest_flood"].items() if k not in ("note", "shape")})
h = d.get("value_host_to_host")
if h:
    print("warm %.1f h2h %.1f ratio %.3f" % ((d.get("warm_cache") or {"value": 0})["value"] / 1e6, h["value"] / 1e6, h["ratio_to_value"]),
          {k: round(v["verifies_per_s"] / 1e6, 1) for k, v in d["pcie_inclusive"]["mix_streaming"].items() if isinstance(v, dict)})
if "cfg4_gossip_replay" in o:
    print("cfg4 %.1f cfg5 %.1f" % (o["cfg4_gossip_replay"]["verifies_per_s"] / 1e6, o["cfg5_commit_storm_superbatch"]["verifies_per_s"] / 1e6))
    print("cfg5 streaming", {k: round(v["verifies_per_s"] / 1e6, 1) for k, v in o["cfg5_commit_storm_streaming"].items() if isinstance(v, dict)}, "one per flush",
          {k: (round(v["batches_per_s"]), round(v.get("p50_ms", 0), 3), round(v.get("p99_ms", 0), 3)) for k, v in o["cfg5_commit_storm_one_commitment_per_flush"].items() if isinstance(v, dict)})
    print("sweep", {k: round(v["verifies_per_s"] / 1e6, 1) for k, v in o["key_reuse_sweep"].items() if isinstance(v, dict)})
if "latency" in d:
    print("latency", {k: (round(v["p50_ms"], 3), round(v["p99_ms"], 3)) for k, v in d["latency"].items() if isinstance(v, dict) and "p50_ms" in v},
          "commit first", round(d["latency"].get("commitment_484_one_htlc_key", {}).get("first_sight_ms", 0), 3),
          "commitment_signed one call", d["latency"].get("commitment_signed_one_call_484"),
          "cfg1 ns/call", round(d["latency"].get("cfg1_one_by_one_check_signed_hash", {}).get("ns_per_call", 0)))
ss = d.get("strong_scaling_1gpu")
if ss:
    if "error" in ss:
        print("strong scaling sweep error", ss["error"])
    else:
        for k in ("cfg4_gossip_replay", "cfg5_commit_storm_streaming"):
            print(k, {W: (round(ss[k][W]["slowest_ms"], 3), round(ss[k][W].get("predicted_speedup", 1), 2)) for W in ("1", "2", "4", "8")}, "mism", ss[k]["mismatches"])
        print("gather alone ms", round(ss["gather_alone_ms"], 4), "1/8 shards cfg4", [round(x, 2) for x in ss["cfg4_gossip_replay"]["8"]["shard_ms"]])
if d.get("cpu_baseline"):
    print("cpu", d["cpu_baseline"]["kind"], round(d["cpu_baseline"]["value"]), "parity", d["parity"]["mismatches"])
