#!/usr/bin/env python3
"""Summarise the rocprofv3 passes written by tools/pmc_run.sh (CSV output) into a text file + profiles/pmc_latest.json.
usage: pmc_summary.py gpurun_out/pmc_rNN profiles/rNN"""
import collections
import csv
import glob
import json
import sys

base, outp = sys.argv[1], sys.argv[2]


def load(sub, suffix):
    f = glob.glob("%s/%s/runc/*_%s.csv" % (base, sub, suffix))
    return list(csv.DictReader(open(f[0]))) if f else []


def short(name):
    return name.split("(")[0].replace("void ", "")


# ---- kernel trace: durations per kernel, in launch order; ECDSA launches come before Schnorr launches in every step
trace = load("trace", "kernel_trace")
dur, bigdur = collections.defaultdict(list), collections.defaultdict(list)
for r in sorted(trace, key=lambda r: int(r["Start_Timestamp"])):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    dur[short(r["Kernel_Name"])].append(d)
    if int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) >= 500000:      # the 1 M-row launches of the timed loops / isolated calls
        bigdur[short(r["Kernel_Name"])].append(d)
lines = ["# rocprofv3 summary, MI355X, `python bench.py --roofline-only --steps 6 --warmup 2` (the chained cold loop: 1 M ECDSA-65 + 1 M BIP-340 per step)",
         "# pass 1: --kernel-trace --stats; passes 2-5: --kernel-trace --pmc ... (FETCH_SIZE | WRITE_SIZE | SQ set 1 | SQ set 2), one counter group per run", "",
         "[kernel trace, launches over >= 500 000 work items only (the 1 M-row batches; the bench's latency legs launch the same kernels over a few hundred rows): calls, mean us, min us, max us]"]
for k, v in sorted(bigdur.items(), key=lambda kv: -sum(kv[1])):
    if k.startswith("k_") and not k.startswith("k_gen") and not k.startswith("k_gtable"):
        lines.append("  %-32s %4d %12.1f %12.1f %12.1f" % (k, len(v), sum(v) / len(v), min(v), max(v)))
lines += ["", "[kernel trace, every launch: calls, mean us, min us]"]
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if k.startswith("k_"):
        lines.append("  %-32s %4d %12.1f %12.1f" % (k, len(v), sum(v) / len(v), min(v)))
lines.append("")
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write", "sq", "sq2"):
    rows = load(sub, "counter_collection")
    per = collections.defaultdict(list)
    for r in rows:
        per[(short(r["Kernel_Name"]), r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["Grid_Size"])))
    for (k, c), lst in per.items():
        lst.sort()
        big = [v for d, v, g in lst if g >= 500000]      # the 1 M-row launches (drop the generators' / tiny launches)
        if not big:
            big = [v for d, v, g in lst]
        # alternate ECDSA / Schnorr launches
        if k.startswith("k_ecmult") or k.startswith("k_kc_") or k == "k_keys":
            vals[k + " [ecdsa]"][c] = big[0::2]
            vals[k + " [schnorr]"][c] = big[1::2]
        else:
            vals[k][c] = big
mean = lambda l: sum(l) / len(l) if l else float("nan")
summary = {}
for k in sorted(vals):
    if not k.startswith("k_"):
        continue
    lines.append("[%s]  (per launch, mean)" % k)
    d = {}
    for c in sorted(vals[k]):
        d[c] = mean(vals[k][c])
        lines.append("  %-22s %.6g" % (c, d[c]))
    summary[k] = d
    lines.append("")
# ---- the whole step's VALU work, kernel by kernel (round 6: the pipeline is priced as one thing -- every kernel of a step competes for the same
# issue slots): SQ_INSTS_VALU summed over EVERY dispatch of a kernel in the process / the steps the process ran (2 table-driven launches per step)
step_valu, n_big = collections.defaultdict(float), 0
for r in load("sq", "counter_collection"):
    k = short(r["Kernel_Name"])
    if r["Counter_Name"] != "SQ_INSTS_VALU" or not k.startswith("k_") or k.startswith("k_gen") or k.startswith("k_gtable") or k.startswith("k_mul32"):
        continue
    step_valu[k] += float(r["Counter_Value"])
    if k.startswith("k_ecmult_keyed<false") and int(r["Grid_Size"]) >= 500000:
        n_big += 1
steps_run = max(1, n_big // 2)
step_valu = {k: v / steps_run for k, v in step_valu.items()}
tot = sum(step_valu.values())
lines += ["[VALU wave-instructions per STEP (1 M ECDSA-65 + 1 M BIP-340), every dispatch of the process / %d steps]" % steps_run]
for k, v in sorted(step_valu.items(), key=lambda kv: -kv[1]):
    if v > 0:
        lines.append("  %-32s %.4g  (%.1f %%)" % (k, v, 100 * v / tot))
lines += ["  %-32s %.4g" % ("total", tot),
          "  issue time of the total at one wave-instruction per SIMD and 4 cycles (1024 SIMDs): %.3f ms at 2.0 GHz" % (tot * 4 / 1024 / 2.0e9 * 1e3), ""]

cand = [k for k in summary if k.startswith("k_ecmult_keyed") and "[ecdsa]" in k and summary[k].get("SQ_INSTS_VALU", 0) > 0] or \
       [k for k in summary if k.startswith("k_ecmult") and "[ecdsa]" in k]
hot = max(cand, key=lambda k: summary[k].get("SQ_INSTS_VALU", 0))   # the table-driven kernel that does the work (not the empty careful / 10-tooth launches)
e = summary[hot]
kname = hot[:hot.rindex(" [")]
t = mean(bigdur[kname][0::2]) * 1e-6 if len(bigdur[kname]) > 1 else mean(dur[kname]) * 1e-6
# counter passes serialise the kernels (no overlap between the engine's lanes): take that pass's own duration for rates
def pass_t(sub):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9 for r in sorted(load(sub, "kernel_trace"), key=lambda r: int(r["Start_Timestamp"]))
         if short(r["Kernel_Name"]) == kname and int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) >= 500000]
    return mean(d[0::2]) if d else t
t_sq, t_fetch = pass_t("sq"), pass_t("fetch")
nwaves = e.get("SQ_WAVES", 15625)
hbm = (e.get("FETCH_SIZE", 0) + e.get("WRITE_SIZE", 0)) * 1024
lines += ["[derived: %s, ECDSA launch]" % kname,
          "  kernel duration: trace pass (lanes overlap) %.3f ms; counter passes (kernels serialised) %.3f ms" % (t * 1e3, t_sq * 1e3),
          "  HBM-side bytes (FETCH_SIZE+WRITE_SIZE)*1024             %.3e B -> %.0f GB/s (%.1f %% of 8 TB/s); FETCH x2 reading: %.3e B" % (
              hbm, hbm / t_fetch / 1e9, hbm / t_fetch / 8e12 * 100, (2 * e.get("FETCH_SIZE", 0) + e.get("WRITE_SIZE", 0)) * 1024),
          "  VALU wave-instructions per wave (= per signature lane)  %.0f" % (e["SQ_INSTS_VALU"] / nwaves),
          "  shader clock (GRBM_GUI_ACTIVE / 8 XCDs / t)             %.2f GHz" % (e["GRBM_GUI_ACTIVE"] / 8 / t_sq / 1e9),
          "  VALU issue: wave-instr / SIMD / cycle                   %.3f (0.25 = saturated for half-rate ops such as v_mad_u64_u32)" % (
              e["SQ_INSTS_VALU"] / 1024 / (e["GRBM_GUI_ACTIVE"] / 8)),
          "  SQ_WAIT_ANY / SQ_WAVE_CYCLES                            %.3f" % (e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]),
          "  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                       %.3f" % (e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"]), ""]
# ---- FETCH_SIZE calibration (tools/gather_calib.hip under --pmc FETCH_SIZE): requested bytes of a gather of known size / counter
import re
calib = {}
for E in (64, 96):
    try:
        txt = open("%s/calib%d.txt" % (base, E)).read()
        req = float(re.findall(r"= ([0-9.e+]+) bytes requested", txt)[-1])
        rows = [r for r in csv.DictReader(open(glob.glob("%s/calib%d/runc/*_counter_collection.csv" % (base, E))[0]))
                if r["Counter_Name"] == "FETCH_SIZE" and "k_gather" in r["Kernel_Name"]]
        fs = mean([float(r["Counter_Value"]) for r in rows]) * 1024
        calib[E] = {"requested_bytes": req, "fetch_size_bytes": fs, "requested_over_counter": req / fs}
    except Exception as ex:
        calib[E] = {"error": repr(ex)}
factor, fsrc = 1.0, "uncalibrated"
if all("requested_over_counter" in calib[E] for E in (64, 96)):
    # per row: 12 entries of 64 B and 37 of 96 B -- weight the two calibrations by the bytes each class asks for
    w64, w96 = 12 * 64.0, 37 * 96.0
    factor = (w64 * calib[64]["requested_over_counter"] + w96 * calib[96]["requested_over_counter"]) / (w64 + w96)
    factor = max(1.0, factor)      # a counter that over-counts a gather (whole 128-byte lines fetched for 64-96 bytes asked) is real traffic: keep it
    fsrc = "gather calibration: requested/FETCH_SIZE = %.3f (64-byte entries), %.3f (96-byte entries)" % (
        calib[64]["requested_over_counter"], calib[96]["requested_over_counter"])
lines += ["[FETCH_SIZE calibration on random gathers of known size over a 3 GiB table (tools/gather_calib.hip)]"] + [
    "  %d-byte entries: %s" % (E, calib[E]) for E in (64, 96)] + ["  factor applied to FETCH_SIZE of the dominant kernel: %.3f (%s)" % (factor, fsrc), ""]
open(outp + "_pmc_summary.txt", "w").write("\n".join(lines))
import os
os.makedirs(os.path.dirname(outp) or ".", exist_ok=True)
json.dump({"k_ecmult_ecdsa_1M": {"kernel": kname, "hbm_bytes_per_launch": hbm, "fetch_kib": e.get("FETCH_SIZE"), "write_kib": e.get("WRITE_SIZE"),
                                  "valu_insts_per_verify": e["SQ_INSTS_VALU"] / nwaves,
                                  "valu_issue_per_simd_cycle": e["SQ_INSTS_VALU"] / 1024 / (e["GRBM_GUI_ACTIVE"] / 8),
                                  "shader_clock_GHz": e["GRBM_GUI_ACTIVE"] / 8 / t_sq / 1e9,
                                  "fetch_size_factor": factor, "fetch_size_calibration": calib,
                                  "step_valu_wave_instr": step_valu, "step_valu_wave_instr_total": tot, "steps_in_pmc_run": steps_run,
                                  "source": outp + "_pmc_summary.txt (rocprofv3 --pmc of `bench.py --roofline-only`, separate passes; FETCH_SIZE x %.3f, %s)" % (factor, fsrc)}},
          open(outp + "_pmc_latest.json", "w"), indent=1)
# (run on the GPU box the output prefix lies under gpurun_out/: copy <prefix>_pmc_latest.json to profiles/pmc_latest.json, which bench.py reads)
print("\n".join(lines))
