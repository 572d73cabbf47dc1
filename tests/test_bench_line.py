"""bench.py's one stdout line (a 20 KB line the driver could not parse) and the multi-rank path the driver's `--gpus N` run executes.

* the line is built from a details dict by `bench.compact_line`: canned numbers in, at most LINE_LIMIT bytes out, every key the bench contract
  names present, a dict that would not fit refused;
* bench.py itself, launched exactly as the driver launches it (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
  --steps K --warmup W`), at world 1, 2 and 8 over gloo with tests/bench_stub.py in the engine's place: ONE line on stdout, n_gpus = world,
  the sharded configs present with the whole verdict vector on every rank, no mismatches.
No GPU, no oracle: what is tested is the bench's own plumbing."""
import copy
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def canned():
    long_note = "x" * 900
    return {
        "metric": "signature verifies/sec (ECDSA+Schnorr mix)", "value": 249912345.678901, "unit": "verifies/s", "n_gpus": 1, "steps": 20, "warmup": 5,
        "ms_per_step": 8.00345678901, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (256-bit modular integer)", "data": "synthetic",
        "config": {"workload": "configs[1]+configs[2]: 1000000 ECDSA (65-byte keys, 65536 distinct) + 1000000 BIP-340 Schnorr per GPU per step, 90% valid / 10% invalid, inputs resident in HBM",
                   "rows_per_gpu_per_step": 2000000, "parallelism": "shard-by-row x1, RCCL all-gather of verdicts", "key_table_cache": "off (tables rebuilt every call)",
                   "value_host_to_host": 236123456.789, "host_to_host_over_value": 0.9451234567, "predicted_speedup_8": {"cfg4": 6.0312345, "cfg5": 5.0612345}},
        "steady_state": {"value": 251234567.8, "unit": "verifies/s", "steps": 250, "seconds": 1.99, "ms_per_step": 7.96, "mismatches": 0, "note": long_note},
        "rates": {"note": long_note},
        "roofline": {"mode": "chained", "launches_timed": 40, "avg_launch_ms": 3.6591234, "rows_in_launch": 952683.5, "sum_of_launches_le_step": True,
                     "kernel": "k_ecmult_keyed<false, 3>: 7-tooth signed comb, bare formulas (1 M-row ECDSA-65 / BIP-340 launches)", "bound": "valu-int32-mul (not hbm, not mfma)",
                     "achieved": 15.12345678, "peak": 35.1312345, "unit": "Tmul32/s", "frac": 0.43012345, "peak_sustained": 35.1312345, "peak_boost": 36.9,
                     "frac_step": 0.3941234, "frac_isolated": 0.601234, "executed_mul32_per_verify": 58086, "traffic": 2117926912.0, "traffic_unit": "HBM bytes per launch",
                     "traffic_over_algorithmic": 13.0736, "algorithmic_bytes_per_launch": 162000000, "valu_instr_per_verify": 89927.9, "valu_issue_frac": 0.93,
                     "peak_note": long_note, "timing": long_note, "rows_note": long_note, "isolated": {"note": long_note}, "pipeline": {"note": long_note},
                     "peak_sustained_detail": {"waves3": {"Tmul32_per_s": 33.7}}},
        "cpu_baseline": {"value": 155123.456, "unit": "verifies/s", "cores": 16, "kind": "port", "sample": "first 400000 ECDSA + 400000 Schnorr rows " + long_note,
                         "note": long_note, "C1_1thread": 9712.3, "C1_all_cores": 155123.456, "C2": 4123.4, "C0": "unavailable", "ns_per_verify_1thread": 102961.2,
                         "legs": {"C1_oracle_1_thread": {"note": long_note}}, "legs_note": long_note, "seconds": 9.1},
        "parity": {"rows_checked": 2000000, "mismatches": 0, "oracle_rows_checked": 800000, "oracle_mismatches": 0, "against": long_note, "mismatches_by_leg": {"a": 0}},
        "pcie_inclusive": {"mix_streaming": {"note": long_note}}, "strong_scaling_1gpu": {"note": long_note}, "latency": {"note": long_note},
        "other_configs_1gpu": {"note": long_note * 4}, "phase_seconds": {"total": 29.0},
    }


def test_line_from_canned_numbers_fits_and_has_the_contract_keys():
    d = canned()
    line, text = bench.compact_line(d)
    assert len(text) < bench.LINE_LIMIT and "\n" not in text
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity"):
        assert k in back, k
    assert back["config"]["workload"].startswith("configs[1]+configs[2]") and back["config"]["predicted_speedup_8"] == {"cfg4": 6.03123, "cfg5": 5.06123}
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "rows_in_launch", "frac_step", "frac_isolated", "peak_sustained",
              "executed_mul32_per_verify", "traffic_over_algorithmic", "valu_instr_per_verify", "kernel"):
        assert k in back["roofline"], k
    assert abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample", "C1_1thread", "C1_all_cores", "C2", "C0"):
        assert k in back["cpu_baseline"], k
    assert len(back["cpu_baseline"]["note"]) <= 160 and len(back["cpu_baseline"]["sample"]) <= 160
    assert back["parity"] == {"rows_checked": 2000000, "mismatches": 0, "oracle_rows_checked": 800000, "oracle_mismatches": 0}
    # sub-reports and prose never travel
    for k in ("rates", "pcie_inclusive", "strong_scaling_1gpu", "latency", "other_configs_1gpu", "phase_seconds"):
        assert k not in back
    assert "peak_note" not in back["roofline"] and "legs" not in back["cpu_baseline"]
    assert back["value"] == 249912000.0          # six significant digits


def test_line_that_would_not_fit_or_lacks_a_key_is_refused():
    d = canned()
    d["config"]["workload"] = "w" * bench.LINE_LIMIT
    with pytest.raises(ValueError, match="bytes"):
        bench.compact_line(d)
    for sect, key in (("roofline", "frac"), ("roofline", "traffic"), ("config", "parallelism"), ("parity", "mismatches")):
        d = canned()
        del d[sect][key]
        with pytest.raises(ValueError, match=key):
            bench.compact_line(d)
    d = canned()
    del d["value"]
    with pytest.raises(KeyError):
        bench.compact_line(d)
    # a multi-rank run has no CPU leg: the key stays, null
    d = canned()
    d["cpu_baseline"] = None
    d["n_gpus"] = 8
    d["sharded_configs"] = {"cfg4_gossip_replay_sharded": {"verifies_per_s": 1.5e9, "ms": 2.7, "ranks": 8, "mismatches": 0, "scaling": "strong", "note": "n" * 500,
                                                           "shard_messages": list(range(8))}}
    line, text = bench.compact_line(d)
    assert line["cpu_baseline"] is None and line["sharded_configs"]["cfg4_gossip_replay_sharded"] == {"verifies_per_s": 1.5e9, "ms": 2.7, "ranks": 8, "mismatches": 0, "scaling": "strong"}


def test_floats_are_rounded_and_non_finite_ones_become_null():
    assert bench._r({"a": [1.23456789012, float("nan")], "b": True, "c": 7, "d": float("inf")}) == {"a": [1.23457, None], "b": True, "c": 7, "d": None}


def _run_bench(world, tmp_path, extra=()):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    details = str(tmp_path / ("details_%d.json" % world))
    args = ["--gpus", str(world), "--steps", "3", "--warmup", "1", "--rows", "2000", "--div", "1000", "--cpu-sample", "0", "--steady-steps", "4", "--h2h-steps", "2",
            "--details", details] + list(extra)
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:   # the driver's own command line
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "bench.py")] + args
    env = dict(os.environ, LAMD_BENCH_STUB=os.path.join(ROOT, "tests", "bench_stub.py"), OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line: %r" % p.stdout[:2000]
    assert len(lines[0]) < bench.LINE_LIMIT
    return json.loads(lines[0]), json.load(open(details)), p.stderr


def test_bench_with_one_rank_over_the_stub(tmp_path):
    line, det, err = _run_bench(1, tmp_path)
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1 and line["parity"]["mismatches"] == 0
    assert line["roofline"]["mode"] == "chained" and line["config"]["rows_per_gpu_per_step"] == 4000
    assert set(line["config"]["predicted_speedup_8"]) == {"cfg4", "cfg5"} and line["config"]["value_host_to_host"] > 0
    assert line["steady_state"]["steps"] == 4
    # the side file holds what the line dropped, and stderr carries it too
    for k in ("rates", "strong_scaling_1gpu", "pcie_inclusive", "phase_seconds"):
        assert k in det, k
    assert "no communicator" in det["strong_scaling_1gpu"]["gather"]
    assert det["strong_scaling_1gpu"]["cfg5_commit_storm_streaming"]["mismatches"] == 0 and det["strong_scaling_1gpu"]["cfg4_gossip_replay"]["mismatches"] == 0
    c4 = det["strong_scaling_1gpu"]["cfg4_gossip_replay"]
    assert len(c4["8"]["shard_ms"]) == 8 and len(c4["8"]["shard_messages"]) == 2 and "one_cut" in c4 and len(c4["one_cut"]["8"]["shard_ms"]) == 8
    assert "BENCH_DETAILS {" in err
    assert "gloo" not in err.lower() or "connected" not in err.lower()        # a one-rank run creates no process group


@pytest.mark.parametrize("world", [2, 8])
def test_bench_multi_rank_path_under_gloo(world, tmp_path):
    line, det, err = _run_bench(world, tmp_path)
    assert line["n_gpus"] == world and line["scaling"] == "weak" and line["cpu_baseline"] is None
    assert line["parity"] == {"rows_checked": world * 4000, "mismatches": 0}
    assert line["roofline"]["mode"] == "pipeline"
    assert abs(line["value"] - world * 4000 * 3 / (line["ms_per_step"] * 3e-3)) / line["value"] < 1e-3     # whole-job rows / max-over-ranks time
    sc = line["sharded_configs"]
    assert set(sc) == {"cfg4_gossip_replay_sharded", "cfg5_commit_storm_streaming_sharded"}
    for v in sc.values():
        assert v["ranks"] == world and v["mismatches"] == 0 and v["scaling"] == "strong" and v["verifies_per_s"] > 0
    d4, d5 = det["sharded_configs"]["cfg4_gossip_replay_sharded"], det["sharded_configs"]["cfg5_commit_storm_streaming_sharded"]
    # configs[3] is cut per message kind: every rank holds range r of the announcements and range r of the updates; the one-cut form is measured beside it
    assert d4["verdicts_on_every_rank"] == d4["messages"] == sum(sum(k) for k in d4["shard_messages"])
    assert len(d4["shard_messages"]) == 2 and all(len(k) == world for k in d4["shard_messages"]) and d4["one_cut_ms"] > 0
    assert d5["verdicts_on_every_rank"] == d5["verifies"] == sum(d5["shard_rows"]["ecdsa"]) + sum(d5["shard_rows"]["schnorr"])
    assert all(r % 484 == 0 for rows in d5["shard_rows"].values() for r in rows)          # a commitment's 484 signatures stay on one rank
