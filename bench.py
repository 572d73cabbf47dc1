#!/usr/bin/env python3
"""Headline benchmark: signature verifies/sec (ECDSA + BIP-340 Schnorr mix) on MI355X.

One "step" = one pass of the hot path over one resident batch: BASELINE.json configs[1] (1 M
ECDSA verifies, 65-byte keys, 32-byte hashes) followed by configs[2] (1 M BIP-340 verifies,
x-only keys), i.e. the ECDSA+Schnorr mix the metric is quoted on.  Inputs are already in HBM
when the timed region starts; every rank processes its own 2 M-row batch (weak scaling, no
data-path collective); with more than one rank the verdict vectors are all-gathered over RCCL
inside the timed region, as the north star asks.

    python bench.py                       # = --gpus 1 --steps 250 --warmup 5 (a timed region of ~2 s)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N              # without RANK in the environment: re-executes itself under torch.distributed.run
    python bench.py --extras              # + latency, PCIe, the other configs, ingest flood, fee grind, recovery, key-reuse sweep

stdout carries exactly ONE JSON line of at most LINE_LIMIT bytes (rank 0): the contract's keys,
`roofline` (HIP-event timing of the dominant kernel), `cpu_baseline` (the CPU oracle on a bounded
sample of the same rows, timed on this host's cores; its verdicts must equal the GPU's) and
`parity`.  Everything else a run measures goes to `bench_details.json` (next to this file, and
under gpurun_out/ when that directory exists) and, as one line, to stderr.  One number per run,
as the reference's own micro-benchmark prints it (onchaind/test/run-grind_feerate.c:146-154).
"""
import argparse
import contextlib
import importlib.util
import json
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before the HIP runtime starts: the engine's lanes need their own hardware queues
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LINE_LIMIT = 6000       # bytes of the one stdout line (round 5's line was 20 KB and the driver could not parse it)

# algorithmic 32x32->64 multiplies per verification (implementation-independent yardstick)
W_ECDSA65 = 1.32e5
W_SCHNORR = 1.65e5
# 32x32->64 multiply-adds the ecmult kernel EXECUTES per verification.  A field multiplication is 99
# v_mad_u64_u32 (one generated block, wrap-around included), a squaring 63; the fused forms add 9 for an addend and make a*b + c*d
# 180.  Mixed addition (group.h gej_add_ge_fast): S + (M+9) + M + (M+9) + S + M + M + (S+9) + 180 + M = 990; doubling: 3 S +
# (S+9) + M + (M+9) + M = 567; acceptance test: 3 M + 1 S.
M_MUL, M_SQR, M_ADD, M_DBL = 99, 63, 990, 567
# Round 6: the G windows are a run of additions in XYZZ coordinates (group.h gexz_add_ge_fast: the Jacobian form's Z^2 is carried, not recomputed):
# 990 - 63 = 927 per window, entering the run costs Z^2 and Z^3 once (S + M), and the acceptance test r*ZZ == X is one multiplication (its Z^2 is there).
M_ADD_XZ = M_ADD - M_SQR
def _mads(dbl, add, g_windows=0, xyzz=False):
    """dbl doublings + add Jacobian mixed additions (+ g_windows G additions), the multiplication by the table's Zc and the acceptance test"""
    if xyzz:
        return dbl * M_DBL + add * M_ADD + M_MUL + (M_SQR + M_MUL) + g_windows * M_ADD_XZ + M_MUL
    return dbl * M_DBL + (add + g_windows) * M_ADD + 3 * M_MUL + M_SQR
# pairs first (verify_core.h "Pairs first", k_ecmult_keyed_pairs): the two comb entries of a column, and two G windows, are summed as affine
# points first -- pass 1 one multiplication per pair (the prefix product), pass 2 M + M + M + (S+9) + (M+9) = 482 per pair -- and enter the
# accumulator by ONE mixed addition; the inversion by division steps is ~1 840 32-bit multiplies (20 batches x 92), shared by the rows of a lane's batch
M_PAIR, M_INV = M_MUL + 3 * M_MUL + (M_SQR + 9) + (M_MUL + 9), 1840
def _mads_pairs(cols, g_windows, rows_per_inversion):
    pairs = cols + g_windows // 2
    adds = (cols - 1) + g_windows // 2 + (g_windows & 1)
    return pairs * M_PAIR + adds * M_ADD + (cols - 1) * M_DBL + M_MUL + 3 * M_MUL + M_SQR + M_INV / max(1.0, rows_per_inversion)
# ladder: 132 doublings (+1 for 2Q), 66 + 6 table additions, G windows; combs: both halves made odd by a lattice vector (no
# repair additions), the first table point initialises the accumulator: 2D - 1 additions + D - 1 doublings + G windows.  G windows: one
# mixed addition per window of the static table -- 11 with the 24-bit windows that ship from round 4 on (12 with 22 bits: rounds 1-3)
def w_exec_table(g_windows, pairs=False, rows_per_inversion=4.8, xyzz=True):
    if pairs:
        return {0: _mads(132 + 1, 66 + g_windows + 6), 7: _mads_pairs(19, g_windows, rows_per_inversion), 10: _mads_pairs(13, g_windows, rows_per_inversion)}
    return {0: _mads(132 + 1, 66 + 6, g_windows), 7: _mads(18, 37, g_windows, xyzz), 10: _mads(12, 25, g_windows, xyzz)}   # ladder (Jacobian G run), 7-tooth comb, 10-tooth comb
def g_windows_of(gtable_bytes):
    for bits in (16, 22, 24, 26):
        w = (256 + bits - 1) // bits
        if gtable_bytes in (w * (64 << bits), w * (72 << bits)):
            return w
    return 11
W_EXEC = w_exec_table(11)
# measured dependent-free v_mad_u64_u32 issue rate of one MI355X (profiles/r01_microbench_valu_rates.txt)
P_MUL32 = 3.69e13
HBM_PEAK_GBS = 8000.0
BYTES_ECDSA65 = 32 + 64 + 65 + 1
BYTES_SCHNORR = 32 + 32 + 64 + 1


# ------------------------------------------------------------------------------------------------------------------------------------
# the ONE stdout line
def _r(x, digits=6):
    """floats to `digits` significant digits (a line of at most LINE_LIMIT bytes has no room for 17-digit floats)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    if hasattr(x, "item"):         # numpy scalars
        return _r(x.item(), digits)
    return x


LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
CONFIG_KEYS = ("workload", "rows_per_gpu_per_step", "parallelism", "key_table_cache", "value_host_to_host", "host_to_host_over_value", "predicted_speedup_8")
ROOFLINE_KEYS = ("kernel", "bound", "mode", "avg_launch_ms", "rows_in_launch", "executed_mul32_per_verify", "achieved", "peak", "unit", "frac",
                 "peak_sustained", "peak_boost", "frac_step", "frac_isolated", "traffic", "traffic_unit", "traffic_over_algorithmic",
                 "algorithmic_bytes_per_launch", "valu_instr_per_verify", "valu_issue_frac", "valu_issue_frac_step", "valu_share_key_tables", "launches_timed", "sum_of_launches_le_step")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "note", "C1_1thread", "C1_all_cores", "C2", "C0", "ns_per_verify_1thread", "seconds")
PARITY_KEYS = ("rows_checked", "mismatches", "oracle_rows_checked", "oracle_mismatches")
REQUIRED = {"": LINE_KEYS + ("config", "roofline", "cpu_baseline", "parity"),
            "config": ("workload", "rows_per_gpu_per_step", "parallelism"),
            "roofline": ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "rows_in_launch"),
            "parity": ("rows_checked", "mismatches")}


def compact_line(d):
    """the details dict of a run -> the dict printed as the ONE stdout line: the bench contract's keys, `config`, `roofline`, `cpu_baseline`
    and `parity` cut to the keys above; notes and sub-reports stay in bench_details.json.  Raises when a required key is missing or the
    line would not fit LINE_LIMIT (tests/test_bench_line.py runs this on canned numbers)."""
    line = {k: d[k] for k in LINE_KEYS}
    line["config"] = {k: d["config"][k] for k in CONFIG_KEYS if d["config"].get(k) is not None}
    line["roofline"] = {k: d["roofline"].get(k) for k in ROOFLINE_KEYS if k in d["roofline"]}
    cb = d.get("cpu_baseline")
    if cb is not None:
        cb = {k: cb[k] for k in CPU_KEYS if cb.get(k) is not None}
        for k in ("note", "sample"):
            if len(str(cb.get(k, ""))) > 160:
                cb[k] = str(cb[k])[:157] + "..."
    line["cpu_baseline"] = cb
    line["parity"] = {k: d["parity"][k] for k in PARITY_KEYS if k in d["parity"]}
    if d.get("steady_state"):
        line["steady_state"] = {k: d["steady_state"][k] for k in ("value", "steps", "ms_per_step")}
    if d.get("sharded_configs"):
        line["sharded_configs"] = {name: {k: v[k] for k in ("verifies_per_s", "ms", "ranks", "mismatches", "scaling") if k in v}
                                   for name, v in d["sharded_configs"].items()}
    line["details"] = d.get("details_file", "bench_details.json")
    line = _r(line)
    for sect, keys in REQUIRED.items():
        have = line if sect == "" else line[sect]
        missing = [k for k in keys if k not in have]
        if missing:
            raise ValueError("bench line: %s lacks %s" % (sect or "top level", missing))
    text = json.dumps(line, separators=(", ", ": "))
    if len(text) >= LINE_LIMIT:
        raise ValueError("bench line is %d bytes (limit %d): move the new keys to bench_details.json" % (len(text), LINE_LIMIT))
    return line, text


def write_outputs(details, json_fd, details_path=None):
    """details -> bench_details.json (+ gpurun_out/ copy), one line on stderr, the compact line on the saved stdout"""
    paths = [details_path or os.path.join(ROOT, "bench_details.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")) and details_path is None:
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_details.json"))
    details["details_file"] = os.path.relpath(paths[0], ROOT)
    line, text = compact_line(details)
    blob = json.dumps(_r(details, 9), indent=1)
    for p in paths:
        try:
            with open(p, "w") as f:
                f.write(blob + "\n")
        except OSError as e:        # a read-only tree must not cost the line
            sys.stderr.write("bench_details: %s\n" % e)
    sys.stderr.write("BENCH_DETAILS " + json.dumps(_r(details, 6)) + "\n")
    sys.stderr.flush()
    os.write(json_fd, (text + "\n").encode())
    return line


# ------------------------------------------------------------------------------------------------------------------------------------
# the machine under the bench: an MI355X through PyTorch-ROCm -- or, for the CPU tests of the multi-rank path, a stand-in module
def bind_to_gpu_numa_node(torch, local_rank):
    """The host side of a rank -- the producer that fills the staging sets, the threads the engine starts -- on the CPUs of the NUMA node its GPU hangs on,
    BEFORE anything is allocated (first touch decides where the workload's host arrays live).  On the two-socket hosts of the pool a process that floats
    over both sockets got its rows on the far node in about one run of three: the whole-job stream of configs[4] then read 30-33 ms instead of 23-24
    (profiles/r06_strong_scaling.txt, session ac) and host -> host 0.91 x instead of 0.98 x of the resident rate.  What every RCCL launcher does per rank;
    LAMD_BENCH_NUMA=0 leaves the affinity alone.  -> {"gpu_node", "cpus", "bound"} for the details file"""
    info = {"gpu_node": None, "cpus": None, "bound": False}
    if os.environ.get("LAMD_BENCH_NUMA", "1") == "0":
        return info
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        info["gpu_node"] = node
        if node < 0:
            return info
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if len(cpus) >= 8:           # (a cpuset that leaves the rank fewer CPUs on that node than the baseline's threads: stay where we are)
            os.sched_setaffinity(0, cpus)
            info["cpus"], info["bound"] = len(cpus), True
    except Exception as e:           # no sysfs, no such attribute: the run goes on unbound
        info["error"] = str(e)[:80]
    return info


class GpuPlatform:
    backend, is_stub = "nccl", False

    def __init__(self, local_rank):
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: lightning_amd has no CPU fallback")
        torch.cuda.set_device(local_rank)
        self.numa = bind_to_gpu_numa_node(torch, local_rank)
        from lightning_amd import Engine, workload
        self.torch, self.Engine, self.workload = torch, Engine, workload
        self.device = "cuda:%d" % local_rank

    def synchronize(self):
        self.torch.cuda.synchronize()

    def stream_ptr(self):
        return self.torch.cuda.current_stream().cuda_stream

    def new_event(self):
        ev = self.torch.cuda.Event()
        ev.record()
        return ev


def load_platform(local_rank):
    """LAMD_BENCH_STUB=<module.py>: tests/bench_stub.py -- host-memory "devices", gloo, verdicts a fixed function of the row bytes -- so that
    the code the driver's `--gpus 8` run executes (argument handling, late gathers, sharded configs, the line) runs in the CPU suite"""
    stub = os.environ.get("LAMD_BENCH_STUB")
    if not stub:
        return GpuPlatform(local_rank)
    spec = importlib.util.spec_from_file_location("lamd_bench_stub", stub)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Platform(local_rank)


class Clock:
    """wall-clock seconds per phase of the run (bench_details.json `phase_seconds`)"""

    def __init__(self):
        self.t = {}

    @contextlib.contextmanager
    def __call__(self, name):
        t0 = time.perf_counter()
        try:
            yield
        finally:
            self.t[name] = self.t.get(name, 0.0) + time.perf_counter() - t0


# ------------------------------------------------------------------------------------------------------------------------------------
def pin_storm(eng, st):
    """the storm's host columns pinned once, outside every clock (lamd_host_register) -- the state a sidecar's shared blocks are in: their rows are
    then queued IN PLACE (no staging copy on the host).  -> True when the runtime pinned every column"""
    return all(eng.host_register(c) for kind in ("ecdsa", "schnorr") if kind in st for c in st[kind].cols)


def unpin_storm(eng, st):
    for kind in ("ecdsa", "schnorr"):
        if kind in st:
            for c in st[kind].cols:
                eng.host_unregister(c)


def stream_shard(eng, st, bounds, k, grp, depth, first_grp=0, inplace=False):
    """configs[4] for shard k: the commitments [bounds[kind][k], bounds[kind][k+1]) of BOTH kinds through the streaming queue as ONE pipeline -- flushes
    of `grp` rows, the two kinds taking turns in proportion to their length, up to `depth` flushes in flight -- from host memory to verdicts in host
    memory.  (Until round 5 the ECDSA rows were streamed and drained before the first BIP-340 flush went out: two pipeline fills and two drains per
    shard, 2.6 ms of a 5 ms 1/8 shard.)  first_grp > 0: the first flush of each kind is that many rows (the device starts sooner).  inplace: the rows
    are queued where they are (lamd_queue_*_batch_inplace; the columns pinned by pin_storm) instead of being copied into the staging set.
    -> {kind: uint8 verdicts of the shard}"""
    import numpy as np
    jobs = []
    per_unit = int(st["per"])
    for kind in ("ecdsa", "schnorr"):
        if kind not in st:
            continue
        a, z = int(bounds[kind][k]), int(bounds[kind][k + 1])
        span = max(1, z - a)
        o, step = a, (max(per_unit, first_grp // per_unit * per_unit) if first_grp else grp)
        while o < z:
            e = min(z, o + step)
            jobs.append(((o - a) / span, kind, o, e))
            o, step = e, grp
    jobs.sort(key=lambda j: j[0])
    got = {kind: np.zeros(int(bounds[kind][k + 1]) - int(bounds[kind][k]), dtype=np.uint8) for kind in ("ecdsa", "schnorr") if kind in st}
    pend = []

    def collect():
        kind, o, e = pend.pop(0)
        a = int(bounds[kind][k])
        got[kind][o - a:e - a] = eng.wait()
    for _, kind, o, e in jobs:
        wl = st[kind]
        if kind == "ecdsa":
            (eng.queue_ecdsa_batch_inplace if inplace else eng.queue_ecdsa_batch)(wl.cols[0][o:e], wl.cols[1][o:e], wl.cols[2][o:e])
        else:
            (eng.queue_schnorr_batch_inplace if inplace else eng.queue_schnorr_batch)(wl.cols[0][o:e], wl.cols[1][o:e], wl.cols[2][o:e])
        eng.flush()
        pend.append((kind, o, e))
        if len(pend) == depth:
            collect()
    while pend:
        collect()
    return got


def gossip_spans(g, ranges, device):
    """the arguments of ONE lamd_sigcheck_gossip_spans_device call over several ranges of a resident gossip workload (made outside the clock, like the
    rebased row offsets of a range call): (n, d_start, d_len, d_ids, d_rowbase, rows, d_verdict)"""
    import torch
    starts = torch.cat([g.d_off[lo:hi] for lo, hi in ranges]).contiguous()
    lens = torch.cat([g.d_off[lo + 1:hi + 1] - g.d_off[lo:hi] for lo, hi in ranges]).contiguous()
    ids = torch.cat([g.d_ids[lo:hi] for lo, hi in ranges]).contiguous()
    rpm = torch.cat([g.d_rowbase[lo + 1:hi + 1] - g.d_rowbase[lo:hi] for lo, hi in ranges])
    rowbase = torch.cat([torch.zeros(1, dtype=rpm.dtype, device=rpm.device), torch.cumsum(rpm, 0)]).contiguous()
    n = int(starts.numel())
    return n, starts, lens, ids, rowbase, int(rowbase[-1].item()), torch.zeros(n, dtype=torch.int8, device=device)


def storm_first_flush(per):
    """rows of a shard's first flush per kind (LAMD_BENCH_FIRST_FLUSH commitments; default a quarter of the 256-commitment flush)"""
    return int(os.environ.get("LAMD_BENCH_FIRST_FLUSH", "64")) * per


def sharded_configs(plat, eng, rank, world, tstream, div=1):
    """BASELINE configs[3] and [4] the way the north star words them: ONE global job split over the ranks on message /
    commitment boundaries (lightning_amd.sharding.run_sharded: shard -> verify locally -> ragged RCCL all-gather of the verdict
    bytes), every rank ending with the whole verdict vector.  Strong scaling: the job is fixed, the time is max over ranks.
    Every rank generates the same synthetic job (same seed) and touches only its shard."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from lightning_amd import sharding
    workload, device = plat.workload, plat.device
    out = {}

    def timed(fn, reps):
        ts, res = [], None
        for _ in range(reps):
            dist.barrier()
            plat.synchronize(); eng.synchronize()
            t1 = time.perf_counter()
            res = fn()
            plat.synchronize(); eng.synchronize()
            t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ts.append(float(t.item()))
        return ts, res

    # ---- configs[3]: gossip replay, 500 k channel_announcement + 2 M channel_update.  Cut PER MESSAGE KIND (sharding.segment_bounds: every rank takes
    # range r of the announcements AND range r of the updates, two asynchronous calls that overlap on the engine's lanes, one ragged all-gather) --
    # with one cut over the whole job five ranks of eight hold nothing but announcements (the ladder) and three nothing but updates (every node
    # key's 10-tooth table on every one of them): that form is measured beside it (`one_cut`).
    g = workload.make_gossip(eng, 500_000 // div, 2_000_000 // div, n_nodes=max(16, 15000 // div), device=device)
    gw = sharding.gossip_weights(g.msgs, g.off)      # cuts on message boundaries, balanced by what a message costs inside its kind
    seg = [0, g.n_cann, g.n]
    sb = sharding.segment_bounds(seg, world, gw)
    b1 = sharding.shard_bounds(g.n, world, None, gw)
    prepared = {}

    def prepare(lo, hi):
        if hi > lo and (lo, hi) not in prepared:
            prepared[(lo, hi)] = ((g.d_rowbase[lo:hi + 1] - g.d_rowbase[lo]).contiguous(), int(g.rowbase[hi] - g.rowbase[lo]),
                                  torch.zeros(hi - lo, dtype=torch.int8, device=device))
    for s in range(sb.shape[0]):
        prepare(int(sb[s, rank]), int(sb[s, rank + 1]))
    prepare(int(b1[rank]), int(b1[rank + 1]))
    plat.synchronize()

    def gossip_range(a, z):
        rb, rows, d_v = prepared[(a, z)]
        eng.sigcheck_gossip_device(z - a, g.d_msgs, g.d_off[a:z + 1], g.d_ids[a:z], rb, rows, d_v)
        return d_v
    wait = lambda: eng.stream_wait_results(tstream)     # the collective (torch's stream) starts when the verdicts exist: device-side edge
    empty = torch.empty(0, dtype=torch.int8, device=device)
    reps = 2 + eng.info()["lanes"]
    # ONE spans call over the rank's ranges (lamd_sigcheck_gossip_spans_device, sharding's `verify_ranges`): one front end per rank, and -- since the cold
    # rows' ladder starts right behind the key classification -- the announcements' ladder runs under the updates' table kernels inside the one call.
    # Measured on one GPU playing every rank (profiles/r06_strong_scaling.txt, sessions aa-ad): 3.3 against 3.4-3.8 ms for two asynchronous range calls
    # at W = 8, 18.2 against 19.0 ms at W = 1 (6.0 against 5.8 at W = 4).  The two-call form is timed beside it.
    mine = [(int(sb[s, rank]), int(sb[s, rank + 1])) for s in range(sb.shape[0]) if sb[s, rank + 1] > sb[s, rank]]
    sp = gossip_spans(g, mine, device) if mine else None
    plat.synchronize()

    def gossip_ranges(ranges):
        assert ranges == mine
        eng.sigcheck_gossip_spans_device(sp[0], g.d_msgs, sp[1], sp[2], sp[3], sp[4], sp[5], sp[6])
        return sp[6]
    ts, (full, _) = timed(lambda: sharding.run_sharded_segments(g.n, seg, rank, world, gossip_range, gw, wait, empty, verify_ranges=gossip_ranges), reps)
    bad = int((full.cpu().numpy() != g.expect).sum())
    ts2, (full2, _) = timed(lambda: sharding.run_sharded_segments(g.n, seg, rank, world, gossip_range, gw, wait, empty), reps)
    bad += int((full2.cpu().numpy() != g.expect).sum())

    def one_cut(a, z):
        v = gossip_range(a, z)
        wait()
        return v
    ts1, (full1, _) = timed(lambda: sharding.run_sharded(g.n, rank, world, one_cut, None, gw), reps)
    bad += int((full1.cpu().numpy() != g.expect).sum())
    out["cfg4_gossip_replay_sharded"] = {"messages": g.n, "verifies": g.rows, "ranks": world, "split": "per message kind (announcements | updates), range r of each per rank: ONE spans call",
                                         "two_calls_ms": min(ts2[-2:]) * 1e3, "two_calls_verifies_per_s": g.rows / min(ts2[-2:]),
                                         "shard_messages": [[int(sb[s, k + 1] - sb[s, k]) for k in range(world)] for s in range(sb.shape[0])],
                                         "verifies_per_s": g.rows / min(ts[-2:]), "messages_per_s": g.n / min(ts[-2:]), "ms": min(ts[-2:]) * 1e3,
                                         "one_cut_ms": min(ts1[-2:]) * 1e3, "one_cut_verifies_per_s": g.rows / min(ts1[-2:]),
                                         "mismatches": bad, "scaling": "strong", "verdicts_on_every_rank": int(full.numel()),
                                         "note": "raw wire messages resident in HBM; per rank: framing + SHA256d + verification of its ranges, then the "
                                                 "ragged all-gather of int8 verdicts; every rank checks the WHOLE gathered vector against construction"}
    del g, prepared, gw
    # ---- configs[4]: commit_tx storm, 10 k channels x 484, streaming batches from host memory, 484-row groups kept whole
    st = workload.make_commit_storm(eng, max(world, 10_000 // div), device=device)
    per, grp = st["per"], 256 * st["per"]

    def storm():
        bb = {kind: sharding.shard_bounds(st[kind].n, world, [per] * (st[kind].n // per)) for kind in ("ecdsa", "schnorr")}
        got = stream_shard(eng, st, bb, rank, grp, min(8, eng.info()["queue_sets"] - 1), storm_first_flush(per))
        return {kind: (sharding.all_gather_verdicts(torch.from_numpy(got[kind]).to(device), bb[kind], rank, world), bb[kind]) for kind in ("ecdsa", "schnorr")}
    ts, res = timed(storm, 3)
    bad, shard_rows, nfull = 0, {}, 0
    for kind in ("ecdsa", "schnorr"):
        full, bb = res[kind]
        bad += int((full.cpu().numpy().astype(bool) != st[kind].expect).sum())
        nfull += int(full.numel())
        shard_rows[kind] = [int(bb[k + 1] - bb[k]) for k in range(world)]
        assert all(int(x) % per == 0 for x in bb)
    nv = st["ecdsa"].n + st["schnorr"].n
    out["cfg5_commit_storm_streaming_sharded"] = {"channels": max(world, 10_000 // div), "verifies": nv, "ranks": world, "shard_rows": shard_rows,
                                                  "verifies_per_s": nv / min(ts[1:]), "ms": min(ts[1:]) * 1e3, "mismatches": bad, "scaling": "strong",
                                                  "verdicts_on_every_rank": nfull,
                                                  "note": "inputs in host memory: per rank its commitments stream through the pinned staging queue "
                                                          "(256 commitments per flush, up to 8 flushes in flight), then the ragged all-gather of the verdict bytes"}
    # every rank must have ended with the whole vector and agree on the count of bad verdicts
    tot = torch.tensor([out["cfg4_gossip_replay_sharded"]["mismatches"] + bad], dtype=torch.int64, device=device)
    dist.all_reduce(tot)
    out["mismatches_summed_over_ranks"] = int(tot.item())
    return out


def strong_scaling_sweep(plat, eng, tstream, div=1, in_place_too=False):
    """What ONE rank of a strong-scaling run of BASELINE configs[3] / configs[4] would see, measured on one GPU.  For W in 1, 2, 4, 8 the global
    job is cut as `sharding.run_sharded` cuts it for W ranks and EVERY shard k of W is run by itself -- verification of the shard, then a
    stand-in for the all-gather of its (padded) verdict bytes: a device copy of the same size on torch's stream, ordered after the verdicts by the
    same device-side edge the collective uses (a --gpus 1 run creates no RCCL communicator; launched under torch.distributed.run with
    LAMD_BENCH_GATHER=1 the copy is a real one-rank all-gather) -- and timed from submit to "gathered vector complete".  T(W) = the slowest shard
    of W; predicted_speedup_W = T(1) / T(W)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from lightning_amd import sharding
    workload, device = plat.workload, plat.device
    out = {}

    def best(fn, reps):
        ts = []
        for _ in range(reps):
            plat.synchronize(); eng.synchronize()
            t1 = time.perf_counter()
            fn()
            plat.synchronize(); eng.synchronize()
            ts.append(time.perf_counter() - t1)
        return min(ts[1:]) if len(ts) > 1 else ts[0]

    def gather(local):
        buf = local.view(torch.uint8)
        m = (buf.numel() + 15) // 16 * 16
        pad = torch.zeros(m, dtype=torch.uint8, device=device)
        pad[:buf.numel()] = buf
        res = torch.empty(m, dtype=torch.uint8, device=device)
        if dist.is_initialized():
            dist.all_gather_into_tensor(res, pad)
        else:
            res.copy_(pad)
        return res
    probe = torch.zeros(max(16, 312_500 // div), dtype=torch.uint8, device=device)
    gather(probe)
    t_gather = best(lambda: gather(probe), 6)
    lanes = eng.info()["lanes"]
    # ---- configs[3]: gossip replay.  Shard k of W = range k of the announcements AND range k of the updates (sharding.segment_bounds: cut per message
    # kind, balanced by cost inside the kind, two asynchronous calls that overlap on the engine's lanes); the one-cut form of rounds 1-5 beside it
    g = workload.make_gossip(eng, 500_000 // div, 2_000_000 // div, n_nodes=max(16, 15000 // div), device=device)
    gw = sharding.gossip_weights(g.msgs, g.off)
    seg = [0, g.n_cann, g.n]

    def shard_fn(ranges):
        """-> (one(), verdict tensors, ranges): the calls of one rank for its ranges, then the gather stand-in over all of its verdict bytes"""
        prep = []
        for lo, hi in ranges:
            if hi > lo:
                prep.append((lo, hi, (g.d_rowbase[lo:hi + 1] - g.d_rowbase[lo]).contiguous(), int(g.rowbase[hi] - g.rowbase[lo]),
                             torch.zeros(hi - lo, dtype=torch.int8, device=device)))
        plat.synchronize()

        def one():
            for lo, hi, rb, rows, d_v in prep:
                eng.sigcheck_gossip_device(hi - lo, g.d_msgs, g.d_off[lo:hi + 1], g.d_ids[lo:hi], rb, rows, d_v)
            eng.stream_wait_results(tstream)
            return gather(torch.cat([p[4] for p in prep]) if len(prep) > 1 else prep[0][4])
        return one, prep
    res3, res3_one, bad3 = {}, {}, 0
    for W in (1, 2, 4, 8):
        sb = sharding.segment_bounds(seg, W, gw)
        b1 = sharding.shard_bounds(g.n, W, None, gw)
        shard_ms, one_cut_ms, spans_ms = [], [], []
        for k in range(W):
            for form, ranges, dst in (("kinds", [(int(sb[s, k]), int(sb[s, k + 1])) for s in range(sb.shape[0])], shard_ms),
                                      ("kinds_one_spans_call", [(int(sb[s, k]), int(sb[s, k + 1])) for s in range(sb.shape[0])], spans_ms),
                                      ("one_cut", [(int(b1[k]), int(b1[k + 1]))], one_cut_ms)):
                if form == "kinds_one_spans_call":      # ONE spans call over the shard's ranges (lamd_sigcheck_gossip_spans_device)
                    live = [(lo, hi) for lo, hi in ranges if hi > lo]
                    sp = gossip_spans(g, live, device)
                    plat.synchronize()

                    def one(sp=sp):
                        eng.sigcheck_gossip_spans_device(sp[0], g.d_msgs, sp[1], sp[2], sp[3], sp[4], sp[5], sp[6])
                        eng.stream_wait_results(tstream)
                        return gather(sp[6])
                    for _ in range(lanes if W == 1 and k == 0 else 1):
                        one()
                    dst.append(best(one, 5 if W > 1 else 6) * 1e3)
                    bad3 += int((sp[6].cpu().numpy() != np.concatenate([g.expect[lo:hi] for lo, hi in live])).sum())
                    continue
                one, prep = shard_fn(ranges)
                for _ in range(lanes if W == 1 and k == 0 else 1):   # every lane allocates its workspaces for the largest shape once
                    one()
                dst.append(best(one, 5 if W > 1 else 6) * 1e3)
                for lo, hi, rb, rows, d_v in prep:
                    bad3 += int((d_v.cpu().numpy() != g.expect[lo:hi]).sum())
        res3[str(W)] = {"shard_ms": spans_ms, "slowest_ms": max(spans_ms),
                        "shard_messages": [[int(sb[s, k + 1] - sb[s, k]) for k in range(W)] for s in range(sb.shape[0])]}
        res3[str(W)]["two_calls_shard_ms"] = shard_ms
        if os.environ.get("LAMD_BENCH_BY_KEY", "0") == "1" and not plat.is_stub:
            # experiment (profiles/r06_strong_scaling.txt): shard k = every message whose SIGNER KEY hashes to k -- the node id of a channel_update, node_id_1 of a
            # channel_announcement (synthetic layout: feature length 0, the key at byte 300) -- as ONE spans call: a rank builds the tables of ITS node keys only
            if W == 1:
                key33 = np.where(np.arange(g.n)[:, None] < g.n_cann, 0, g.ids).astype(np.uint8)
                o = g.off[:g.n_cann].astype(np.int64)
                key33[:g.n_cann] = g.msgs[(o[:, None] + 300 + np.arange(33)[None, :])]
                h = np.zeros(g.n, dtype=np.uint64) + np.uint64(1469598103934665603)
                for b in range(33):
                    h = (h ^ key33[:, b].astype(np.uint64)) * np.uint64(1099511628211)
                key_hash = (h >> np.uint64(17))
            by_key_ms = []
            for k in range(W):
                sel = np.nonzero(key_hash % np.uint64(W) == np.uint64(k))[0]
                d_sel = torch.from_numpy(sel).to(device)
                rpm = (g.d_rowbase[1:] - g.d_rowbase[:-1])[d_sel]
                rowbase = torch.cat([torch.zeros(1, dtype=rpm.dtype, device=device), torch.cumsum(rpm, 0)]).contiguous()
                sp = (len(sel), g.d_off[:-1][d_sel].contiguous(), (g.d_off[1:] - g.d_off[:-1])[d_sel].contiguous(), g.d_ids[d_sel].contiguous(), rowbase,
                      int(rowbase[-1].item()), torch.zeros(len(sel), dtype=torch.int8, device=device))
                plat.synchronize()

                def one(sp=sp):
                    eng.sigcheck_gossip_spans_device(sp[0], g.d_msgs, sp[1], sp[2], sp[3], sp[4], sp[5], sp[6])
                    eng.stream_wait_results(tstream)
                    return gather(sp[6])
                for _ in range(lanes if W == 1 and k == 0 else 1):
                    one()
                by_key_ms.append(best(one, 5 if W > 1 else 6) * 1e3)
                bad3 += int((sp[6].cpu().numpy() != g.expect[sel]).sum())
            res3[str(W)]["by_signer_key_shard_ms"] = by_key_ms
        res3_one[str(W)] = {"shard_ms": one_cut_ms, "slowest_ms": max(one_cut_ms), "shard_messages": [int(b1[k + 1] - b1[k]) for k in range(W)]}
    t1 = min(res3["1"]["slowest_ms"], res3_one["1"]["slowest_ms"], max(res3["1"]["two_calls_shard_ms"]))      # T(1): the best way to run the whole job on one GPU
    for W in ("2", "4", "8"):
        res3[W]["predicted_speedup"] = t1 / res3[W]["slowest_ms"]
        res3_one[W]["predicted_speedup"] = t1 / res3_one[W]["slowest_ms"]
    out["cfg4_gossip_replay"] = dict(res3, verifies=g.rows, messages=g.n, mismatches=bad3, predicted_speedup_8=res3["8"]["predicted_speedup"],
                                     verifies_per_s_predicted_8=g.rows / (res3["8"]["slowest_ms"] * 1e-3), t1_ms=t1,
                                     split="per message kind: shard k = range k of the announcements + range k of the updates as ONE spans call (two_calls_shard_ms: as two asynchronous range calls)",
                                     one_cut=dict(res3_one, predicted_speedup_8=res3_one["8"]["predicted_speedup"],
                                                  note="one cut over the whole job, balanced by cost (rounds 1-5): the slowest shard is a kind's worst case"))
    del g
    # ---- configs[4]: commit storm, streaming from host memory, commitments kept whole
    st = workload.make_commit_storm(eng, max(8, 10_000 // div), device=device)
    per, grp = st["per"], 256 * st["per"]
    depth = min(8, eng.info()["queue_sets"] - 1)
    first = storm_first_flush(per)
    # (the in-place producer beside the copying one only under --extras: it pins 600 MB of host columns, and the default run stays the plainest path)
    pinned = in_place_too and pin_storm(eng, st)

    def sweep5(inplace, Ws=(1, 2, 4, 8)):
        res5, bad5 = {}, 0
        for W in Ws:
            bb = {kind: sharding.shard_bounds(st[kind].n, W, [per] * (st[kind].n // per)) for kind in ("ecdsa", "schnorr")}
            shard_ms = []
            for k in range(W):
                keep = {}

                def one():
                    keep.update(stream_shard(eng, st, bb, k, grp, depth, first, inplace=inplace))
                    for kind in ("ecdsa", "schnorr"):
                        gather(torch.from_numpy(keep[kind]).to(device))
                shard_ms.append(best(one, 3 if W < 4 else 5) * 1e3)      # short shards: more repetitions (a 5 ms shard that meets one hiccup reads 10 ms)
                for kind, got in keep.items():
                    bad5 += int((got.astype(bool) != st[kind].expect[int(bb[kind][k]):int(bb[kind][k + 1])]).sum())
            res5[str(W)] = {"shard_ms": shard_ms, "slowest_ms": max(shard_ms), "shard_commitments": [int((bb["ecdsa"][k + 1] - bb["ecdsa"][k] + bb["schnorr"][k + 1] - bb["schnorr"][k]) // per) for k in range(W)]}
        for W in res5:
            if W != "1":
                res5[W]["predicted_speedup"] = res5["1"]["slowest_ms"] / res5[W]["slowest_ms"]
        return res5, bad5
    nv = st["ecdsa"].n + st["schnorr"].n
    # the producer copies its rows from pageable host memory into the engine's pinned staging set (lamd_queue_*_batch); beside it, at W = 1 and 8, the rows
    # where a sidecar holds them -- pinned host memory, queued in place (no host copy; the form lamd_served streams its clients' blocks with)
    res5, bad5 = sweep5(False)
    out["cfg5_commit_storm_streaming"] = dict(res5, verifies=nv, mismatches=bad5, predicted_speedup_8=res5["8"]["predicted_speedup"],
                                              verifies_per_s_predicted_8=nv / (res5["8"]["slowest_ms"] * 1e-3), first_flush_rows=first,
                                              producer="rows copied from pageable host memory into the pinned staging set (lamd_queue_*_batch)")
    if pinned:
        resi, badi = sweep5(True, (1, 8))
        bad5 += badi
        out["cfg5_commit_storm_streaming"]["mismatches"] = bad5
        out["cfg5_commit_storm_streaming"]["in_place_producer"] = {W: {"slowest_ms": resi[W]["slowest_ms"], "predicted_speedup": resi[W].get("predicted_speedup")} for W in resi}
        unpin_storm(eng, st)
    out["gather_alone_ms"] = t_gather * 1e3
    out["gather"] = "one-rank RCCL all-gather" if dist.is_initialized() else "device copy of the padded verdict bytes (no communicator in a --gpus 1 run)"
    out["note"] = ("one GPU plays every rank of W = 1, 2, 4, 8 in turn: shard k of W as sharding.run_sharded cuts it (message / commitment boundaries; gossip "
                   "balanced by cost), verification + the gather stand-in; T(W) = slowest shard; predicted_speedup_W = T(1) / T(W).  Not an 8-GPU "
                   "measurement: no xGMI transfer, no second process")
    return out


def host_to_host(eng_cold, eng_warm, we, ws, n, steps):
    """SURVEY 8(d)'s own wording of the metric on the headline MIX: both batches of a step start in (pageable) host memory and their verdicts end in
    host memory, through the streaming queue (lamd_queue_*_batch -> pinned staging set, lamd_flush, lamd_wait): while the device works on one flush
    the host fills the next staging set and its H2D copies run under the kernels of the flushes before it (up to eight in flight).  Staging memcpy +
    H2D + verification + D2H inside the clock, which runs from an empty pipeline to the last verdict in host memory.  The verdict vectors are
    compared with construction AFTER the clock (the check is the bench's, not the path's).  -> (dict of legs, mismatches)"""
    depth = min(8, eng_cold.info()["queue_sets"] - 1)

    def host_mix(e, reps):
        pend, got = [], []
        t1 = time.perf_counter()
        for r in range(reps):
            for wl in (we, ws):
                if wl is we:
                    e.queue_ecdsa_batch(wl.cols[0], wl.cols[1], wl.cols[2])
                else:
                    e.queue_schnorr_batch(wl.cols[0], wl.cols[1], wl.cols[2])
                e.flush()
                pend.append(wl)
                if len(pend) == depth:
                    got.append((e.wait(cap=n), pend.pop(0)))
        while pend:
            got.append((e.wait(cap=n), pend.pop(0)))
        dt_ = time.perf_counter() - t1
        return dt_, sum(int((v != wl.expect).sum()) for v, wl in got)

    # the producer's form of the same loop: the rows already sit in the pinned staging sets (lamd_queue_reserve: a sidecar receives its
    # callers' triples straight into them), so a step is reserve + flush + wait -- H2D, verification and D2H inside the clock, no
    # host-side copy.  Each (staging set, kind) is filled the first time the loop meets it, i.e. during the priming pass.
    def host_mix_in_place(e, reps, filled):
        pend, got = [], []
        t1 = time.perf_counter()
        for r in range(reps):
            for wl in (we, ws):
                _, a, b_, c = e.queue_reserve(n, 65 if wl is we else 32)
                if a.ctypes.data not in filled:
                    filled.add(a.ctypes.data)
                    a[:] = wl.cols[0]
                    if wl is we:
                        b_[:], c[:] = wl.cols[1], wl.cols[2]
                    else:                        # BIP-340 columns are (msg, x-only key, signature)
                        c[:], b_[:] = wl.cols[1], wl.cols[2]
                e.flush()
                pend.append(wl)
                if len(pend) == depth:
                    got.append((e.wait(cap=n), pend.pop(0)))
        while pend:
            got.append((e.wait(cap=n), pend.pop(0)))
        dt_ = time.perf_counter() - t1
        return dt_, sum(int((v != wl.expect).sum()) for v, wl in got)
    hm, bad = {}, 0
    legs = [("cold_tables_rebuilt_every_flush", eng_cold, False), ("in_place_cold", eng_cold, True)]
    if eng_warm is not None and eng_warm is not eng_cold:
        legs += [("key_table_cache_on", eng_warm, False), ("in_place_key_table_cache_on", eng_warm, True)]
    for name, e, in_place in legs:
        seen = set()
        prime = 9 if not in_place else 9        # staging sets and per-lane workspaces are allocated on first use: nine sets x two kinds
        if in_place:
            host_mix_in_place(e, prime, seen)
            dtm, badm = host_mix_in_place(e, steps, seen)
        else:
            host_mix(e, prime)
            dtm, badm = host_mix(e, steps)
        hm[name] = {"verifies_per_s": 2 * steps * n / dtm, "ms_per_2M_step": dtm / steps * 1e3, "steps": steps, "mismatches": badm}
        bad += badm
    hm.update(rows_per_flush=n, flushes_in_flight=depth,
              note="1 M ECDSA-65 + 1 M BIP-340 per step from host memory to verdicts in host memory (289 MB in per step); in_place_* = rows written into "
                   "the pinned staging set by the producer (lamd_queue_reserve), no host-side copy inside the clock; compare with `value` (inputs resident in HBM)")
    return hm, bad


def cpu_baseline(we, ws, got_e, got_s, sample, n):
    """C0 = the reference's real CPU path (libsecp256k1 through dlopen, called as bitcoin/signature.c:188,425 call it) if this
    machine has the library -- else "unavailable"; C1 = the restated C oracle, 1 thread and all cores; C2 = OpenSSL ECDSA_do_verify +
    libsecp256k1's range / low-S rules, 1 thread.  Monotonic clock around each whole batch, verifies/s and ns per verification (the shape
    of onchaind/test/run-grind_feerate.c:146-154).  Every leg's verdicts must equal the GPU's on the rows it was given.
    -> (cpu_baseline dict, mismatches)"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc  # test infrastructure: the checker / CPU baseline only
    t_all = time.perf_counter()
    m = min(sample, n)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a cgroup CPU quota caps what those threads can really use
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    ce_cols = [np.ascontiguousarray(x[:m]) for x in we.cols]
    cs_cols = [np.ascontiguousarray(x[:m]) for x in ws.cols]

    def leg(fn_e, fn_s, rows_e, rows_s):
        """-> (dict, verdict mismatches against the GPU)"""
        t1 = time.perf_counter()
        ve = fn_e([c[:rows_e] for c in ce_cols]) if fn_e and rows_e else None
        t2 = time.perf_counter()
        vs = fn_s([c[:rows_s] for c in cs_cols]) if fn_s and rows_s else None
        t3 = time.perf_counter()
        bad = 0
        d = {}
        if ve is not None:
            bad += int((ve.astype(bool) != got_e[:rows_e]).sum())
            d.update(ecdsa_rows=rows_e, ecdsa_verifies_per_s=rows_e / (t2 - t1), ecdsa_ns_per_verify=(t2 - t1) / rows_e * 1e9)
        if vs is not None:
            bad += int((vs.astype(bool) != got_s[:rows_s]).sum())
            d.update(schnorr_rows=rows_s, schnorr_verifies_per_s=rows_s / (t3 - t2), schnorr_ns_per_verify=(t3 - t2) / rows_s * 1e9)
        rows = (rows_e if ve is not None else 0) + (rows_s if vs is not None else 0)
        secs = (t2 - t1 if ve is not None else 0) + (t3 - t2 if vs is not None else 0)
        d.update(value=rows / secs if secs else None, seconds=secs, gpu_vs_cpu_verdict_mismatches=bad)
        return d, bad
    orc.ecdsa_verify_batch(ce_cols[0][:64], ce_cols[1][:64], ce_cols[2][:64], 65, cores)  # table init outside the timed part
    legs = {}
    one = max(1, min(m, 10_000))
    legs["C1_oracle_1_thread"], b1 = leg(lambda c: orc.ecdsa_verify_batch(c[0], c[1], c[2], 65, 1), lambda c: orc.schnorr_verify_batch(c[0], c[1], c[2], 1), one, one)
    legs["C1_oracle_all_cores"], b2 = leg(lambda c: orc.ecdsa_verify_batch(c[0], c[1], c[2], 65, cores), lambda c: orc.schnorr_verify_batch(c[0], c[1], c[2], cores), m, m)
    legs["C1_oracle_1_thread"]["threads"], legs["C1_oracle_all_cores"]["threads"] = 1, cores
    ossl_rows = max(1, min(m, 5_000))
    legs["C2_openssl_ecdsa_do_verify_plus_rules_1_thread"], b3 = leg(lambda c: orc.ossl_ecdsa_verify_rules_batch(c[0], c[1], c[2], 65), None, ossl_rows, 0)
    legs["C2_openssl_ecdsa_do_verify_plus_rules_1_thread"]["threads"] = 1
    secp = orc.libsecp_available()
    if secp:
        legs["C0_libsecp256k1_1_thread"], b0 = leg(lambda c: orc.libsecp_ecdsa_verify_batch(c[0], c[1], c[2], 65),
                                                   lambda c: orc.libsecp_schnorr_verify_batch(c[0], c[1], c[2]), min(m, 100_000), min(m, 100_000))
        legs["C0_libsecp256k1_1_thread"].update(threads=1, library=secp)
    else:
        legs["C0_libsecp256k1_1_thread"], b0 = "unavailable: no libsecp256k1.so can be dlopen()ed on this node (the reference's copy is an empty submodule)", 0
    cm = b0 + b1 + b2 + b3
    ac, c1 = legs["C1_oracle_all_cores"], legs["C1_oracle_1_thread"]
    ref = legs["C0_libsecp256k1_1_thread"] if secp else None
    cb = {"value": ref["value"] if ref else ac["value"], "unit": "verifies/s", "cores": 1 if ref else cores, "kind": "reference" if ref else "port",
          "sample": ("libsecp256k1 via dlopen (%s), 1 thread, first %d ECDSA + %d Schnorr rows" % (os.path.basename(str(secp)), ref["ecdsa_rows"], ref.get("schnorr_rows", 0))) if ref else
                    ("first %d ECDSA + %d Schnorr rows of the batch, %d threads, restated C oracle (not libsecp256k1: absent here)" % (m, m, cores)),
          "note": None if ref else "restated oracle, 2-4x slower per verify than libsecp256k1 (SURVEY 6): GPU/CPU ratios are flattered by that",
          "C1_1thread": c1["value"], "C1_all_cores": ac["value"], "C2": legs["C2_openssl_ecdsa_do_verify_plus_rules_1_thread"]["value"],
          "C0": ref["value"] if ref else "unavailable", "ns_per_verify_1thread": 1e9 / c1["value"] if c1["value"] else None,
          "ecdsa_verifies_per_s": (ref or ac)["ecdsa_verifies_per_s"], "schnorr_verifies_per_s": (ref or ac).get("schnorr_verifies_per_s"),
          "host_cores": cores, "libsecp256k1_found": secp, "legs": legs,
          "legs_note": "BASELINE.md 3: C0 the reference's library (if present), C1 this repo's restated oracle, C2 OpenSSL's generic secp256k1 + "
                       "libsecp256k1's acceptance rules; monotonic clock around each batch; every leg's verdicts compared with the GPU's",
          "gpu_vs_cpu_verdict_mismatches": cm, "seconds": time.perf_counter() - t_all}
    return cb, cm, 2 * m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", "--n", dest="n", type=int, default=1_000_000, help="rows per kind per rank (1 M = BASELINE configs[1], [2])")
    ap.add_argument("--cpu-sample", type=int, default=400_000, help="rows per kind timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--extras", action="store_true", help="also: warm key-table cache, latency, PCIe-inclusive calls, configs[3]/[4] on one GPU, gossip ingest flood, "
                                                          "fee grind, key recovery, key-reuse sweep (tools/bench_extras.py; all of it lands in bench_details.json)")
    ap.add_argument("--no-scaling", action="store_true", help="skip the one-GPU strong-scaling sweep (config.predicted_speedup_8) / the sharded configs of a multi-rank run")
    ap.add_argument("--no-h2h", action="store_true", help="skip the host-to-host legs (config.value_host_to_host)")
    ap.add_argument("--div", type=int, default=1, help="divide the sizes of configs[3]/[4] (quick runs, the CPU tests)")
    ap.add_argument("--details", default=None, help="where bench_details.json goes (default: next to bench.py, and gpurun_out/)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="ONLY the loop `roofline.frac` is computed from (the cold loop with chained ecmult launches, warm-up + steps): every "
                         "k_ecmult_keyed<false, 3> launch of the process is one of that loop's, so the per-kernel average of `rocprofv3 "
                         "--kernel-trace --stats` over this command is directly comparable with roofline.avg_launch_ms")
    ap.add_argument("--ab", action="store_true", help="A/B runs: the cold loop, the chained loop and the isolated calls only")
    ap.add_argument("--steady-steps", type=int, default=250,
                    help="when --steps gives a timed region under ~1 s: steps of an extra, longer cold loop reported as `steady_state` (0 = skip)")
    ap.add_argument("--h2h-steps", type=int, default=60, help="steps of each host-to-host leg")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        # a plain `python bench.py --gpus N`: become the launcher -- one process per GPU over RCCL, rendezvous on 127.0.0.1
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    # stdout carries exactly ONE line, the JSON: everything libraries print there (RCCL's version banner on the first
    # collective) goes to stderr instead
    json_fd = os.dup(1)
    os.dup2(2, 1)
    t_start = time.perf_counter()
    clock = Clock()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    plat = load_platform(local_rank)
    device = plat.device
    Engine, workload = plat.Engine, plat.workload
    from lightning_amd import sharding
    # launched by torch.distributed.run (RANK set): the collective path runs even with one rank, so that a 1-GPU box can test it
    multi = world > 1 or ("RANK" in os.environ and os.environ.get("LAMD_BENCH_GATHER", "0") == "1")
    # the engine first: its streams take their hardware queues before RCCL creates its own (the other order costs ~8 %:
    # measured with one rank forced through the collective path, 180 vs 196 M verifies/s)
    # `eng_cold` rebuilds every key's comb table in every call (LAMD_CACHE=0: what a stateless library does, and what `value` is measured on).
    # It is ALONE in the process while `value` is measured (a serving process holds one engine): the default engine (key-table cache on) exists
    # only under --extras and is created after the cold legs -- two engines are 20 streams on 16 hardware queues, and the second one's share cost
    # the cold loop 3-5 %.  With more than one rank ONLY the cold engine exists: one engine = one 11 GiB G table per rank.
    with clock("engine_init"):
        os.environ["LAMD_CACHE"] = "0"
        eng_cold = Engine(local_rank)
        del os.environ["LAMD_CACHE"]
    eng = eng_cold
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the only collective is an all-gather of <= 1 MB of verdict bytes per rank: one or two RCCL channels carry it, and every
        # channel RCCL opens beyond that is a stream competing with the engine's lanes for hardware queues
        os.environ.setdefault("NCCL_MIN_NCHANNELS", "1")
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "2")
        with clock("process_group"):
            if plat.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device(device))
            else:
                dist.init_process_group(plat.backend)
    eng_cold.set_timing(True)

    n = args.n
    with clock("workload"):
        we = workload.make_ecdsa(eng, n, seed=workload.SEED_CFG2 + rank, nkeys=65536, publen=65, device=device)
        ws = workload.make_schnorr(eng, n, seed=workload.SEED_CFG3 + rank, nkeys=65536, device=device)
    # collective path: one all-gather per batch kind and step, issued one step late (after the calls of the next step: see step())
    ok_all_e = torch.zeros(world * n, dtype=torch.uint8, device=device) if multi else None
    ok_all_s = torch.zeros(world * n, dtype=torch.uint8, device=device) if multi else None

    kernel_ms = {"ecdsa": [], "schnorr": []}
    keyed = {}

    tstream = plat.stream_ptr()

    # multi-rank: verdicts rotate over `nbuf` buffers per kind so that the all-gather of a step's batch (on torch's stream, ordered
    # after that call by device-side events) can still be reading buffer k while later steps write the others.  Every dependency is per call,
    # by events, with no host synchronisation (sharding.LateGather).  Six buffers: with two, step k + 2 waited for the gather of
    # step k, a small copy kernel that takes 0.3-1 ms to get its waves onto the saturated chip (2 / 4 / 6 buffers: 0.93 / 0.987 / 0.994 of the plain loop on one rank).
    nbuf = max(2, min(8, int(os.environ.get("LAMD_BENCH_GATHER_BUFS", "6"))))
    ok_e = [we.d_ok] + [torch.zeros_like(we.d_ok) for _ in range(nbuf - 1)] if multi else [we.d_ok]
    ok_s = [ws.d_ok] + [torch.zeros_like(ws.d_ok) for _ in range(nbuf - 1)] if multi else [ws.d_ok]

    lg = {}      # per engine: sharding.LateGather (the marks are the engine's)
    def late_gather(eng):
        if id(eng) not in lg:
            lg[id(eng)] = sharding.LateGather(eng, ("e", "s"), {"e": ok_e, "s": ok_s}, {"e": ok_all_e, "s": ok_all_s}, tstream,
                                              dist.all_gather_into_tensor, plat.new_event)
        return lg[id(eng)]
    stepno = [0]
    def step(eng, poison=False):
        # no host synchronisation inside a step: successive calls rotate over the engine's lanes, so the front end (key
        # de-duplication, table building) of one batch runs under the ecmult kernels of the batches before it
        b = stepno[0] % len(ok_e)
        stepno[0] += 1
        g = late_gather(eng) if multi else None
        if poison:   # the LAST timed step writes into poisoned verdict buffers: a launch that wrote nothing cannot pass the parity check
            ok_e[b].fill_(7)
            ok_s[b].fill_(7)
            eng.wait_stream(tstream)
        if multi:
            g.before_call("e", b)
        eng.verify_ecdsa_device(we.dev[0], we.dev[1], we.dev[2], ok_e[b])
        if multi:
            g.after_call("e", b)
            g.before_call("s", b)
        eng.verify_schnorr_device(ws.dev[0], ws.dev[1], ws.dev[2], ok_s[b])
        if multi:
            g.after_call("s", b)
            g.end_step(b)      # issues the all-gathers of the PREVIOUS step

    def record_kernel_times(eng):
        # HIP events recorded on the lanes' own streams around each kernel group of the LAST step inside the timed region
        # (ECDSA ran on one lane, BIP-340 on the other), read after the closing fence
        for lane in range(eng.info()["lanes"]):
            inf = eng.info(lane)
            which = "schnorr" if inf["last_mode"] else "ecdsa"
            kernel_ms[which].append(inf["last_kernel_ms"])
            keyed[which] = (inf["last_keyed"], inf["last_unique_keys"])

    def fence(eng):
        if multi:
            late_gather(eng).flush()       # the last step's all-gathers belong to the timed region
            dist.barrier()
        plat.synchronize()
        eng.synchronize()

    launch_ms = {}

    def timed(eng):
        eng.auto_order = False   # the inputs were generated and synchronised before the loop: no per-call ordering after torch's stream
        stepno[0] = 0
        if multi:
            late_gather(eng).reset()
        for _ in range(args.warmup):
            step(eng)
        fence(eng)
        eng.set_timing(True)     # restart the per-launch duration sums of the dominant kernel: they cover exactly the timed steps
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(eng, poison=(k == args.steps - 1))
        fence(eng)
        dt = time.perf_counter() - t0
        eng.auto_order = True
        # HIP events right around every table-driven ecmult launch of the timed steps, on the lane stream that launched it
        launch_ms[id(eng)] = [[sum(eng.info(l)["keyed_ecmult_ms_sum"][m] for l in range(eng.info()["lanes"])),
                               sum(eng.info(l)["keyed_ecmult_launches"][m] for l in range(eng.info()["lanes"]))] for m in (0, 1)]
        if multi:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        got = (ok_e[(stepno[0] - 1) % len(ok_e)].cpu().numpy(), ok_s[(stepno[0] - 1) % len(ok_s)].cpu().numpy())
        bad = int((got[0] != we.expect.astype(np.uint8)).sum() + (got[1] != ws.expect.astype(np.uint8)).sum())
        return dt, bad

    def rows_per_kind(eng):
        # rows the table-driven launch of the last call of each kind really carried (rows whose signature scalars are certain to fail the
        # preparation were rejected by the row-list builders, rows under rare / unparsable keys went to the ladder list)
        r = {}
        for lane in range(eng.info()["lanes"]):
            inf = eng.info(lane)
            if inf["last_hot_rows"]:
                r[int(inf["last_mode"])] = int(inf["last_hot_rows"])
        return r

    # the headline first: cold, every table rebuilt in every call
    full = not args.roofline_only
    default_legs = full and not args.ab
    dt, mism_cold, steady = float("nan"), 0, None
    if full:
        with clock("headline_loop"):
            dt, mism_cold = timed(eng_cold)
        record_kernel_times(eng_cold)
        if args.steady_steps > 0 and dt < 1.0 and not multi:
            # the driver's --steps 20 is a 0.17 s region: the same loop once more over a region of seconds, reported beside `value`
            with clock("steady_loop"):
                k_steps, args.steps = args.steps, args.steady_steps
                dt_st, mism_st = timed(eng_cold)
                args.steps = k_steps
            steady = {"value": world * 2 * n * args.steady_steps / dt_st, "unit": "verifies/s", "steps": args.steady_steps, "seconds": dt_st,
                      "ms_per_step": dt_st / args.steady_steps * 1e3, "mismatches": mism_st,
                      "note": "the timed loop of `value` again over a region of seconds (box-to-box spread of the short region: +-3 %)"}
            mism_cold += mism_st
    # THE ROOFLINE LOOP: the same cold loop with the large ecmult launches chained one after the other (lamd_set_ecmult_chain): ONE such launch in
    # flight at any time, in the company of the other lanes' front-end kernels only, so the HIP-event pair around a launch brackets that launch
    # (in the default mode two or three of them overlap and every bracket measures its neighbours too).  roofline.frac comes from here; `value`
    # from the default mode above (2-3 % more throughput: the tail of one launch filled by the head of the next).
    chained = None
    if not multi:
        with clock("chained_loop"):
            lm_cold = launch_ms.get(id(eng_cold))
            eng_cold.set_ecmult_chain(True)
            dt_ch, mism_ch = timed(eng_cold)
            chained = {"dt": dt_ch, "mismatches": mism_ch, "lm": launch_ms[id(eng_cold)], "rows": rows_per_kind(eng_cold)}
            if not full:
                record_kernel_times(eng_cold)
                dt = dt_ch
            eng_cold.set_ecmult_chain(False)
            if lm_cold is not None:
                launch_ms[id(eng_cold)] = lm_cold
            mism_cold += mism_ch
    # the same kernels once more, one call at a time (nothing else on the GPU): the isolated durations
    isolated = {"ecdsa": [], "schnorr": []}
    eng_cold.set_timing(True)
    rows_in_launch = n
    with clock("isolated_calls"):
        for _ in range(2 if full else 0):
            eng_cold.verify_ecdsa_device(we.dev[0], we.dev[1], we.dev[2], we.d_ok)
            eng_cold.synchronize()
            isolated["ecdsa"].append(eng_cold.info()["last_kernel_ms"])
            rows_in_launch = int(eng_cold.info()["last_hot_rows"]) or n
            eng_cold.verify_schnorr_device(ws.dev[0], ws.dev[1], ws.dev[2], ws.d_ok)
            eng_cold.synchronize()
            isolated["schnorr"].append(eng_cold.info()["last_kernel_ms"])
    lanes_n = eng_cold.info()["lanes"]
    iso_launch = [sum(eng_cold.info(l)["keyed_ecmult_ms_sum"][m] for l in range(lanes_n)) /
                  max(1, sum(eng_cold.info(l)["keyed_ecmult_launches"][m] for l in range(lanes_n))) for m in (0, 1)] if full else [0.0, 0.0]
    if not full:
        rows_in_launch = (chained or {}).get("rows", {}).get(0, n)
        isolated = {k: [[float("nan")] * 4] for k in isolated}
    # the roofline's denominator at the clock long launches sustain: a dependency-free v_mad_u64_u32 stream on every SIMD in launches of >= 4 ms,
    # at the ecmult kernel's occupancy (3 waves per SIMD) and at 8; the GPU is idle around it (everything above is synchronised)
    peak_sust = None
    with clock("mul32_peak"):
        try:
            eng_cold.synchronize()
            p3 = eng_cold.mul32_peak(3, 4.0, 4)
            p8 = eng_cold.mul32_peak(8, 4.0, 4)
            peak_sust = {"waves3": {"Tmul32_per_s": p3[0] / 1e12, "launch_ms": p3[1], "memtime_per_realtime": p3[2]},
                         "waves8": {"Tmul32_per_s": p8[0] / 1e12, "launch_ms": p8[1], "memtime_per_realtime": p8[2]}}
        except Exception as e:   # an older library without the entry point (LAMD_LIB_PATH experiments)
            peak_sust = {"error": repr(e)}
    for k in isolated:
        if not kernel_ms[k]:          # LAMD_LANES=1: only the last call's events survive the timed region
            kernel_ms[k] = isolated[k]
            keyed.setdefault(k, keyed.get("schnorr", (0, 0)))
    # ---- parity on every row of this rank (verdicts known by construction): the poisoned last step of the timed loops, and
    # the isolated calls just made
    got_e = we.d_ok.cpu().numpy().astype(bool)
    got_s = ws.d_ok.cpu().numpy().astype(bool)
    mism_iso = int((got_e != we.expect).sum() + (got_s != ws.expect).sum()) if full else 0
    mism = mism_iso + mism_cold
    if multi:
        # every rank must hold every other rank's verdicts after the all-gather: all ranks' expected verdicts are gathered too and compared WHOLE
        exp_all = [torch.zeros(world * n, dtype=torch.uint8, device=device) for _ in range(2)]
        dist.all_gather_into_tensor(exp_all[0], torch.from_numpy(we.expect.astype(np.uint8)).to(device))
        dist.all_gather_into_tensor(exp_all[1], torch.from_numpy(ws.expect.astype(np.uint8)).to(device))
        plat.synchronize()
        mism += int((ok_all_e != exp_all[0]).sum().item() + (ok_all_s != exp_all[1]).sum().item())
        m = torch.tensor([mism], dtype=torch.int64, device=device)
        dist.all_reduce(m)
        mism = int(m.item())

    # ---- the two "8 GPUs" configs of BASELINE.json as ONE job split over the ranks (all ranks take part in the collectives)
    sharded = None
    if multi and default_legs and not args.no_scaling:
        with clock("sharded_configs"):
            sharded = sharded_configs(plat, eng_cold, rank, world, tstream, args.div)
        mism += sharded["mismatches_summed_over_ranks"] if rank == 0 else 0

    out = None
    if rank == 0:
        total = world * 2 * n * args.steps
        value = total / dt
        # ---- roofline of the dominant kernel, k_ecmult_keyed<false, 3>.  Numerator: the 32x32->64 multiply-adds the kernel's
        # algorithm EXECUTES per row (W_EXEC) x the rows its launches carried.  Denominator: the average duration of those launches, HIP events
        # on the launching lane's stream right around every one of them, in the CHAINED loop (one such launch in flight at a time: the brackets do
        # not overlap, their sum per step is below the step time, and `rocprofv3 --kernel-trace --stats -- python bench.py --roofline-only`
        # -- the same loop and nothing else -- gives the same per-kernel average: profiles/).  ECDSA and BIP-340 launches are the same kernel
        # (their acceptance tests differ by two field multiplications of ~650), so the average is over both kinds, as rocprofv3's is.
        teeth = int(keyed.get("ecdsa", (0, 0))[0])
        g_windows = g_windows_of(int(eng_cold.info()["gtable_bytes"]))
        pairs_first = os.environ.get("LAMD_PAIRS", "0") == "1"          # experiment knob: k_ecmult_keyed_pairs for large calls (the default is k_ecmult_keyed<false, 3>)
        resident_lanes = int(eng_cold.info()["compute_units"]) * 3 * 4 * 64
        w_exec_t = w_exec_table(g_windows, pairs_first, min(6.0, max(1.0, rows_in_launch / resident_lanes)), os.environ.get("LAMD_BENCH_G_RUN", "xyzz") == "xyzz")
        w_exec = w_exec_t.get(teeth, w_exec_t[0])
        kernel_name = "k_ecmult_keyed_pairs<3>" if pairs_first and teeth else ("k_ecmult_keyed<false, 3>" if teeth else "k_ecmult<3>")
        # the peak the fraction is priced against: the multiply-add's SUSTAINED issue rate (launches >= 4 ms, measured in this process a moment ago),
        # the better of the kernel's own occupancy and full occupancy; P_MUL32 (a sub-millisecond micro-benchmark of round 1: boost clock) stays beside it
        p_sust = P_MUL32
        if peak_sust and "waves3" in peak_sust:
            p_sust = max(peak_sust["waves3"]["Tmul32_per_s"], peak_sust["waves8"]["Tmul32_per_s"]) * 1e12
        lm_ov = launch_ms.get(id(eng_cold)) if full else None          # default mode (the launches overlap): reported, never the roofline
        pipeline = {"ms": dt / args.steps * 1e3, "achieved": 2 * w_exec * rows_in_launch / (dt / args.steps) / 1e12,
                    "frac": 2 * w_exec * rows_in_launch / (dt / args.steps) / p_sust, "frac_vs_boost_peak": 2 * w_exec * rows_in_launch / (dt / args.steps) / P_MUL32,
                    "note": "both table-driven launches' executed multiply-adds of a step / the step time of the loop `value` is measured on (everything "
                            "else a step does -- key tables, scalar preparation, de-duplication -- counts as lost time here)"}
        if chained is not None and chained["lm"][0][1] + chained["lm"][1][1]:
            cl = chained["lm"]
            n_l = int(cl[0][1] + cl[1][1])
            t_ecmult = (cl[0][0] + cl[1][0]) / n_l * 1e-3                 # seconds per launch, both kinds
            rows_k = [chained["rows"].get(m, rows_in_launch) for m in (0, 1)]
            rows_avg = (rows_k[0] * cl[0][1] + rows_k[1] * cl[1][1]) / n_l
            # (a lane keeps at most lamd_ctx::KEV event pairs per interval: over a long loop fewer launches are timed than run -- the step holds one
            # launch of each kind, so its launches sum to the two averages)
            sum_per_step = (cl[0][0] / cl[0][1] if cl[0][1] else 0.0) + (cl[1][0] / cl[1][1] if cl[1][1] else 0.0)
            roof_mode = {"mode": "chained", "launches_timed": n_l, "avg_launch_ms": t_ecmult * 1e3,
                         "avg_launch_ms_ecdsa": cl[0][0] / cl[0][1] if cl[0][1] else None, "avg_launch_ms_schnorr": cl[1][0] / cl[1][1] if cl[1][1] else None,
                         "rows_in_launch": rows_avg, "rows_in_launch_by_kind": {"ecdsa": rows_k[0], "schnorr": rows_k[1]},
                         "sum_of_launch_ms_per_step": sum_per_step, "ms_per_step": chained["dt"] / args.steps * 1e3,
                         "sum_of_launches_le_step": bool(sum_per_step <= chained["dt"] / args.steps * 1e3),
                         "verifies_per_s": world * 2 * n * args.steps / chained["dt"], "mismatches": chained["mismatches"]}
            achieved = w_exec * rows_avg / t_ecmult
        else:
            # no chained loop (collective path: more than one rank): the pipeline figure, which needs no per-launch bracket
            t_ecmult, rows_avg = dt / args.steps / 2, rows_in_launch
            roof_mode = {"mode": "pipeline", "launches_timed": 2 * args.steps, "avg_launch_ms": t_ecmult * 1e3, "rows_in_launch": rows_avg,
                         "sum_of_launch_ms_per_step": dt / args.steps * 1e3, "ms_per_step": dt / args.steps * 1e3, "sum_of_launches_le_step": True}
            achieved = w_exec * rows_avg / t_ecmult
        # HBM traffic per launch: separate rocprofv3 --pmc passes of the roofline loop (tools/pmc_run.sh -> profiles/pmc_latest.json).  FETCH_SIZE on
        # gfx950 counts memory-side read requests at 64 B each; for this kernel's access pattern (49 random table entries of 64-96 B per row) the
        # counter is calibrated on a gather of known size (tools/microbench.hip `gather`; factor and source in pmc_latest.json)
        traffic = traffic_raw = traffic_src = fetch_factor = None
        valu_issue = tables = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))["k_ecmult_ecdsa_1M"]
            if n == 1_000_000:
                fetch_factor = float(pm.get("fetch_size_factor", 1.0))
                traffic_raw = pm["hbm_bytes_per_launch"]
                traffic = (pm["fetch_kib"] * fetch_factor + pm["write_kib"]) * 1024.0
                traffic_src = pm["source"]
            valu_issue = {"wave_instr_per_simd_cycle": pm["valu_issue_per_simd_cycle"], "saturated_at": 0.25,
                          "frac": pm["valu_issue_per_simd_cycle"] / 0.25, "valu_instr_per_verify": pm["valu_insts_per_verify"],
                          "source": pm["source"]}
            if pm.get("step_valu_wave_instr_total") and n == 1_000_000:
                # THE WHOLE STEP against the issue roofline (round 6): every kernel of a step competes for the same VALU issue slots, so the step is priced
                # as one thing -- VALU wave-instructions of all its kernels (PMC, the same command) x 4 cycles / (SIMDs x clock x step time).  The clock is
                # the multiply-add probe's (measured in this process; the kernels themselves run ~7 % below it), so the fraction is a lower bound.
                simds = int(eng_cold.info()["compute_units"]) * 4
                clk = p_sust / (int(eng_cold.info()["compute_units"]) * 64)
                tot = float(pm["step_valu_wave_instr_total"])
                sv = pm.get("step_valu_wave_instr", {})
                tab = sum(v for k, v in sv.items() if k.startswith("k_kc_") or k.startswith("k_keys_bases"))
                valu_issue["step"] = {"valu_wave_instr_per_step": tot, "issue_ms_at_probe_clock": tot * 4 / simds / clk * 1e3, "ms_per_step": dt / args.steps * 1e3,
                                      "frac": tot * 4 / simds / clk / (dt / args.steps), "probe_clock_GHz": clk / 1e9,
                                      # ... and at the clock the dominant kernel itself runs at (GRBM_GUI_ACTIVE / 8 / t in the counter pass: power holds it ~7 %
                                      # below the dependency-free probe): how full the issue port is while the step runs
                                      "kernel_clock_GHz": pm.get("shader_clock_GHz"),
                                      "frac_at_kernel_clock": (tot * 4 / simds / (pm["shader_clock_GHz"] * 1e9) / (dt / args.steps)) if pm.get("shader_clock_GHz") else None,
                                      "share_ecmult": sum(v for k, v in sv.items() if k.startswith("k_ecmult_keyed<false")) / tot,
                                      "share_key_tables": tab / tot,
                                      "note": "all kernels of a step: VALU wave-instructions x 4 cycles / (1024 SIMDs x clock x step time)"}
                # the key-table kernels priced like the dominant one (VERDICT r05 "next" 3): instructions per key, lanes, issue fraction of the isolated stage
                keys_step = sum(int(v[1]) for v in keyed.values()) if keyed else 0
                if keys_step and isolated.get("ecdsa") and isolated.get("schnorr"):
                    wb = sum(v for k, v in sv.items() if k.startswith("k_keys_bases"))
                    wf = sum(v for k, v in sv.items() if k.startswith("k_kc_"))
                    iso_tab_ms = float(np.mean(np.array(isolated["ecdsa"]), axis=0)[1] + np.mean(np.array(isolated["schnorr"]), axis=0)[1])
                    kclk = (pm.get("shader_clock_GHz") or clk / 1e9) * 1e9
                    tables = {"kernels": "k_keys_bases_both (key parse + the 114-doubling chain, one lane per key) + k_kc_finish_both (Gray-code chains of mixed additions, Z products, "
                                         "rescale: four lanes per 7-tooth key)",
                              "distinct_keys_per_step": keys_step,
                              "valu_wave_instr_per_step": {"bases": wb, "finish": wf},
                              "valu_lane_instr_per_key": {"bases": wb * 64 / keys_step, "finish": wf * 64 / keys_step, "both": (wb + wf) * 64 / keys_step},
                              "valu_lane_instr_per_verify_at_this_reuse": (wb + wf) * 64 / (2.0 * n),
                              "lanes_per_step": {"bases": keys_step, "finish": 4 * keys_step}, "lane_slots_of_the_chip_at_4_waves_per_simd": simds * 4 * 64,
                              "waves_per_call_over_simds": {"bases": keys_step / 2 / 64 / simds, "finish": 4 * keys_step / 2 / 64 / simds},
                              "isolated_stage_ms_per_step": iso_tab_ms,
                              "issue_frac_isolated": (wb + wf) * 4 / simds / kclk / (iso_tab_ms * 1e-3),
                              "share_of_step_valu": tab / tot,
                              "note": "counts from the PMC passes of the cold loop; isolated_stage_ms = the keys_and_tables brackets of one ECDSA and one BIP-340 call alone on the chip "
                                      "(they hold the de-duplication's tail too); issue_frac_isolated = instructions x 4 cycles / (SIMDs x kernel clock x that time): "
                                      "alone, the stage is latency-bound (waves_per_call_over_simds: about 1.2 and 4.8 waves per SIMD, one dependent chain each) -- inside the loop its instructions fill "
                                      "the slots ecmult leaves (valu_issue.step)"}
        except Exception:
            pass
        iso_ms = iso_launch[0] or float(np.mean(np.array(isolated["ecdsa"]), axis=0)[2])
        algo_bytes = BYTES_ECDSA65 * n
        ke = np.mean(np.array(kernel_ms["ecdsa"]), axis=0)      # prep, keys, ecmult [ms]
        ks = np.mean(np.array(kernel_ms["schnorr"]), axis=0)
        out = {
            "metric": "signature verifies/sec (ECDSA+Schnorr mix)", "value": value, "unit": "verifies/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (256-bit modular integer)",
            "data": "synthetic",
            "config": {"workload": "configs[1]+configs[2]: %d ECDSA (65-byte keys, 65536 distinct) + %d BIP-340 Schnorr per GPU per step, "
                                   "90%% valid / 10%% invalid, inputs resident in HBM%s" % (n, n, "" if full else "; --roofline-only: the chained loop"),
                       "rows_per_gpu_per_step": 2 * n, "parallelism": "shard-by-row x%d, RCCL all-gather of verdicts" % world,
                       "key_table_cache": "off (tables rebuilt every call)"},
            "steady_state": steady,
            "rates": {"ecdsa65_verifies_per_s_1gpu": n / (ke.sum() * 1e-3), "schnorr_verifies_per_s_1gpu": n / (ks.sum() * 1e-3),
                      "kernel_ms_ecdsa": {"prep": ke[0], "keys_and_tables": ke[1], "ecmult": ke[2], "parity_stage": ke[3]},
                      "kernel_ms_schnorr": {"prep": ks[0], "keys_and_tables": ks[1], "ecmult": ks[2], "parity_stage": ks[3]},
                      "kernel_ms_note": "stage brackets (HIP events on each lane's stream, last timed step): a stage's interval includes its waits for the "
                                        "lane's side streams and the other lanes' share of the chip -- the dominant kernel's own launch duration is roofline.avg_launch_ms; "
                                        "*_isolated = one call at a time",
                      "kernel_ms_ecdsa_isolated": dict(zip(("prep", "keys_and_tables", "ecmult", "parity_stage"), np.mean(np.array(isolated["ecdsa"]), axis=0).tolist())),
                      "kernel_ms_schnorr_isolated": dict(zip(("prep", "keys_and_tables", "ecmult", "parity_stage"), np.mean(np.array(isolated["schnorr"]), axis=0).tolist())),
                      "keyed_path": {k: {"per_key_tables": bool(v[0]), "distinct_keys": int(v[1])} for k, v in keyed.items()}},
            "roofline": dict(roof_mode, **{
                "kernel": "%s (1 M-row ECDSA-65 / BIP-340 launches)" % ("%s: %d-tooth signed comb, bare formulas%s" % (kernel_name, teeth, ", pairs first" if pairs_first else "") if teeth else "k_ecmult"),
                "bound": "valu-int32-mul (not hbm, not mfma)",
                "achieved": achieved / 1e12, "peak": p_sust / 1e12, "unit": "Tmul32/s", "frac": achieved / p_sust,
                "peak_sustained": p_sust / 1e12, "peak_boost": P_MUL32 / 1e12, "frac_vs_boost_peak": achieved / P_MUL32,
                "peak_note": "peak = peak_sustained: dependency-free v_mad_u64_u32 on every SIMD in launches of >= 4 ms, measured in THIS process after the timed "
                             "loops (lamd_debug_mul32_peak; the better of 3 and 8 waves per SIMD); peak_boost = 36.9: the round-1 micro-benchmark's sub-millisecond launches",
                "peak_sustained_detail": peak_sust,
                # the whole step against the peak: both table-driven launches' executed multiply-adds / ms_per_step (key tables, scalar preparation,
                # de-duplication and every stall count as lost time)
                "frac_step": pipeline["frac"],
                "executed_mul32_per_verify": w_exec, "g_table_windows": g_windows,
                "rows_note": "of a batch's %d rows: the others were decided before the ecmult (early reject of signatures whose scalars cannot pass "
                             "the preparation: r, s range and low-S; keys that do not parse; rows under rare keys take the ladder kernel)" % n,
                "timing": "HIP event pair on the launching lane's stream right before and after every " + kernel_name + " launch of the timed "
                          "steps of the CHAINED cold loop (lamd_set_ecmult_chain(1): a launch waits for the one submitted before it, so ONE is in flight "
                          "at a time and a bracket holds that launch plus the other lanes' front-end kernels).  `rocprofv3 --kernel-trace --stats -- "
                          "python bench.py --roofline-only` runs this loop only: its per-kernel average is the same quantity",
                "frac_isolated": (w_exec * rows_in_launch / (iso_ms * 1e-3) / p_sust) if full else None,
                "isolated": None if not full else {
                    "launch_ms": iso_ms, "launch_ms_schnorr": iso_launch[1] or None,
                    "achieved": w_exec * rows_in_launch / (iso_ms * 1e-3) / 1e12, "frac": w_exec * rows_in_launch / (iso_ms * 1e-3) / p_sust,
                    "note": "one call at a time, nothing else on the GPU (measured right after the timed loops)"},
                "overlapped": None if not (lm_ov and lm_ov[0][1]) else {
                    "avg_launch_ms": lm_ov[0][0] / lm_ov[0][1], "avg_launch_ms_schnorr": (lm_ov[1][0] / lm_ov[1][1]) if lm_ov[1][1] else None,
                    "sum_of_launch_ms_per_step": (lm_ov[0][0] / lm_ov[0][1]) + ((lm_ov[1][0] / lm_ov[1][1]) if lm_ov[1][1] else 0.0),
                    "note": "the same brackets in the DEFAULT mode (the loop `value` is measured on): 1.1-1.5 such launches are in flight at any time, every "
                            "bracket holds its neighbours' share too and their sum exceeds the step time -- not a kernel duration"},
                "pipeline": pipeline,
                "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                "traffic_over_algorithmic": (traffic / algo_bytes) if traffic else None,
                "algorithmic_bytes_per_launch": algo_bytes,
                "traffic_detail": None if traffic is None else {"fetch_size_factor": fetch_factor, "uncorrected_bytes": traffic_raw,
                                                               "achieved_TBs_at_avg_launch": traffic / t_ecmult / 1e12, "hbm_peak_frac": traffic / t_ecmult / 1e9 / HBM_PEAK_GBS},
                "traffic_source": traffic_src,
                # the multiplier instructions are 60 % of the kernel's VALU instructions and the VALU issue port is the limit
                "valu_issue": valu_issue,
                "valu_instr_per_verify": valu_issue["valu_instr_per_verify"] if valu_issue else None,
                "valu_issue_frac": valu_issue["frac"] if valu_issue else None,
                "valu_issue_frac_step": valu_issue["step"]["frac"] if valu_issue and "step" in valu_issue else None,
                "valu_share_key_tables": valu_issue["step"]["share_key_tables"] if valu_issue and "step" in valu_issue else None,
                "tables": tables,
                # SURVEY 8(d)'s implementation-independent yardstick (1.32e5 mul32 for a generic ECDSA verification) over the same time: NOT a
                # utilisation figure (the combs execute 2.2x fewer multiplies than the yardstick's generic algorithm)
                "survey_yardstick": {"mul32_per_verify": W_ECDSA65, "yardstick_Tmul32_per_s_in_loop": W_ECDSA65 * n / t_ecmult / 1e12,
                                     "note": "rate at which SURVEY 8(d)'s generic-algorithm multiplies would have to run to finish in the same time; not a fraction of peak"},
                "hbm": {"algorithmic_bytes_per_launch": algo_bytes, "achieved_GBs": algo_bytes / t_ecmult / 1e9,
                        "peak_GBs": HBM_PEAK_GBS, "frac": algo_bytes / t_ecmult / 1e9 / HBM_PEAK_GBS}}),
            "parity": {"rows_checked": world * 2 * n, "mismatches": mism, "mismatches_by_leg": {"cold_loops": mism_cold, "isolated_calls": mism_iso},
                       "against": "verdicts known by construction (all rows; the last timed step of every loop writes into poisoned verdict buffers)"},
        }
        if sharded is not None:
            out["sharded_configs"] = {k: v for k, v in sharded.items() if isinstance(v, dict)}
        # ---- the metric as SURVEY 8(d) words it: host buffers in -> verdicts in host memory out (never `value`; config.value_host_to_host)
        eng_warm = None
        if world == 1 and args.extras and default_legs and not multi:
            with clock("warm_engine_init"):
                eng_warm = Engine(local_rank)
                eng_warm.set_timing(True)
        if world == 1 and default_legs and not multi and not args.no_h2h:
            with clock("host_to_host"):
                hm, badm = host_to_host(eng_cold, eng_warm, we, ws, n, args.h2h_steps)
            mism += badm
            best_cold = max(hm["cold_tables_rebuilt_every_flush"]["verifies_per_s"], hm["in_place_cold"]["verifies_per_s"])
            out["value_host_to_host"] = {"value": best_cold, "unit": "verifies/s", "ratio_to_value": best_cold / value,
                                         "ratio_to_steady_state": (best_cold / steady["value"]) if steady else None,
                                         "what": "SURVEY 8(d)'s wording of the metric: the same 1 M ECDSA-65 + 1 M BIP-340 step with both batches starting in host "
                                                 "memory and the verdicts ending in host memory (streaming queue, tables rebuilt every flush; best of the copying "
                                                 "and the in-place producer form).  `value` is the HBM-resident loop, as the bench contract defines it"}
            out["config"]["value_host_to_host"] = best_cold
            out["config"]["host_to_host_over_value"] = best_cold / value
            out["pcie_inclusive"] = {"mix_streaming": hm}
        # ---- the strong-scaling floor of configs[3] / configs[4], measured on this one GPU
        if world == 1 and default_legs and not multi and not args.no_scaling:
            with clock("strong_scaling_1gpu"):
                try:
                    ss = strong_scaling_sweep(plat, eng_cold, tstream, args.div, in_place_too=args.extras)
                    out["strong_scaling_1gpu"] = ss
                    out["config"]["predicted_speedup_8"] = {"cfg4": ss["cfg4_gossip_replay"]["predicted_speedup_8"], "cfg5": ss["cfg5_commit_storm_streaming"]["predicted_speedup_8"]}
                    mism += ss["cfg4_gossip_replay"]["mismatches"] + ss["cfg5_commit_storm_streaming"]["mismatches"]
                except Exception as e:   # must not take the headline down
                    out["strong_scaling_1gpu"] = {"error": repr(e)}
        # ---- everything else a single GPU can tell (bench_details.json only)
        if world == 1 and args.extras and default_legs and not multi:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_extras
            with clock("extras"):
                mism += bench_extras.run(plat=plat, eng=eng_warm, eng_cold=eng_cold, we=we, ws=ws, n=n, args=args, out=out, timed=timed, clock=clock,
                                         local_rank=local_rank, root=ROOT)
        out["parity"]["mismatches"] = mism
        if args.cpu_sample > 0 and world == 1 and default_legs and not plat.is_stub:   # the CPU baseline is a rank-0, N=1 leg
            with clock("cpu_baseline"):
                cb, cm, rows = cpu_baseline(we, ws, got_e, got_s, args.cpu_sample, n)
            out["cpu_baseline"] = cb
            out["parity"]["oracle_rows_checked"] = rows
            out["parity"]["oracle_mismatches"] = cm
            mism += cm
            out["parity"]["mismatches"] = mism
        else:
            out["cpu_baseline"] = None
        out["phase_seconds"] = dict(clock.t, total=time.perf_counter() - t_start)
        out["host_numa"] = getattr(plat, "numa", None)
        write_outputs(out, json_fd, args.details)
        if eng_warm is not None:
            eng_warm.close()
    eng_cold.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if not args.no_parity and rank == 0 and mism:
        raise SystemExit("PARITY FAILURE: %d mismatching verdicts" % mism)


if __name__ == "__main__":
    main()
