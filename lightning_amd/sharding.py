"""Multi-GPU partitioning of a verification batch.

Every (hash, key, signature) row is independent, so the path shards trivially: rank g of G takes
a contiguous range of rows and no data-path collective is needed for correctness.  Ranges are cut
on GROUP boundaries when the caller supplies group sizes, so that the four signatures of one
channel_announcement, or the 484 signatures of one commitment_signed, stay on one GPU (their
shared key is then parsed / tabulated once).  The only collective is the all-gather of the
verdict bytes (RCCL over xGMI on GPUs; gloo in the CPU tests) so that every rank ends with the
whole verdict vector, as the north star asks.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_rows, world, group_sizes=None, weights=None):
    """-> int64 array b of world+1 row offsets; rank g owns rows [b[g], b[g+1]).
    group_sizes: optional sequence summing to n_rows; cuts fall only between groups.
    weights: optional cost per group (per row when there are no groups); the cuts then balance the WEIGHT of the shards instead of their
    row count -- a gossip replay is cut on message boundaries but balanced by what a message costs (a channel_announcement is four
    signatures, two of them under bitcoin keys that never recur: GOSSIP_WEIGHT_*), or the strong-scaling time is the first shard's."""
    if group_sizes is None and weights is None:
        return np.array([(n_rows * g) // world for g in range(world + 1)], dtype=np.int64)
    if group_sizes is None:
        group_sizes = np.ones(n_rows, dtype=np.int64)
    ends = np.cumsum(np.asarray(group_sizes, dtype=np.int64))
    if len(ends) == 0 or ends[-1] != n_rows:
        raise ValueError("group sizes do not add up to n_rows")
    if weights is None:
        wends, total = ends, n_rows
    else:
        wends = np.cumsum(np.asarray(weights, dtype=np.int64))
        if len(wends) != len(ends):
            raise ValueError("one weight per group")
        total = int(wends[-1])
    b = [0]
    for g in range(1, world):
        target = (total * g) // world
        j = int(np.searchsorted(wends, target, side="left"))  # first group whose cumulative weight reaches the target
        cut = int(ends[j]) if j < len(ends) else n_rows
        b.append(max(cut, b[-1]))
    b.append(n_rows)
    return np.array(b, dtype=np.int64)


# cost of one gossip message in units of one channel_update under a node id the key-table cache knows, measured on one MI355X (round 5,
# tools/call_trace_probe.py: 1/8 shards of BASELINE configs[3], T = 0.64 ms + 28.8 ns per channel_announcement / 2.43 ns per channel_update):
# an announcement is four signatures, four key parses and a 430-byte SHA256d, and two of its signatures are under bitcoin keys that never recur,
# i.e. on the per-signature ladder -- ten to twelve updates' worth.  With weight 10 the 1/8 shards of configs[3] come out level (2.85 ms for 87.5 k
# announcements, 2.9 ms for 875 k updates; with 12: 2.8 against 3.2 ms) and a shard of announcements stays below one full round of the ladder
# kernel (196 608 lanes): 2 x 87 k cold rows take 1.5 ms, 2 x 104 k (one row over a round for 6 % of the lanes) 2.5 ms.
GOSSIP_WEIGHT_CANN, GOSSIP_WEIGHT_OTHER = 10, 1


def gossip_weights(msgs, off):
    """per-message shard weights of a packed gossip batch (type 256 = channel_announcement)"""
    off = np.asarray(off, dtype=np.int64)
    m = np.asarray(msgs)
    has_type = (off[1:] - off[:-1]) >= 2          # a message shorter than its type field weighs 1 (as lamd_multi_sigcheck_gossip_batch has it)
    at = np.where(has_type, off[:-1], 0)
    is_cann = has_type & (m[at] == 1) & (m[np.minimum(at + 1, len(m) - 1)] == 0) if len(m) else np.zeros(len(off) - 1, dtype=bool)
    return np.where(is_cann, GOSSIP_WEIGHT_CANN, GOSSIP_WEIGHT_OTHER).astype(np.int64)


def all_gather_verdicts(ok_local, bounds, rank, world):
    """ok_local: uint8 tensor holding this rank's verdicts (len = bounds[rank+1]-bounds[rank]).
    Returns the full verdict vector (len bounds[-1]) on every rank.  Shards may be ragged, so the
    payload is padded to the longest shard (a <= 1 MB, latency-bound collective either way)."""
    if world == 1:
        return ok_local
    dt = ok_local.dtype
    if dt != torch.uint8:            # int8 gossip verdicts travel as bytes
        ok_local = ok_local.view(torch.uint8)
    sizes = [int(bounds[g + 1] - bounds[g]) for g in range(world)]
    m = max(sizes)
    pad = torch.zeros(m, dtype=torch.uint8, device=ok_local.device)
    pad[:sizes[rank]] = ok_local
    out = torch.empty(world * m, dtype=torch.uint8, device=ok_local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[g * m:g * m + sizes[g]] for g in range(world)]).view(dt)


def run_sharded(n_verdicts, rank, world, verify_range, group_sizes=None, weights=None):
    """The north-star split of ONE global job over `world` ranks: rank g verifies verdict positions [b[g], b[g+1]) -- cut on group
    boundaries (a channel_announcement's four signatures, a commitment's 484) -- through `verify_range(lo, hi)` (-> 1-D uint8 / int8
    tensor of hi-lo verdicts on the rank's device: the engine on GPUs, a CPU checker in the gloo test), then every rank receives the
    whole verdict vector (ragged all-gather).  Returns (full_verdicts, bounds).  bench.py --gpus N runs BASELINE configs[3] and [4]
    through this function; tests/test_sharding_gloo.py runs the same function under gloo."""
    b = shard_bounds(n_verdicts, world, group_sizes, weights)
    lo, hi = int(b[rank]), int(b[rank + 1])
    local = verify_range(lo, hi)
    if local.numel() != hi - lo:
        raise ValueError("verify_range returned %d verdicts for %d positions" % (local.numel(), hi - lo))
    return all_gather_verdicts(local, b, rank, world), b


def segment_bounds(seg_edges, world, weights=None):
    """A job whose MIX changes along its length -- a gossip replay is 500 k channel_announcements (four signatures each, two of them under keys that
    never recur: the per-signature ladder) followed by 2 M channel_updates (one signature under a node key with a table) -- is cut segment by
    segment: `seg_edges` = [0, e1, .., n] (non-decreasing) names the segments, EVERY segment is cut into `world` contiguous ranges (balanced by
    `weights` inside the segment, cuts on unit boundaries), and rank r takes range r of every segment.  Every rank then holds the same mix -- with one
    cut over the whole job five ranks of eight get nothing but announcements, and the job's time is the slowest kind's (profiles/r06_shard_timeline_cold.txt).
    -> int64 array [segments, world + 1] of absolute offsets."""
    seg_edges = [int(e) for e in seg_edges]
    if len(seg_edges) < 2 or seg_edges[0] != 0 or any(b < a for a, b in zip(seg_edges, seg_edges[1:])):
        raise ValueError("segment edges must start at 0 and not decrease")
    out = np.zeros((len(seg_edges) - 1, world + 1), dtype=np.int64)
    for s, (a, z) in enumerate(zip(seg_edges, seg_edges[1:])):
        w = None if weights is None else np.asarray(weights)[a:z]
        out[s] = a + shard_bounds(z - a, world, None, w if (w is not None and z > a) else None) if z > a else a
    return out


def run_sharded_segments(n_verdicts, seg_edges, rank, world, verify_range, weights=None, before_gather=None, empty=None, verify_ranges=None):
    """run_sharded() for a job of several segments (segment_bounds): rank r verifies range r of every segment -- `verify_range(lo, hi)` once per
    non-empty range; on GPUs the calls are asynchronous and overlap on the engine's lanes -- then ONE ragged all-gather carries every rank's verdicts
    of all its ranges, and every rank ends with the whole vector in job order.  `before_gather()` (optional) is called between the last
    verify_range and the collective (the engine's device-side edge to the consumer stream); `empty` = a zero-length tensor of the verdicts' dtype on
    the rank's device, for a rank that gets no position at all (default: uint8 on the CPU).  `verify_ranges(list of (lo, hi))`, when given, replaces
    the per-range calls: ONE call for all of the rank's non-empty ranges that returns their verdicts concatenated in that order (the engine's
    spans form, lamd_sigcheck_gossip_spans_device: one front end per rank instead of one per range).
    Returns (full_verdicts, bounds[segments, world + 1])."""
    sb = segment_bounds(seg_edges, world, weights)
    if int(sb[-1, -1]) != n_verdicts:
        raise ValueError("segments do not cover the job")
    parts = []
    mine = [(int(sb[s, rank]), int(sb[s, rank + 1])) for s in range(sb.shape[0]) if sb[s, rank + 1] > sb[s, rank]]
    if verify_ranges is not None and mine:
        v = verify_ranges(mine)
        if v.numel() != sum(hi - lo for lo, hi in mine):
            raise ValueError("verify_ranges returned %d verdicts for %d positions" % (v.numel(), sum(hi - lo for lo, hi in mine)))
        parts.append(v)
    else:
        for lo, hi in mine:
            v = verify_range(lo, hi)
            if v.numel() != hi - lo:
                raise ValueError("verify_range returned %d verdicts for %d positions" % (v.numel(), hi - lo))
            parts.append(v)
    if before_gather is not None:
        before_gather()
    sizes = [int(sum(sb[s, g + 1] - sb[s, g] for s in range(sb.shape[0]))) for g in range(world)]
    if not parts:      # a rank without a single position still takes part in the collective
        parts = [empty if empty is not None else torch.empty(0, dtype=torch.uint8)]
    local = torch.cat(parts) if len(parts) > 1 else parts[0]
    dt = local.dtype
    if world == 1:
        gathered = [local.view(torch.uint8) if dt != torch.uint8 else local]
    else:
        m = max(sizes)
        pad = torch.zeros(m, dtype=torch.uint8, device=local.device)
        pad[:sizes[rank]] = local.view(torch.uint8) if dt != torch.uint8 else local
        flat = torch.empty(world * m, dtype=torch.uint8, device=local.device)
        dist.all_gather_into_tensor(flat, pad)
        gathered = [flat[g * m:g * m + sizes[g]] for g in range(world)]
    full = torch.empty(n_verdicts, dtype=torch.uint8, device=local.device)
    for g in range(world):
        o = 0
        for s in range(sb.shape[0]):
            a, z = int(sb[s, g]), int(sb[s, g + 1])
            full[a:z] = gathered[g][o:o + z - a]
            o += z - a
    return full.view(dt), sb


class LateGather:
    """The collective step of the weak-scaling headline (bench.py --gpus N): every rank verifies its own batches, and the verdict
    bytes of every batch are all-gathered -- with every dependency per call and by device-side events, no host synchronisation:

    * a verdict buffer (`nbuf` per batch kind, taken in turn step by step: SIX in bench.py since round 5 -- with two, step k + 2 could not start before
      the gather of step k had run, and that gather is a small copy kernel that waits 0.3-1 ms for wave slots on the saturated chip: the calls bunched up in
      pairs with holes of 3-4 ms without a table-driven launch between them, 0.91 of the plain loop) is written again only after the gather that read it:
      `before_call(kind, b)` makes the engine wait for that gather's event (lamd_wait_event);
    * a gather waits for "everything submitted up to its call": `after_call(kind, b)` marks it (lamd_results_mark), the consumer
      stream joins the mark when the gather is issued (lamd_stream_wait_mark);
    * the gathers of step k are issued by `end_step(b)` of step k+1, when step k is (nearly) done, so that the consumer stream
      never carries a wait that lasts a whole step; `flush()` issues the last step's.

    One join per step instead (the next-but-one step waiting for it) couples the calls of a step: the ECDSA lane cannot start its
    next front end before the BIP-340 call of the same step has finished -- measured -12 % on one rank.

    eng: results_mark(slot) / stream_wait_mark(slot, stream_ptr) / wait_event(event_ptr) (lightning_amd.Engine; a stand-in in the
    gloo test).  bufs[kind] = [tensor, tensor] local verdict buffers, outs[kind] = tensor of world * n gathered verdicts.
    all_gather(out, src) and new_event() (-> object with .cuda_event, recorded on the consumer stream) come from the caller:
    torch.distributed / torch.cuda on GPUs."""

    def __init__(self, eng, kinds, bufs, outs, stream_ptr, all_gather, new_event):
        self.nbuf = len(next(iter(bufs.values())))
        if self.nbuf * len(kinds) > 16:
            raise ValueError("lamd_results_mark has sixteen slots: buffers x batch kinds must not exceed them")
        self.eng, self.kinds, self.bufs, self.outs = eng, list(kinds), bufs, outs
        self.stream_ptr, self.all_gather, self.new_event = stream_ptr, all_gather, new_event
        self.consumed = {k: [None] * self.nbuf for k in self.kinds}
        self.pending = None
        self.log = []          # (kind, buffer index) in the order the gathers were issued

    def reset(self):
        self.consumed = {k: [None] * self.nbuf for k in self.kinds}
        self.pending = None

    def slot(self, kind, b):
        return len(self.kinds) * b + self.kinds.index(kind)

    def before_call(self, kind, b):
        ev = self.consumed[kind][b]
        if ev is not None:
            self.eng.wait_event(ev.cuda_event)

    def after_call(self, kind, b):
        # the gather of this call's verdicts needs this call only: one event on its lane (lamd_results_mark_last) where the engine has it
        # (LAMD_BENCH_MARK_ALL=1: the round-4 form, an event on every lane)
        import os
        mark = getattr(self.eng, "results_mark_last", None)
        if mark is None or os.environ.get("LAMD_BENCH_MARK_ALL", "0") == "1":
            mark = self.eng.results_mark
        mark(self.slot(kind, b))

    def end_step(self, b):
        if self.pending is not None:
            self._gather(self.pending)
        self.pending = b

    def flush(self):
        if self.pending is not None:
            self._gather(self.pending)
            self.pending = None

    def _gather(self, b):
        for kind in self.kinds:
            self.eng.stream_wait_mark(self.slot(kind, b), self.stream_ptr)
            self.all_gather(self.outs[kind], self.bufs[kind][b])
            self.consumed[kind][b] = self.new_event()
            self.log.append((kind, b))
