"""world_size-2 gloo test of the N>1 path: shard-by-row, verify shards independently, all-gather
the verdict bytes; every rank must end with exactly the single-process verdict vector.  The
per-shard verifier here is the CPU oracle (this file is a test; on GPUs the shard goes through
Engine.verify_*_device and the collective is RCCL)."""
import os
import random
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lightning_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_even_and_grouped():
    b = sharding.shard_bounds(10, 4)
    assert list(b) == [0, 2, 5, 7, 10]
    assert list(sharding.shard_bounds(0, 3)) == [0, 0, 0, 0]
    # channel_announcements (4 rows) followed by channel_updates (1 row): cuts never split a message
    groups = [4] * 5 + [1] * 7
    b = sharding.shard_bounds(27, 8, groups)
    ends = set(np.cumsum(groups).tolist()) | {0}
    assert b[0] == 0 and b[-1] == 27 and all(int(x) in ends for x in b) and all(b[i] <= b[i + 1] for i in range(8))
    # commit_tx storm: 484-row batches stay whole
    b = sharding.shard_bounds(484 * 10, 8, [484] * 10)
    assert all(int(x) % 484 == 0 for x in b)
    with pytest.raises(ValueError):
        sharding.shard_bounds(5, 2, [2, 2])
    # cuts balanced by COST, not by count: a replay that starts with its channel_announcements (weight GOSSIP_WEIGHT_CANN each) must not hand
    # the first rank a shard that is all announcements and as long as the others'
    w = np.array([sharding.GOSSIP_WEIGHT_CANN] * 500 + [sharding.GOSSIP_WEIGHT_OTHER] * 2000)
    b = sharding.shard_bounds(2500, 8, None, w)
    cost = [int(w[b[i]:b[i + 1]].sum()) for i in range(8)]
    assert b[0] == 0 and b[-1] == 2500 and max(cost) - min(cost) <= 2 * sharding.GOSSIP_WEIGHT_CANN, cost
    assert int(b[1]) < 2500 // 8 // 2      # far fewer messages in an announcement shard
    # ... weights and groups together (cuts on group boundaries, balanced by the groups' weights)
    b = sharding.shard_bounds(12, 2, [4, 4, 1, 1, 1, 1], [12, 12, 1, 1, 1, 1])
    assert list(b) == [0, 8, 12] or list(b) == [0, 4, 12]
    msgs = np.frombuffer(bytes([1, 0]) + bytes(10) + bytes([1, 2]) + bytes(5) + bytes([1, 1]) + bytes(3), dtype=np.uint8)
    assert list(sharding.gossip_weights(msgs, [0, 12, 19, 24])) == [sharding.GOSSIP_WEIGHT_CANN, 1, 1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rnd = random.Random(4242)  # same batch on every rank
    n, N = 101, 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    hs, sg, pk = [], [], []
    for i in range(n):
        d = rnd.randrange(1, N).to_bytes(32, "big")
        h = rnd.randbytes(32)
        s = orc.ecdsa_sign(h, d, rnd.randrange(1, N).to_bytes(32, "big"))
        p = orc.pubkey_create(d)
        if i % 5 == 0:
            h = bytes([h[0] ^ 2]) + h[1:]
        hs.append(h); sg.append(s); pk.append(bytes([2 + (p[64] & 1)]) + p[1:33])
    A = lambda rows, w: np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), w)
    hs, sg, pk = A(hs, 32), A(sg, 64), A(pk, 33)
    groups = [4] * 20 + [1] * 21
    seen = []

    def verify_range(lo, hi):          # the role Engine.verify_*_device / the streaming queue play in bench.py's sharded_configs()
        seen.append((lo, hi))
        return torch.from_numpy(orc.ecdsa_verify_batch(np.ascontiguousarray(hs[lo:hi]), np.ascontiguousarray(sg[lo:hi]),
                                                       np.ascontiguousarray(pk[lo:hi]), 33, 1))
    full, b = sharding.run_sharded(n, rank, world, verify_range, groups)    # exactly what bench.py --gpus N calls for configs[4]
    lo, hi = seen[0]
    assert (lo, hi) == (int(b[rank]), int(b[rank + 1]))
    ref = orc.ecdsa_verify_batch(hs, sg, pk, 33, 1)
    ok = bool(np.array_equal(full.numpy(), ref))
    # configs[3] shape: raw gossip messages, int8 verdicts (first bad signature), sharded by message
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gossip_stream as gs
    net = gs.Net(orc, 777, n_nodes=8, n_chans=9)
    msgs, ids = [], []
    for c in range(9):
        m = net.cann(c)
        msgs.append(gs.damage(net.rnd, m, "sig") if c % 4 == 1 else m); ids.append(bytes(33))
    for k in range(23):
        c, d = k % 9, k & 1
        m = net.cupd(c, d, gs.NOW - k)
        msgs.append(gs.damage(net.rnd, m, "tail") if k % 6 == 2 else m); ids.append(net.node_id[net.chans[c]["n"][d]])
    off = np.concatenate([[0], np.cumsum([len(m) for m in msgs])]).astype(np.uint64)
    blob = np.frombuffer(b"".join(msgs) + b"\x00", dtype=np.uint8)
    idarr = np.frombuffer(b"".join(ids), dtype=np.uint8).reshape(len(ids), 33)

    def gossip_range(lo, hi):
        sub = off[lo:hi + 1] - off[lo]
        return torch.from_numpy(orc.sigcheck_gossip_batch(np.ascontiguousarray(blob[int(off[lo]):int(off[hi]) + 1]), sub.astype(np.uint64),
                                                          np.ascontiguousarray(idarr[lo:hi]), 1))
    gfull, gb = sharding.run_sharded(len(msgs), rank, world, gossip_range)
    gref = orc.sigcheck_gossip_batch(blob, off, idarr, 1)
    ok = ok and gfull.dtype == torch.int8 and bool(np.array_equal(gfull.numpy(), gref)) and int((gref > 0).sum()) >= 5 and int((gref == 0).sum()) >= 15
    # a job smaller than the world (three commitments of 4 rows on up to 8 ranks): most shards are EMPTY, the gather is still exact
    def tiny_range(a, z):
        if a == z:
            return torch.zeros(0, dtype=torch.uint8)
        return torch.from_numpy(orc.ecdsa_verify_batch(np.ascontiguousarray(hs[a:z]), np.ascontiguousarray(sg[a:z]), np.ascontiguousarray(pk[a:z]), 33, 1))
    tfull, tb = sharding.run_sharded(12, rank, world, tiny_range, [4, 4, 4])
    ok = ok and bool(np.array_equal(tfull.numpy(), ref[:12])) and all(int(x) % 4 == 0 for x in tb) and len(tb) == world + 1
    q.put((rank, ok, int(ref.sum()), lo, hi))
    dist.destroy_process_group()


def test_gossip_weights_of_short_and_empty_messages():
    """a batch whose last message is shorter than its type field (or empty) has weight 1, as lamd_multi_sigcheck_gossip_batch has it --
    it used to raise IndexError"""
    cann, cupd = bytes([1, 0]) + bytes(10), bytes([1, 2]) + bytes(4)
    msgs = [cann, cupd, b"\x01", b"", cann, b""]
    blob = np.frombuffer(b"".join(msgs), dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum([len(m) for m in msgs])]).astype(np.uint64)
    w = sharding.gossip_weights(blob, off)
    assert w.tolist() == [sharding.GOSSIP_WEIGHT_CANN, 1, 1, 1, sharding.GOSSIP_WEIGHT_CANN, 1]
    assert sharding.gossip_weights(np.zeros(0, np.uint8), np.zeros(3, np.uint64)).tolist() == [1, 1]
    b = sharding.shard_bounds(len(msgs), 2, None, w)
    assert b[0] == 0 and b[-1] == len(msgs)


def test_two_rank_shard_and_all_gather():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert res[0][2] == 101 - 21  # every 5th row corrupted
    ranges = sorted((r[3], r[4]) for r in res)
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == 101


class _StandInEngine:
    """plays lightning_amd.Engine for sharding.LateGather: records the protocol, checks that a join only follows its mark"""

    def __init__(self):
        self.trace, self.marked = [], set()

    def results_mark(self, slot):
        assert 0 <= slot < 16
        self.marked.add(slot)
        self.trace.append(("mark", slot))

    def stream_wait_mark(self, slot, stream):
        assert slot in self.marked and stream == 1234
        self.trace.append(("join", slot))

    def wait_event(self, ev):
        self.trace.append(("wait", ev))


def _worker_late(rank, world, port, q, nbuf=2):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, steps = 97, 5 if nbuf == 2 else 11
    kinds = ("e", "s")
    bufs = {k: [torch.zeros(n, dtype=torch.uint8) for _ in range(nbuf)] for k in kinds}
    outs = {k: torch.zeros(world * n, dtype=torch.uint8) for k in kinds}
    pattern = lambda r, step, k: ((torch.arange(n) * (3 + kinds.index(k)) + 7 * r + 5 * step) % 3 == 0).to(torch.uint8)
    eng, shots, events = _StandInEngine(), [], []

    class Ev:
        def __init__(self, ident):
            self.cuda_event = ident

    def new_event():
        events.append(len(events) + 100)
        return Ev(events[-1])

    def all_gather(out, src):
        dist.all_gather_into_tensor(out, src)
        shots.append(out.clone())
    g = sharding.LateGather(eng, kinds, bufs, outs, 1234, all_gather, new_event)
    for step in range(steps):                        # the loop of bench.py's step(), with a synchronous stand-in for the device calls
        b = step % nbuf
        for k in kinds:
            g.before_call(k, b)
            bufs[k][b].copy_(pattern(rank, step, k))
            g.after_call(k, b)
        g.end_step(b)
        assert len(shots) == 2 * step                # the gathers of step k are issued by step k+1
    g.flush()
    ok = len(shots) == 2 * steps and g.log == [(k, s % nbuf) for s in range(steps) for k in kinds]
    for s in range(steps):                           # gather j carried step j's verdicts of BOTH ranks, whatever was written since
        for i, k in enumerate(kinds):
            want = torch.cat([pattern(r, s, k) for r in range(world)])
            ok = ok and bool(torch.equal(shots[2 * s + i], want))
    # protocol: from step nbuf on every call first waits for the event of the gather that read its buffer nbuf steps earlier (bench.py: four buffers)
    waits = [t[1] for t in eng.trace if t[0] == "wait"]
    ok = ok and waits == [100 + 2 * (s - nbuf) + i for s in range(nbuf, steps) for i in range(2)]
    # ... and every join follows the mark of the same slot
    joins = [t[1] for t in eng.trace if t[0] == "join"]
    ok = ok and joins == [2 * (s % nbuf) + i for s in range(steps) for i in range(2)]
    q.put((rank, ok))
    dist.destroy_process_group()


def test_two_rank_late_gather_protocol_and_payload():
    """sharding.LateGather (the collective step of bench.py's weak-scaling headline) under gloo, world 2: per-call marks and event
    waits in the right order, every step's verdicts of both ranks gathered exactly once, the last step's by flush()"""
    world = 2
    ctx = mp.get_context("spawn")
    for nbuf in (2, 4):   # two verdict buffers per kind (rounds 2-4) and four (bench.py since round 5)
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_worker_late, args=(r, world, port, q, nbuf)) for r in range(world)]
        for p in ps:
            p.start()
        res = [q.get(timeout=120) for _ in range(world)]
        for p in ps:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert sorted(r[0] for r in res) == [0, 1] and all(r[1] for r in res), (nbuf, res)


def test_eight_rank_ragged_and_empty_shards():
    """the same worker at world 8 (the size the driver scales to): 101 rows in groups of 4 and 1 -> ragged shards cut on group
    boundaries, 32 gossip messages over 8 ranks, and a 12-row job that leaves most ranks with an empty shard; every rank must end
    with the single-process verdict vector"""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res
    ranges = sorted((r[3], r[4]) for r in res)
    assert ranges[0][0] == 0 and ranges[-1][1] == 101 and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    assert len({hi - lo for lo, hi in ranges}) > 1          # ragged


def test_segment_bounds_cut_every_segment_for_every_rank():
    sb = sharding.segment_bounds([0, 10, 30], 4)
    assert sb.tolist() == [[0, 2, 5, 7, 10], [10, 15, 20, 25, 30]]
    # weights balance INSIDE a segment; an empty segment and a segment shorter than the world are fine
    w = np.ones(30, dtype=np.int64)
    w[:5] = 10
    sb = sharding.segment_bounds([0, 10, 10, 12, 30], 4, w)
    assert sb.shape == (4, 5) and sb[1].tolist() == [10] * 5 and sb[2, 0] == 10 and sb[2, -1] == 12 and sb[3, 0] == 12 and sb[3, -1] == 30
    assert sb[0, 1] < 3                      # the heavy head of segment 0 is cut early
    assert all(sb[s, g] <= sb[s, g + 1] for s in range(4) for g in range(4))
    with pytest.raises(ValueError):
        sharding.segment_bounds([1, 5], 2)
    with pytest.raises(ValueError):
        sharding.segment_bounds([0, 5, 3], 2)
    # one rank: the job comes back in job order, several dtypes
    for dt in (torch.uint8, torch.int8):
        full, b = sharding.run_sharded_segments(30, [0, 10, 30], 0, 1, lambda lo, hi: (torch.arange(lo, hi) % 5).to(dt))
        assert full.dtype == dt and full.tolist() == [i % 5 for i in range(30)]
    with pytest.raises(ValueError):
        sharding.run_sharded_segments(31, [0, 10, 30], 0, 1, lambda lo, hi: torch.zeros(hi - lo, dtype=torch.uint8))


def _worker_segments(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok, calls = True, []
    # a "gossip replay": 37 announcements (int8 verdict 0..4) then 203 updates (0..1); weights 10 / 1; a job that leaves ranks without a position
    expect = np.concatenate([(np.arange(37) * 7) % 5, (np.arange(203) * 3) % 2]).astype(np.int8)
    w = np.concatenate([np.full(37, 10), np.ones(203)]).astype(np.int64)

    def verify(lo, hi):
        calls.append((lo, hi))
        return torch.from_numpy(expect[lo:hi].copy())
    marks = []
    full, sb = sharding.run_sharded_segments(240, [0, 37, 240], rank, world, verify, w, before_gather=lambda: marks.append(len(calls)),
                                             empty=torch.empty(0, dtype=torch.int8))
    ok &= full.dtype == torch.int8 and np.array_equal(full.numpy(), expect)
    ok &= len(calls) <= 2 and marks == [len(calls)] and all(hi > lo for lo, hi in calls)
    ok &= all((lo < 37) == (hi <= 37) for lo, hi in calls)          # a range never straddles the two kinds
    # ... and as ONE call over all of the rank's ranges (the engine's spans form)
    seen = []

    def verify_all(ranges):
        seen.append(list(ranges))
        return torch.from_numpy(np.concatenate([expect[lo:hi] for lo, hi in ranges]))
    full1, _ = sharding.run_sharded_segments(240, [0, 37, 240], rank, world, verify, w, empty=torch.empty(0, dtype=torch.int8), verify_ranges=verify_all)
    ok &= np.array_equal(full1.numpy(), expect) and len(seen) == 1 and seen[0] == calls[:2] and len(calls) == 2
    tiny = np.array([1, 0, 1], dtype=np.uint8)
    full2, _ = sharding.run_sharded_segments(3, [0, 1, 3], rank, world, lambda lo, hi: torch.from_numpy(tiny[lo:hi].copy()))
    ok &= np.array_equal(full2.numpy(), tiny)
    q.put((rank, bool(ok), calls))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_segmented_job_over_ranks_every_rank_gets_the_whole_vector(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_segments, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world)) and all(r[1] for r in res), res
    # every rank holds a piece of BOTH kinds (that is the point of cutting segment by segment)
    assert all(len(r[2]) == 2 for r in res), res
    covered = sorted(rng for r in res for rng in r[2])
    assert covered[0][0] == 0 and covered[-1][1] == 240 and sum(hi - lo for lo, hi in covered) == 240
