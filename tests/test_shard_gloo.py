"""N>1 path on CPU: world_size-2 gloo run of the stream sharding + final packet gather (opus_amd/shard.py), with the
oracle standing in for the per-rank encoder (test infrastructure only).  The gathered result on rank 0 must equal the
unsharded run stream for stream — sharding must be invisible."""
import os, socket, sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from opus_amd.shard import shard_range, owner_of, PacketGather

STRIDE = 1280


def test_shard_range_partitions():
    for total in (0, 1, 5, 64, 65536, 65537, 524288):
        for world in (1, 2, 3, 4, 8):
            edges = [shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
    for s in range(13):
        r, i = owner_of(s, 13, 4)
        lo, hi = shard_range(13, r, 4)
        assert lo + i == s < hi
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)
    with pytest.raises(ValueError):
        owner_of(13, 13, 4)


def _encode_shard(lo, hi, nframes):
    import signals
    from test_oracle_encoder import OracleEnc
    encs = [OracleEnc(2, bitrate=128000, complexity=10) for _ in range(lo, hi)]
    sigs = [signals.music(nframes, seed=100 + s) for s in range(lo, hi)]
    res = []
    for t in range(nframes):
        lens = torch.zeros(hi - lo, dtype=torch.int32); rng = torch.zeros(hi - lo, dtype=torch.int32); out = torch.zeros((hi - lo, STRIDE), dtype=torch.uint8)
        for k, (e, sg) in enumerate(zip(encs, sigs)):
            data, n, fr = e.encode(np.ascontiguousarray(sg[t * 960:(t + 1) * 960]), 960)
            lens[k] = n; rng[k] = np.uint32(fr).astype(np.int32).item()
            out[k, :n] = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        res.append((lens, rng, out))
    return res


def _worker(rank, world, port, total, nframes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = PacketGather(total, STRIDE, torch.device("cpu"), dst=0)
        assert (g.lo, g.hi) == shard_range(total, rank, world)
        steps = _encode_shard(g.lo, g.hi, nframes)
        got = []
        for lens, rng, out in steps:
            dist.barrier()
            r = g(lens, rng, out)
            assert (r is None) == (rank != 0)
            if r is not None:
                got.append(tuple(x.clone().numpy() for x in r))
        if rank == 0:
            q.put(got)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,world", [(5, 2), (4, 2), (7, 4)])          # ragged (3+2), even, and four ranks ragged (2+2+2+1)
def test_gather_equals_unsharded(total, world):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    nframes = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, nframes, q)) for r in range(world)]
    for p in procs: p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _encode_shard(0, total, nframes)
    for (gl, gr, go), (wl, wr, wo) in zip(got, want):
        assert (gl == wl.numpy()).all() and (gr == wr.numpy()).all() and (go == wo.numpy()).all()
        assert (gl > 2).all()


def _synthetic_steps(lo, hi, nsteps, seed=5):
    """deterministic pseudo-packets (any bytes will do for the transport): step t, global stream s -> length and payload derived from (t, s)"""
    steps = []
    for t in range(nsteps):
        n = hi - lo
        lens = torch.zeros(n, dtype=torch.int32); rng = torch.zeros(n, dtype=torch.int32); out = torch.zeros((n, STRIDE), dtype=torch.uint8)
        for k in range(n):
            s = lo + k
            g = torch.Generator(); g.manual_seed(seed * 1000003 + t * 4099 + s)
            ln = int(torch.randint(3, 700, (1,), generator=g))
            lens[k] = ln; rng[k] = int(torch.randint(-2**31, 2**31 - 1, (1,), generator=g, dtype=torch.int64))
            out[k, :ln] = torch.randint(0, 256, (ln,), generator=g, dtype=torch.uint8)
        steps.append((lens, rng, out))
    return steps


def _pipeline_worker(rank, world, port, total, nsteps, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = PacketGather(total, STRIDE, torch.device("cpu"), dst=0, cap_per_stream=cap, depth=2)
        steps = _synthetic_steps(g.lo, g.hi, nsteps)
        got = []
        # two steps in flight: launch t and t + 1, then flush and read both records (what bench.py does, minus the encoder in between)
        for t in range(0, nsteps, 2):
            slots = []
            for u in (t, t + 1):
                if u < nsteps:
                    g.launch(*steps[u]); slots.append(g.last_slot)
            g.flush()
            if rank == 0:
                for slot in slots:
                    ls, rs, os_ = [], [], []
                    for r, (lo, hi) in enumerate(g.sizes):
                        w = g._recv[slot][r]
                        meta = w[:g.meta_bytes].view(torch.int32).view(2, g.smax)
                        l = meta[0, :hi - lo].contiguous(); ls.append(l.clone()); rs.append(meta[1, :hi - lo].clone())
                        from opus_amd.shard import unpack_packets
                        os_.append(unpack_packets(l, w[g.meta_bytes:], STRIDE) if int(l.sum()) <= g.smax * g.cap else torch.zeros((hi - lo, STRIDE), dtype=torch.uint8))
                    got.append((torch.cat(ls).numpy(), torch.cat(rs).numpy(), torch.cat(os_).numpy()))
        st = g.stats()
        if rank == 0: q.put((got, st))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cap,expect_overflow", [(None, False), (760, False), (200, True)])
def test_pipelined_gather_two_steps_in_flight(cap, expect_overflow):
    """launch / launch / flush with double-buffered fixed-size records; a capacity from a bitrate bound that holds (760 B per stream for lengths < 700) is exact, one
    that does not (200) is reported as overflow instead of passing for a result"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    total, world, nsteps = 11, 3, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, total, nsteps, cap, q)) for r in range(world)]
    for p in procs: p.start()
    got, st = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert st["steps"] == nsteps and st["overflow"] == expect_overflow
    want = _synthetic_steps(0, total, nsteps)
    for (gl, gr, go), (wl, wr, wo) in zip(got, want):
        assert (gl == wl.numpy()).all() and (gr == wr.numpy()).all()
        if not expect_overflow: assert (go == wo.numpy()).all()


def test_wire_capacity_rule():
    """the capacity of the wire record comes from the encoder settings (PacketGather.wire_capacity): VBR twice the nominal packet + 64 B per elementary stream, hard CBR at
    least the CBR packet, unknown / OPUS_BITRATE_MAX the whole slot"""
    W = PacketGather.wire_capacity
    assert W(1280, 128000) == 2 * 320 + 64 and W(1280, 24000) == 2 * 60 + 64
    assert W(1280, 128000, cbr=True) >= 320 and W(1280, 510000, cbr=True) == 1275 + 3 and W(1280, 510000) == 1280
    assert W(1280, -1) == 1280 and W(1280, -1000) == 1280 and W(1280, None) == 1280
    assert W(65536, 255 * 64000, sub_streams=255) == 2 * 40800 + 64 * 255 if 2 * 40800 + 64 * 255 < 65536 else W(65536, 255 * 64000, sub_streams=255) == 65536
    assert W(1280, 64000, frame_rate=100) == 2 * 80 + 64


def _sticky_worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = PacketGather(total, STRIDE, torch.device("cpu"), dst=0, bitrate_bps=16000, depth=2)          # capacity from the bitrate: 2 * 40 + 64 = 144 bytes per stream
        n = g.hi - g.lo
        for t in range(6):
            big = t == 1 and rank == world - 1                                                            # ONE step of ONE rank exceeds it, four steps before the statistics are read
            lens = torch.full((n,), 600 if big else 30, dtype=torch.int32)
            g.launch(lens, torch.zeros(n, dtype=torch.int32), torch.zeros((n, STRIDE), dtype=torch.uint8))
        q.put((rank, g.cap, g.stats()["overflow"]))
    finally:
        dist.destroy_process_group()


def test_overflow_flag_is_sticky_and_seen_by_every_rank():
    """launch()-only use (the bench's timed region): a step over capacity that has long rotated out of the double buffer, on a rank other than dst, is still reported -- by every rank"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue(); world = 3
    procs = [ctx.Process(target=_sticky_worker, args=(r, world, port, 10, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    assert [r[1] for r in res] == [144] * world and all(r[2] for r in res), res
