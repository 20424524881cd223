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


@pytest.mark.parametrize("total", [5, 4])          # ragged (3+2) and even shards
def test_two_rank_gather_equals_unsharded(total):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    nframes = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, nframes, q)) for r in range(2)]
    for p in procs: p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _encode_shard(0, total, nframes)
    for (gl, gr, go), (wl, wr, wo) in zip(got, want):
        assert (gl == wl.numpy()).all() and (gr == wr.numpy()).all() and (go == wo.numpy()).all()
        assert (gl > 2).all()
