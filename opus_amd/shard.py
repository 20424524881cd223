"""Stream sharding across the GPUs of one node.

Streams never interact (reference include/opus.h:425-429: separate state blobs), so stream s lives on exactly one
rank for its lifetime and a frame-step needs no data-path collective.  The only exchange is the final gather of
(length, final range, payload) to the rank that owns the output — RCCL over xGMI when the tensors are on GPUs
(torch.distributed backend "nccl"), gloo in the CPU tests.  Nothing here touches the codec itself.
"""
import torch
import torch.distributed as dist


def shard_range(total_streams, rank, world):
    """Contiguous block partition: rank r owns streams [lo, hi).  Blocks differ by at most one stream."""
    if total_streams < 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError("bad shard request")
    q, r = divmod(total_streams, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def owner_of(stream, total_streams, world):
    """Inverse of shard_range: (rank, local index) of a global stream id."""
    if not (0 <= stream < total_streams):
        raise ValueError("stream out of range")
    q, r = divmod(total_streams, world)
    edge = r * (q + 1)
    if stream < edge:
        return stream // (q + 1), stream % (q + 1)
    return r + (stream - edge) // q, (stream - edge) % q


class PacketGather:
    """Final gather of one frame-step's packets to `dst`.

    Every rank passes its shard's lens [s_r] int32, final_range [s_r] int32 and out [s_r, stride] uint8; ragged shards
    (total % world != 0) are padded to the largest shard for the collective and trimmed on `dst`.  Buffers are
    allocated once and reused every step."""

    def __init__(self, total_streams, stride, device, dst=0, group=None):
        self.total, self.stride, self.dst, self.group = total_streams, stride, dst, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lo, self.hi = shard_range(total_streams, self.rank, self.world)
        self.smax = max(shard_range(total_streams, r, self.world)[1] - shard_range(total_streams, r, self.world)[0] for r in range(self.world))
        self._pad = None
        if self.world > 1 and self.hi - self.lo != self.smax:
            self._pad = (torch.zeros(self.smax, dtype=torch.int32, device=device), torch.zeros(self.smax, dtype=torch.int32, device=device),
                         torch.zeros((self.smax, stride), dtype=torch.uint8, device=device))
        self._recv = None
        if self.world > 1 and self.rank == dst:
            self._recv = ([torch.empty(self.smax, dtype=torch.int32, device=device) for _ in range(self.world)],
                          [torch.empty(self.smax, dtype=torch.int32, device=device) for _ in range(self.world)],
                          [torch.empty((self.smax, stride), dtype=torch.uint8, device=device) for _ in range(self.world)])

    def __call__(self, lens, final_range, out):
        """Returns (lens, final_range, out) for ALL streams on dst (views of the receive buffers), None elsewhere."""
        n = self.hi - self.lo
        if lens.shape[0] != n or out.shape != (n, self.stride):
            raise ValueError("shard shape mismatch")
        if self.world == 1:
            return lens, final_range, out
        if self._pad is not None:
            self._pad[0][:n].copy_(lens); self._pad[1][:n].copy_(final_range); self._pad[2][:n].copy_(out)
            lens, final_range, out = self._pad
        r = self._recv
        dist.gather(lens, r[0] if r else None, dst=self.dst, group=self.group)
        dist.gather(final_range, r[1] if r else None, dst=self.dst, group=self.group)
        dist.gather(out, r[2] if r else None, dst=self.dst, group=self.group)
        if r is None:
            return None
        sizes = [shard_range(self.total, k, self.world) for k in range(self.world)]
        return (torch.cat([r[0][k][:hi - lo] for k, (lo, hi) in enumerate(sizes)]), torch.cat([r[1][k][:hi - lo] for k, (lo, hi) in enumerate(sizes)]),
                torch.cat([r[2][k][:hi - lo] for k, (lo, hi) in enumerate(sizes)]))

    def launch(self, lens, final_range, out):
        """Collective only (no concatenation on dst): what bench.py puts inside the timed region."""
        n = self.hi - self.lo
        if self.world == 1:
            return
        if self._pad is not None:
            self._pad[0][:n].copy_(lens); self._pad[1][:n].copy_(final_range); self._pad[2][:n].copy_(out)
            lens, final_range, out = self._pad
        r = self._recv
        dist.gather(lens, r[0] if r else None, dst=self.dst, group=self.group)
        dist.gather(final_range, r[1] if r else None, dst=self.dst, group=self.group)
        dist.gather(out, r[2] if r else None, dst=self.dst, group=self.group)
