"""Stream sharding across the GPUs of one node.

Streams never interact (reference include/opus.h:425-429: separate state blobs), so stream s lives on exactly one
rank for its lifetime and a frame-step needs no data-path collective.  The only exchange is the final gather of
(length, final range, payload) to the rank that owns the output — RCCL over xGMI when the tensors are on GPUs
(torch.distributed backend "nccl"), gloo in the CPU tests.  Nothing here touches the codec itself.

The payload travels compacted: every rank packs its packets back to back on the device (exclusive prefix sum of the lengths,
one scatter — no dynamic shapes, so no host round trip for the packing itself), the fixed-size (length | final range) table
is gathered with one collective, and the packed bytes follow with one point-to-point transfer per rank of exactly the bytes
the packets hold (~320 B per 128 kb/s frame instead of the 1,280 B slot: 21 MB instead of 84 MB per rank per step at 65,536
streams).  The byte count has to be known on the host to size that transfer; reading it is the one synchronisation point of
a step, and it waits for nothing that the transfer would not have had to wait for (the encode of that step).
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


def pack_packets(lens, out, packed):
    """Packs out[s, :lens[s]] back to back into `packed` (capacity >= n * stride + 1 bytes, last byte = spill slot).  Device-side,
    static shapes: bytes beyond a packet's length are scattered to the spill slot.  Returns the exclusive prefix sum of the lengths."""
    n, stride = out.shape
    l = lens.clamp(min=0).to(torch.int64)
    offs = torch.cumsum(l, 0) - l
    if out.is_cuda:                                                     # one wave per packet, coalesced byte copies (opusgpu_pack_packets_dev, opus_amd.hip)
        from . import lib
        r = lib().opusgpu_pack_packets_dev(out.data_ptr(), stride, lens.data_ptr(), offs.data_ptr(), packed.data_ptr(), n, torch.cuda.current_stream(out.device).cuda_stream)
        if r != 0: raise RuntimeError("opusgpu_pack_packets_dev failed: %d" % r)
        return offs
    col = torch.arange(stride, device=out.device, dtype=torch.int64)
    idx = torch.where(col[None, :] < l[:, None], offs[:, None] + col[None, :], torch.full((), packed.numel() - 1, device=out.device, dtype=torch.int64))
    packed.scatter_(0, idx.reshape(-1), out.reshape(-1))
    return offs


def unpack_packets(lens, packed, stride):
    """Inverse of pack_packets on the receiving side: [n, stride] uint8, zero beyond each packet."""
    l = lens.clamp(min=0).to(torch.int64)
    offs = torch.cumsum(l, 0) - l
    col = torch.arange(stride, device=packed.device, dtype=torch.int64)
    valid = col[None, :] < l[:, None]
    idx = torch.where(valid, offs[:, None] + col[None, :], torch.zeros((), device=packed.device, dtype=torch.int64))
    return torch.where(valid, packed[idx.reshape(-1)].reshape(len(l), stride), torch.zeros((), dtype=torch.uint8, device=packed.device))


class PacketGather:
    """Final gather of one frame-step's packets to `dst`.

    Every rank passes its shard's lens [s_r] int32, final_range [s_r] int32 and out [s_r, stride] uint8; ragged shards
    (total % world != 0) are padded to the largest shard for the table collective and trimmed on `dst`.  Buffers are
    allocated once and reused every step."""

    def __init__(self, total_streams, stride, device, dst=0, group=None):
        self.total, self.stride, self.dst, self.group, self.device = total_streams, stride, dst, group, device
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lo, self.hi = shard_range(total_streams, self.rank, self.world)
        self.sizes = [shard_range(total_streams, r, self.world) for r in range(self.world)]
        self.smax = max(hi - lo for lo, hi in self.sizes)
        n = self.hi - self.lo
        # collectives on device tensors need a backend that moves them (nccl = RCCL); with gloo (tests) everything is staged through host memory
        self.stage_cpu = dist.is_initialized() and dist.get_backend(group) == "gloo" and torch.device(device).type != "cpu"
        cdev = torch.device("cpu") if self.stage_cpu else device
        self.cdev = cdev
        self._meta = torch.zeros((self.smax, 2), dtype=torch.int32, device=cdev)
        self._packed = torch.zeros(n * stride + 1, dtype=torch.uint8, device=device)
        self._recv_meta = [torch.empty((self.smax, 2), dtype=torch.int32, device=cdev) for _ in range(self.world)] if (self.world > 1 and self.rank == dst) else None
        self._recv_bytes = [torch.empty((hi - lo) * stride + 1, dtype=torch.uint8, device=cdev) for lo, hi in self.sizes] if (self.world > 1 and self.rank == dst) else None
        self.last_bytes = 0

    def launch(self, lens, final_range, out):
        """The exchange of one step (no re-assembly on dst): what bench.py puts inside the timed region.  Returns the per-rank byte counts on dst."""
        n = self.hi - self.lo
        if lens.shape[0] != n or out.shape != (n, self.stride):
            raise ValueError("shard shape mismatch")
        if self.world == 1:
            return None
        pack_packets(lens, out, self._packed)
        meta = torch.stack([lens, final_range], 1)
        self._meta[:n].copy_(meta)
        dist.gather(self._meta, self._recv_meta, dst=self.dst, group=self.group)
        nbytes = int(lens.clamp(min=0).sum().item())                       # the one host read of the step
        self.last_bytes = nbytes
        if self.rank != self.dst:
            if nbytes:
                buf = self._packed[:nbytes]
                dist.send(buf.cpu() if self.stage_cpu else buf, self.dst, group=self.group)
            return None
        counts = [int(self._recv_meta[r][:hi - lo, 0].clamp(min=0).sum().item()) for r, (lo, hi) in enumerate(self.sizes)]
        for r in range(self.world):
            if r == self.dst:
                continue
            if counts[r]:
                dist.recv(self._recv_bytes[r][:counts[r]], r, group=self.group)
        return counts

    def __call__(self, lens, final_range, out):
        """Returns (lens, final_range, out) for ALL streams on dst (re-assembled [total, stride] slots), None elsewhere."""
        if self.world == 1:
            return lens, final_range, out
        counts = self.launch(lens, final_range, out)
        if self.rank != self.dst:
            return None
        ls, rs, os_ = [], [], []
        for r, (lo, hi) in enumerate(self.sizes):
            m = self._recv_meta[r][:hi - lo]
            l = m[:, 0].contiguous(); ls.append(l); rs.append(m[:, 1].contiguous())
            src = (self._packed.to(self.cdev) if self.stage_cpu else self._packed) if r == self.dst else self._recv_bytes[r]
            os_.append(unpack_packets(l, src, self.stride))
        return torch.cat(ls), torch.cat(rs), torch.cat(os_)
