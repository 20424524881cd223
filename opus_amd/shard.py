"""Stream sharding across the GPUs of one node.

Streams never interact (reference include/opus.h:425-429: separate state blobs), so stream s lives on exactly one
rank for its lifetime and a frame-step needs no data-path collective.  The only exchange is the final gather of
(length, final range, payload) to the rank that owns the output — RCCL over xGMI when the tensors are on GPUs
(torch.distributed backend "nccl"), gloo in the CPU tests.  Nothing here touches the codec itself.

The payload travels compacted and the step never waits for the host: every rank packs its packets back to back on the
device (exclusive prefix sum of the lengths, one scatter) into a wire record of FIXED size -- [lengths | final ranges |
packed bytes up to a capacity derived from the bitrate bound] -- so the transfer size is known without reading a byte
count back.  One `gather` collective per step moves the records of all ranks to dst at once (RCCL: every peer over its
own xGMI link in parallel), issued on a side stream behind an event recorded after the pack, into double-buffered
records: the gather of step t runs while the encoder of step t+1 does, and a record is reused only after the gather
that read it has completed (stream-side wait, no host synchronisation).  A step whose packets exceed the capacity is
flagged (`stats()["overflow"]`: a sticky device-side flag per rank, reduced over the ranks when the statistics are read), never silently truncated into a
valid-looking result.  The capacity itself comes from the encoder settings the caller names (`wire_capacity`: bitrate bound for VBR, the exact packet size for hard CBR).

Transports: "rccl" (the default on GPUs: one `gather` collective per step) and "p2p" -- every rank copies its record straight into dst's double buffer, mapped into
its address space through an IPC handle (a device-to-peer-device copy over the rank's own xGMI link, no collective; control traffic over a gloo group).  "p2p" does not
depend on ProcessGroupNCCL at all: `bench.py --gather p2p` still yields an N > 1 line with the exchange in the timed region should the RCCL gather fail on a node.
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
    """Packs out[s, :lens[s]] back to back into `packed` (last byte = spill slot).  Device-side, static shapes: bytes beyond a packet's
    length -- and, when `packed` is smaller than n * stride + 1, bytes beyond its capacity -- are scattered to the spill slot.  Returns the
    exclusive prefix sum of the lengths."""
    n, stride = out.shape
    l = lens.clamp(min=0).to(torch.int64)
    offs = torch.cumsum(l, 0) - l
    if out.is_cuda:                                                     # one wave per packet, coalesced byte copies (opusgpu_pack_packets_cap_dev, opus_amd.hip)
        from . import lib
        r = lib().opusgpu_pack_packets_cap_dev(out.data_ptr(), stride, lens.data_ptr(), offs.data_ptr(), packed.data_ptr(), n, packed.numel() - 1, torch.cuda.current_stream(out.device).cuda_stream)
        if r != 0: raise RuntimeError("opusgpu_pack_packets_cap_dev failed: %d" % r)
        return offs
    col = torch.arange(stride, device=out.device, dtype=torch.int64)
    spill = packed.numel() - 1
    idx = offs[:, None] + col[None, :]
    idx = torch.where((col[None, :] < l[:, None]) & (idx < spill), idx, torch.full((), spill, device=out.device, dtype=torch.int64))
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
    """Final gather of a frame-step's packets to `dst`, pipelined behind the next step's encode.

    Every rank passes its shard's lens [s_r] int32, final_range [s_r] int32 and out [s_r, stride] uint8; ragged shards
    (total % world != 0) are padded to the largest shard in the wire record and trimmed on `dst`.  `cap_per_stream`
    (default: stride, which can never overflow) sizes the packed part of the record: smax * cap_per_stream bytes.

        g.launch(lens, rng, out)   enqueue the exchange of one step (returns at once; at most `depth` steps in flight)
        g.flush()                  make the current stream (host, for CPU tensors) wait for every exchange enqueued so far
        g(lens, rng, out)          launch + flush + re-assembly on dst: (lens, final_range, out) of ALL streams, None elsewhere
    """

    @staticmethod
    def wire_capacity(stride, bitrate_bps=None, frame_rate=50, cbr=False, sub_streams=1):
        """Bytes per stream in the packed part of the wire record, from the encoder settings of the shard's streams.
        VBR / CVBR: twice the nominal packet (bitrate / 8 / frame_rate) plus 64 bytes for each elementary stream a packet carries (sub_streams > 1: multistream) -- the
        aggregate of a shard stays far below that, and an excess is flagged, not shipped as a valid-looking record; hard CBR: every packet IS the CBR size, so at least that,
        exactly; OPUS_BITRATE_MAX (-1), OPUS_AUTO (-1000) or no bitrate named: the whole output slot (`stride`), which cannot overflow."""
        if bitrate_bps is None or bitrate_bps <= 0: return int(stride)
        nominal = (int(bitrate_bps) + 8 * frame_rate - 1) // (8 * frame_rate)
        cap = nominal + 3 * sub_streams if cbr else 2 * nominal + 64 * sub_streams
        return int(min(stride, max(cap, nominal)))

    def __init__(self, total_streams, stride, device, dst=0, group=None, cap_per_stream=None, depth=2, transport=None, bitrate_bps=None, frame_rate=50, cbr=False, sub_streams=1):
        self.total, self.stride, self.dst, self.group, self.device = total_streams, stride, dst, group, torch.device(device)
        if cap_per_stream is None: cap_per_stream = self.wire_capacity(stride, bitrate_bps, frame_rate, cbr, sub_streams)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lo, self.hi = shard_range(total_streams, self.rank, self.world)
        self.sizes = [shard_range(total_streams, r, self.world) for r in range(self.world)]
        self.smax = max(hi - lo for lo, hi in self.sizes)
        self.cap = min(stride, int(cap_per_stream)) if cap_per_stream else stride
        self.depth = depth
        self.on_gpu = self.device.type == "cuda"
        # collectives on device tensors need a backend that moves them (nccl = RCCL); with gloo (tests) the record is staged through pinned host memory
        self.transport = transport or "rccl"
        if self.transport not in ("rccl", "p2p"): raise ValueError("transport must be 'rccl' or 'p2p'")
        self.p2p = self.transport == "p2p" and self.on_gpu and self.world > 1
        self.stage_cpu = dist.is_initialized() and dist.get_backend(group) == "gloo" and self.on_gpu and not self.p2p
        self.meta_bytes = self.smax * 8
        self.wire_bytes = self.meta_bytes + self.smax * self.cap + 8                                    # [lens int32 | final ranges int32 | packed bytes | spill slot + pad]
        self.wire_bytes = (self.wire_bytes + 15) & ~15
        self._wire = [torch.zeros(self.wire_bytes, dtype=torch.uint8, device=self.device) for _ in range(depth)]
        cdev = torch.device("cpu") if self.stage_cpu else self.device
        self.cdev = cdev
        self._host = [torch.zeros(self.wire_bytes, dtype=torch.uint8).pin_memory() for _ in range(depth)] if self.stage_cpu else None
        is_dst = self.world > 1 and self.rank == dst
        self._recv = [[torch.zeros(self.wire_bytes, dtype=torch.uint8, device=cdev) for _ in range(self.world)] for _ in range(depth)] if is_dst else None
        self._over = torch.zeros((), dtype=torch.int32, device=self.device)                          # sticky: some step of this rank exceeded the record's capacity
        self._ctl = None; self._peer = None
        if self.p2p: self._p2p_setup()
        self._side = torch.cuda.Stream(self.device) if self.on_gpu else None
        self._packed_ev = [torch.cuda.Event() for _ in range(depth)] if self.on_gpu else None
        self._done_ev = [None] * depth                                                                # GPU: event on the side stream after the gather of that slot
        self._work = [None] * depth                                                                   # CPU / staged: the outstanding gloo work of that slot
        self._staged = [False] * depth                                                                # staged: the device->host copy of that slot is enqueued, its gather is not
        self.steps = 0
        self.last_slot = None

    # -- "p2p": dst's receive buffers mapped into every rank --
    def _p2p_setup(self):
        """dst shares the IPC handles of its receive buffers; every rank maps the [depth] views of ITS slot in them.  Control traffic (the handles, the completion barrier of
        flush) goes over a gloo group of its own, so that nothing of this transport touches ProcessGroupNCCL."""
        from torch.multiprocessing.reductions import reduce_tensor
        # (the control group spans the ranks of `group`, not the world: new_group is a collective over the ranks it names; and dst reads _recv only after flush()'s barrier --
        # a peer overwrites its slot `depth` steps later without asking)
        ranks = dist.get_process_group_ranks(self.group) if self.group is not None else None
        self._ctl = dist.new_group(ranks=ranks, backend="gloo") if dist.get_backend(self.group) != "gloo" else self.group
        box = [[reduce_tensor(self._recv[d][r]) for r in range(self.world)] for d in range(self.depth)] if self.rank == self.dst else None
        got = [box]
        dist.broadcast_object_list(got, src=self.dst, group=self._ctl)
        if self.rank == self.dst: self._peer = [self._recv[d][self.rank] for d in range(self.depth)]
        else:
            self._peer = []
            for d in range(self.depth):
                fn, args = got[0][d][self.rank]
                self._peer.append(fn(*args))                                                          # a tensor of THIS process whose memory is dst's buffer
        dist.barrier(group=self._ctl)

    # -- slot lifecycle --
    def _issue_staged(self, slot):
        if self._staged[slot]:
            self._packed_ev[slot].synchronize()                                                       # the copy into pinned memory (long finished when this is called a step later)
            self._work[slot] = dist.gather(self._host[slot], self._recv[slot] if self.rank == self.dst else None, dst=self.dst, group=self.group, async_op=True)
            self._staged[slot] = False

    def _retire(self, slot):
        """the slot's previous exchange has completed before anything overwrites its record"""
        if self.on_gpu and not self.stage_cpu:
            if self._done_ev[slot] is not None: torch.cuda.current_stream(self.device).wait_event(self._done_ev[slot])
        else:
            if self.stage_cpu: self._issue_staged(slot)
            if self._work[slot] is not None: self._work[slot].wait(); self._work[slot] = None

    def launch(self, lens, final_range, out):
        """Enqueue the exchange of one step."""
        n = self.hi - self.lo
        if lens.shape[0] != n or out.shape != (n, self.stride):
            raise ValueError("shard shape mismatch")
        if self.world == 1:
            return None
        slot = self.steps % self.depth
        self._retire(slot)
        w = self._wire[slot]
        meta = w[:self.meta_bytes].view(torch.int32).view(2, self.smax)
        meta[0, :n].copy_(lens); meta[1, :n].copy_(final_range)
        pack_packets(lens, out, w[self.meta_bytes:self.meta_bytes + self.smax * self.cap + 1])
        self._over.copy_(torch.maximum(self._over, (lens.clamp(min=0).sum() > self.smax * self.cap).to(torch.int32)))     # (device-side, no synchronisation)
        if self.on_gpu:
            cur = torch.cuda.current_stream(self.device)
            ev = self._packed_ev[slot]
            if self.p2p:
                ev.record(cur)
                self._side.wait_event(ev)
                with torch.cuda.stream(self._side):
                    self._peer[slot].copy_(w, non_blocking=True)                                      # this rank's record -> its slot of dst's buffer, over its own link
                    d = torch.cuda.Event(); d.record(self._side); self._done_ev[slot] = d
            elif self.stage_cpu:
                ev0 = torch.cuda.Event(); ev0.record(cur)
                self._side.wait_event(ev0)
                with torch.cuda.stream(self._side):
                    self._host[slot].copy_(w, non_blocking=True)
                    ev.record(self._side)
                self._staged[slot] = True
                prev = (self.steps - 1) % self.depth                                                  # the host part of the PREVIOUS step's exchange runs now, under this step's encode
                if self.steps > 0 and prev != slot: self._issue_staged(prev)
            else:
                ev.record(cur)
                self._side.wait_event(ev)
                with torch.cuda.stream(self._side):
                    work = dist.gather(w, self._recv[slot] if self.rank == self.dst else None, dst=self.dst, group=self.group, async_op=True)
                    work.wait()                                                                       # stream-side: the side stream waits for RCCL, the host does not
                    d = torch.cuda.Event(); d.record(self._side); self._done_ev[slot] = d
        else:
            self._work[slot] = dist.gather(w, self._recv[slot] if self.rank == self.dst else None, dst=self.dst, group=self.group, async_op=True)
        self.last_slot = slot
        self.steps += 1
        return None

    def flush(self):
        if self.world == 1: return
        for slot in range(self.depth): self._retire(slot)
        if self.p2p:
            # the copies are this rank's own work on its side stream: dst learns that every peer's have landed through a barrier (host-side; flush ends a timed region anyway)
            self._side.synchronize()
            dist.barrier(group=self._ctl)

    def stats(self):
        """figures of the exchange for the bench line (reads the overflow flags back: call it after the timed region)"""
        over = False
        if self.world > 1:
            self.flush()
            if self.on_gpu: torch.cuda.current_stream(self.device).synchronize()
            # every rank's sticky flag (set on the device by every launch since the object was made, whatever has rotated out of the double buffer since), reduced over the ranks
            if self.p2p or self.stage_cpu or not self.on_gpu:
                f = self._over.detach().to("cpu").reshape(1).clone(); dist.all_reduce(f, op=dist.ReduceOp.MAX, group=self._ctl if self.p2p else self.group)
            else:
                f = self._over.detach().reshape(1).clone(); dist.all_reduce(f, op=dist.ReduceOp.MAX, group=self.group)
            over = bool(int(f.item()))
        return {"steps": self.steps, "wire_bytes_per_rank_per_step": self.wire_bytes, "cap_bytes_per_stream": self.cap, "slot_bytes_per_stream": self.stride, "overflow": bool(over), "in_flight": self.depth,
                "transport": "p2p: every rank copies its record into dst's buffer through an IPC mapping (no collective)" if self.p2p else "gloo, staged through pinned host memory (test hook)" if self.stage_cpu else ("RCCL gather on a side stream" if self.on_gpu else "gloo")}

    def __call__(self, lens, final_range, out):
        """Returns (lens, final_range, out) for ALL streams on dst (re-assembled [total, stride] slots), None elsewhere."""
        if self.world == 1:
            return lens, final_range, out
        self.launch(lens, final_range, out)
        self.flush()
        if self.rank != self.dst:
            return None
        if self.on_gpu and not self.stage_cpu: torch.cuda.synchronize(self.device)
        ls, rs, os_ = [], [], []
        for r, (lo, hi) in enumerate(self.sizes):
            w = self._recv[self.last_slot][r]
            meta = w[:self.meta_bytes].view(torch.int32).view(2, self.smax)
            l = meta[0, :hi - lo].contiguous(); ls.append(l); rs.append(meta[1, :hi - lo].contiguous())
            if int(l.clamp(min=0).sum().item()) > self.smax * self.cap: raise OverflowError("rank %d's packets exceed the wire record's capacity (%d bytes per stream): raise cap_per_stream" % (r, self.cap))
            os_.append(unpack_packets(l, w[self.meta_bytes:], self.stride))
        return torch.cat(ls), torch.cat(rs), torch.cat(os_)
