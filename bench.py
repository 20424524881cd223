#!/usr/bin/env python3
"""bench.py — encoded frames/s of the batched CELT-only Opus encoder on N MI355X (one process per GPU).

Workload = BASELINE.json configs[1]: CELT-only encode, OPUS_APPLICATION_RESTRICTED_LOWDELAY, 48 kHz stereo, 20 ms frames,
128 kb/s CVBR, complexity 10, 65,536 independent streams per GPU; a "step" = one 20 ms frame-step of every stream
(65,536 frames per GPU), state carried in HBM between steps, PCM resident in HBM before the timed region.
Streams shard across ranks with no data-path collective; the only exchange is the final gather of (length, payload)
to rank 0 over RCCL, included in the timed region when N > 1 (weak scaling: per-GPU work fixed).

Prints ONE JSON line (rank 0): metric/value + "roofline" (HBM-bound, algorithmic bytes / measured kernel time) and
"cpu_baseline" (the compiled reference on one host core, bounded sample).
"""
import argparse, ctypes, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def cpu_baseline(seconds=12.0):
    """Reference libopus (default float build, RTCD/AVX2) on ONE host core, same encoder settings, same kind of signal."""
    import signals
    path = os.path.join(ROOT, "oracle/_ref/libopus_ref_fl.so")
    kind = "reference"
    sig = signals.music(500, seed=0)
    if os.path.exists(path):
        L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        L.opus_encoder_create.restype = ctypes.c_void_p
        L.opus_encoder_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.opus_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        err = ctypes.c_int()
        st = L.opus_encoder_create(48000, 2, 2051, ctypes.byref(err))
        L.opus_encoder_ctl(st, 4002, 128000); L.opus_encoder_ctl(st, 4010, 10)
        out = (ctypes.c_ubyte * 1500)()
        enc = lambda ptr: L.opus_encode(st, ptr, 960, out, 1276)
    else:   # the compiled reference did not travel: time our plain-C port instead
        from test_oracle_encoder import OracleEnc
        kind = "port"
        o = OracleEnc(2, bitrate=128000, complexity=10)
        enc = lambda ptr: o.O.oc_opus_encode(o.buf, ptr, 960, o.out, 1276)
    try: os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
    except Exception: pass
    base = sig.ctypes.data
    n = 0; t0 = time.perf_counter()
    while True:
        for i in range(500):
            enc(base + i * 960 * 2 * 2)
        n += 500
        if time.perf_counter() - t0 > seconds: break
    dt = time.perf_counter() - t0
    try: os.sched_setaffinity(0, set(range(os.cpu_count())))
    except Exception: pass
    return {"value": round(n / dt, 1), "unit": "frames/s", "cores": 1, "kind": kind,
            "sample": "%d consecutive 20 ms frames of one synthetic stereo stream (same settings), %.1f s, 1 thread pinned" % (n, dt)}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=65536, help="streams per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode", action="store_true", help="also time the HIP decoder on the packets just produced (extra JSON field, not the headline metric)")
    a = ap.parse_args()
    import torch, torch.distributed as dist
    import opus_amd, signals
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        print("bench.py: launch with torch.distributed.run for --gpus > 1", file=sys.stderr); sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible — the product path has no CPU fallback", file=sys.stderr); sys.exit(3)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    S, K, W, FR, CH = a.streams, a.steps, a.warmup, 960, 2
    T = K + W
    # ---- synthetic input, resident in HBM: a pool of 256 distinct signals, every stream = pool member with its own time offset and gain
    P = 256
    pool = np.stack([(signals.music(T + 2, seed=1000 * rank + p) if p % 4 else signals.noise_bursts(T + 2, seed=1000 * rank + p)).reshape(-1) for p in range(P)])
    pool_d = torch.from_numpy(pool).to(dev)                                  # [P, (T+2)*1920] int16
    g = torch.Generator(device="cpu"); g.manual_seed(1234 + rank)
    pid = torch.randint(0, P, (S,), generator=g).to(dev)
    off = (torch.randint(0, 960, (S,), generator=g) * CH).to(dev)           # sample-aligned start offset (both channels)
    gain = (0.5 + 0.5 * torch.rand((S,), generator=g)).to(dev)
    pcm = torch.empty((T, S, FR * CH), dtype=torch.int16, device=dev)
    ar = torch.arange(FR * CH, device=dev)
    for t in range(T):
        idx = off[:, None] + t * FR * CH + ar[None, :]
        x = pool_d[pid[:, None], idx].to(torch.float32) * gain[:, None]
        pcm[t] = x.round().clamp(-32768, 32767).to(torch.int16)
    del pool_d
    out = torch.zeros((S, 1280), dtype=torch.uint8, device=dev)
    lens = torch.zeros((S,), dtype=torch.int32, device=dev)
    rng = torch.zeros((S,), dtype=torch.int32, device=dev)
    b = opus_amd.EncoderBatch(S, channels=CH, application=opus_amd.OPUS_APPLICATION_RESTRICTED_LOWDELAY, device=local)
    b.ctl(opus_amd.OPUS_SET_BITRATE_REQUEST, 128000); b.ctl(opus_amd.OPUS_SET_COMPLEXITY_REQUEST, 10)
    stream = torch.cuda.current_stream(dev)
    from opus_amd.shard import PacketGather
    gather = PacketGather(S * world, 1280, dev, dst=0) if world > 1 else None

    def step(t):
        b.encode_dev(pcm[t].data_ptr(), FR, out.data_ptr(), 1280, lens.data_ptr(), rng.data_ptr(), hip_stream=stream.cuda_stream)
        if world > 1:   # the only exchange of the path: final gather of the packets (RCCL over xGMI)
            gather.launch(lens, rng, out)

    for t in range(W): step(t)
    torch.cuda.synchronize(dev)
    if world > 1: dist.barrier()
    torch.cuda.synchronize(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    t0 = time.perf_counter()
    for k in range(K):
        ev[k][0].record(stream)
        b.encode_dev(pcm[W + k].data_ptr(), FR, out.data_ptr(), 1280, lens.data_ptr(), rng.data_ptr(), hip_stream=stream.cuda_stream)
        ev[k][1].record(stream)
        if world > 1:
            gather.launch(lens, rng, out)
    torch.cuda.synchronize(dev)
    if world > 1: dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1: dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    dec = None
    if a.decode and world == 1:
        # decoder leg: re-encode the K timed frames keeping every packet, then decode them with the state carried (device-resident throughout)
        b.reset()
        outs = torch.zeros((K + W, S, 1280), dtype=torch.uint8, device=dev); lns = torch.zeros((K + W, S), dtype=torch.int32, device=dev)
        for t in range(K + W):
            b.encode_dev(pcm[t].data_ptr(), FR, outs[t].data_ptr(), 1280, lns[t].data_ptr(), rng.data_ptr(), hip_stream=stream.cuda_stream)
        d = opus_amd.DecoderBatch(S, channels=CH, device=local)
        dpcm = torch.zeros((S, FR * CH), dtype=torch.int16, device=dev); dns = torch.zeros((S,), dtype=torch.int32, device=dev); drng = torch.zeros((S,), dtype=torch.int32, device=dev)
        for t in range(W): d.decode_dev(outs[t].data_ptr(), 1280, lns[t].data_ptr(), dpcm.data_ptr(), FR, dns.data_ptr(), drng.data_ptr(), hip_stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for t in range(W, W + K): d.decode_dev(outs[t].data_ptr(), 1280, lns[t].data_ptr(), dpcm.data_ptr(), FR, dns.data_ptr(), drng.data_ptr(), hip_stream=stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        dms = e0.elapsed_time(e1) / K
        dstate = ctypes.CDLL(opus_amd.LIB_PATH).opusgpu_dec_state_size()
        mean_l = float(lns[W:].float().mean().item())
        # algorithmic bytes/frame: packet in + PCM out + state touched: scalars/energies (in+out) + overlap (in+out) + N new history samples out (+ history taps read by the post-filter, <= 2*1030 words, not counted)
        dalg = mean_l + FR * CH * 2 + 2 * (128 + 4 * 168 + 960) + FR * CH * 4
        dec = {"metric": "decoded frames/s (48 kHz stereo, 20 ms CELT packets)", "value": round(S / (dms * 1e-3), 1), "kernel": "oa_decode_kernel", "kernel_ms": round(dms, 3),
               "all_frames_ok": bool((dns == FR).all().item()), "state_bytes": dstate, "algorithmic_bytes_per_frame": round(dalg, 1),
               "roofline": {"bound": "hbm", "achieved": round(S * dalg / (dms * 1e-3) / 1e9, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(S * dalg / (dms * 1e-3) / 1e9 / 8000.0, 5)}}
        d.close()
    lens_h = lens.cpu().numpy()
    ok = bool((lens_h > 2).all())
    mean_len = float(lens_h.mean())
    if rank == 0:
        frames = S * world * K
        state_bytes = ctypes.CDLL(opus_amd.LIB_PATH).opusgpu_enc_state_size()
        alg_bytes = S * (FR * CH * 2 + mean_len + 4 + 4 + 2 * state_bytes)       # per launch: PCM in + packet/len/range out + state in and out
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None        # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 calibration + WRITE_SIZE), scaled to this launch
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            traffic = int(pt["hbm_bytes_per_frame"] * S)
        except Exception:
            pass
        res = {
            "metric": "encoded frames/s (48 kHz stereo, 20 ms, complexity 10)", "value": round(frames / dt, 1), "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": "CELT-only encode, restricted-lowdelay, 48 kHz stereo, 20 ms, CVBR 128 kb/s, complexity 10, bit-exact fixed-point",
                       "streams_per_gpu": S, "frames_per_step": S * world, "mean_packet_bytes": round(mean_len, 1), "all_packets_valid": ok,
                       "parallelism": "streams sharded over %d GPU(s), no data-path collective%s" % (world, ", final RCCL gather in timed region" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                         "kernel": "oa_encode_kernel", "kernel_ms": round(kern_ms, 3), "algorithmic_bytes_per_frame": round(alg_bytes / S, 1),
                         "note": "latency/issue-bound integer codec path: HBM fraction is small by construction (SURVEY.md 8d)"},
        }
        if dec is not None: res["decode"] = dec
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
            res["speedup_vs_cpu_1core"] = round(res["value"] / res["cpu_baseline"]["value"], 2)
        print(json.dumps(res))
    b.close()
    if world > 1: dist.destroy_process_group()

if __name__ == "__main__":
    main()
