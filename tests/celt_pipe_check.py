"""tests/celt_pipe_check.py — TEST INFRASTRUCTURE: a CELT-only encoder batch through the kernel pipeline (oa_encode_kernel cut before the PVQ -> oa_celt_pvq_kernel, four streams
per wave, celt_enc_pvq4.h -> oa_celt_back_kernel; OPUS_AMD_SET_KERNEL_PIPELINE(1) forces it for the handful of streams the emulator can afford) against one reference encoder
per stream: packet bytes, lengths and final ranges of every frame.
usage: celt_pipe_check.py emu|gpu [case-substring]"""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import capi

def signal(fsz, frames, ch, seed, Fs, kind):
    rng = np.random.default_rng(seed); t = np.arange(fsz * frames) / Fs
    x = np.zeros((fsz * frames, ch))
    for c in range(ch):
        if kind == "tone":
            x[:, c] = 7000 * np.sin(2 * np.pi * (180 + 70 * seed + 31 * c) * t) * (0.3 + 0.7 * (np.sin(2 * np.pi * 2.5 * t + seed) > 0)) + rng.normal(0, 200 + 100 * c, len(t))
            for k in range(frames // 3):
                at = (3 * k + 1) * fsz + (seed * 37 + 211 * k) % fsz
                x[at:at + 300, c] += 14000 * np.sign(np.sin(2 * np.pi * (2500 + 400 * c) * t[at:at + 300]))
        elif kind == "noise":
            x[:, c] = rng.normal(0, 3000, len(t)) * (0.2 + 0.8 * (np.sin(2 * np.pi * 1.3 * t + c) > 0))
        elif kind == "wide":                                        # uncorrelated channels, rich spectrum: dual stereo, many pulses
            x[:, c] = sum(2500 / (1 + 0.3 * k) * np.sin(2 * np.pi * (110 * (c + 1) + 437 * k + 13 * seed) * t + k) for k in range(28)) + rng.normal(0, 800, len(t))
        elif kind == "corr":                                        # nearly identical channels: small side, intensity
            base = sum(3000 / (1 + 0.5 * k) * np.sin(2 * np.pi * (200 + 331 * k + 7 * seed) * t) for k in range(20))
            x[:, c] = base * (1 + 0.05 * c) + rng.normal(0, 30, len(t))
        if seed % 3 == 0 and frames > 5: x[4 * fsz:5 * fsz, c] = 0
    return np.clip(x, -32768, 32767).astype(np.int16)

CASES = [   # name, Fs, channels, frame size, signal, ctls, max_data_bytes
    ("config2",      48000, 2, 960, "tone",  dict(bitrate=128000, complexity=10), 1275),
    ("wide128",      48000, 2, 960, "wide",  dict(bitrate=128000, complexity=10), 1275),
    ("corr64",       48000, 2, 960, "corr",  dict(bitrate=64000, complexity=10), 1275),
    ("mono64",       48000, 1, 960, "tone",  dict(bitrate=64000, complexity=10), 1275),
    ("noise10ms",    48000, 2, 480, "noise", dict(bitrate=96000, complexity=5), 1275),
    ("mono10ms",     48000, 1, 480, "wide",  dict(bitrate=48000, complexity=10), 1275),
    ("low24",        48000, 2, 960, "wide",  dict(bitrate=24000, complexity=10), 1275),
    ("c0",           48000, 2, 960, "tone",  dict(bitrate=96000, complexity=0), 1275),
    ("c8_256k",      48000, 2, 960, "wide",  dict(bitrate=256000, complexity=8), 1275),
    ("max510k",      48000, 2, 960, "noise", dict(bitrate=510000, complexity=10), 1275),
    ("cbr64",        48000, 2, 960, "tone",  dict(bitrate=64000, complexity=10, vbr=0), 1275),
    ("tight",        48000, 2, 960, "wide",  dict(bitrate=128000, complexity=10), 60),
    ("fs24k",        24000, 2, 480, "wide",  dict(bitrate=64000, complexity=10), 1275),
    ("fs16k_mono",   16000, 1, 320, "tone",  dict(bitrate=32000, complexity=9), 1275),
    ("short5ms",     48000, 2, 240, "tone",  dict(bitrate=96000, complexity=10), 1275),     # under 10 ms: the encode kernel keeps the whole call
    ("mb_narrow",    48000, 2, 960, "wide",  dict(bitrate=96000, complexity=10, max_bandwidth=1103), 1275),
]

def run(which, only=None, n=5, frames=10):
    L = capi.load(which)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    L.opusgpu_enc_batch_create.restype = vp; L.opusgpu_enc_batch_create.argtypes = [i32, i32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.opusgpu_enc_batch_ctl.argtypes = [vp, i32, ctypes.c_int, i32]
    L.opusgpu_encode_batch.argtypes = [vp, vp, ctypes.c_int, vp, i32, i32, vp, vp]
    L.opusgpu_enc_batch_destroy.argtypes = [vp]; L.opusgpu_enc_batch_destroy.restype = None
    L.opusgpu_enc_batch_split_stats.argtypes = [vp, vp, vp]
    done = 0
    for name, Fs, ch, fsz, kind, ctl, mdb in CASES:
        if only and only not in name: continue
        err = ctypes.c_int()
        b = L.opusgpu_enc_batch_create(n, Fs, ch, 2051, 0, ctypes.byref(err)); assert b, err.value
        refs = [capi.Enc("ref", Fs, ch, 2051, **ctl) for _ in range(n)]
        for k, v in ctl.items(): assert L.opusgpu_enc_batch_ctl(b, -1, capi.REQ[k], v) == 0
        assert L.opusgpu_enc_batch_ctl(b, -1, 11900, 0) == 0           # (the fixed-point reference without the float API is the comparison here)
        assert L.opusgpu_enc_batch_ctl(b, -1, 11902, 1) == 0           # the kernel pipeline, whatever the width of the launch
        sig = [signal(fsz, frames, ch, 3 * s + ch, Fs, kind) for s in range(n)]
        cut_total = whole_total = 0
        for f in range(frames):
            pcm = np.ascontiguousarray(np.stack([sig[s][f * fsz:(f + 1) * fsz] for s in range(n)]))
            o = np.zeros((n, 1500), np.uint8); lens = np.zeros(n, np.int32); rng = np.zeros(n, np.uint32)
            assert L.opusgpu_encode_batch(b, pcm.ctypes.data, fsz, o.ctypes.data, 1500, mdb, lens.ctypes.data, rng.ctypes.data) == 0
            kept, decl = ctypes.c_uint32(), ctypes.c_uint32()
            assert L.opusgpu_enc_batch_split_stats(b, ctypes.byref(kept), ctypes.byref(decl)) == 0
            cut_total += kept.value; whole_total += decl.value
            for s in range(n):
                pk, ln, fr = refs[s].encode(pcm[s], fsz, mdb)
                assert ln == int(lens[s]) and pk == bytes(o[s, :ln]) and fr == int(rng[s]), (name, f, s, ln, int(lens[s]), fr, int(rng[s]))
        L.opusgpu_enc_batch_destroy(b)
        done += 1
        assert (cut_total > 0) == (fsz * 100 >= Fs), (name, cut_total, whole_total)        # 10 / 20 ms calls go through the pipeline (silent frames and the like excepted)
        print("  %s ok (%d calls cut before the PVQ, %d kept whole)" % (name, cut_total, whole_total), flush=True)
    print("CELT kernel pipeline: %d cases x %d streams x %d frames equal to the reference" % (done, n, frames))

if __name__ == "__main__":
    run(sys.argv[1] if len(sys.argv) > 1 else "emu", sys.argv[2] if len(sys.argv) > 2 else None)
