"""tests/transient_prepass_check.py — TEST INFRASTRUCTURE: a CELT-only encoder batch with the transient pre-pass (oa_celt_transient_kernel: one lane per (stream, channel), ahead of
the encode kernel) against one reference encoder per stream, packet bytes and final ranges.  The pre-pass belongs to wide launches (>= 64 streams); OPUS_AMD_TR_PRE=2 (read once per
process) or OPUS_AMD_SET_TRANSIENT_PREPASS(1) on the batch -- what this file does -- switches it on for the handful of streams the emulator can afford.
usage: transient_prepass_check.py emu|gpu"""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import capi

def signal(fsz, frames, ch, seed, Fs):
    rng = np.random.default_rng(seed); t = np.arange(fsz * frames) / Fs
    x = np.zeros((fsz * frames, ch))
    for c in range(ch):
        x[:, c] = 7000 * np.sin(2 * np.pi * (180 + 70 * seed + 31 * c) * t) * (0.3 + 0.7 * (np.sin(2 * np.pi * 2.5 * t + seed) > 0)) + rng.normal(0, 200 + 100 * c, len(t))
        for k in range(frames // 3):                                # onsets: the frames the transient analysis exists for
            at = (3 * k + 1) * fsz + (seed * 37 + 211 * k) % fsz
            x[at:at + 300, c] += 14000 * np.sign(np.sin(2 * np.pi * (2500 + 400 * c) * t[at:at + 300]))
        if seed % 3 == 0: x[4 * fsz:5 * fsz, c] = 0                 # a frame of digital silence
    return np.clip(x, -32768, 32767).astype(np.int16)

def run_schedules(L, n):
    """the guards of the pre-pass's record: the frame size changes between calls (tr[2] == len), force_channels 1 <-> 2 on a stereo encoder, hard CBR with bitrate changes,
    byte budgets of 3 and 20 -- every change applied to the batch and to the reference encoders at the same frame"""
    Fs, ch = 48000, 2
    sched = [(960, 1275, {}), (480, 1275, {}), (240, 1275, {}), (960, 1275, {"force_channels": 1}), (120, 1275, {}), (960, 1275, {"force_channels": 2}), (480, 20, {}), (960, 3, {}),
             (960, 1275, {"vbr": 0, "bitrate": 64000}), (960, 1275, {"bitrate": 24000}), (480, 1275, {"bitrate": 200000}), (960, 1275, {"vbr": 1, "force_channels": -1000}), (240, 20, {}), (960, 1275, {})]
    err = ctypes.c_int()
    b = L.opusgpu_enc_batch_create(n, Fs, ch, 2051, 0, ctypes.byref(err)); assert b, err.value
    refs = [capi.Enc("ref", Fs, ch, 2051, bitrate=96000, complexity=10) for _ in range(n)]
    for k, v in dict(bitrate=96000, complexity=10).items(): assert L.opusgpu_enc_batch_ctl(b, -1, capi.REQ[k], v) == 0
    assert L.opusgpu_enc_batch_ctl(b, -1, 11900, 0) == 0 and L.opusgpu_enc_batch_ctl(b, -1, 11906, 1) == 0
    total = sum(f for f, _, _ in sched)
    sig = [signal(total, 1, ch, 3 * s + 1, Fs) for s in range(n)]
    pos = 0
    for j, (fsz, mdb, ctl) in enumerate(sched):
        for k, v in ctl.items():
            assert L.opusgpu_enc_batch_ctl(b, -1, capi.REQ[k], v) == 0
            for r in refs: assert r.set(k, v) == 0
        pcm = np.ascontiguousarray(np.stack([sig[s][pos:pos + fsz] for s in range(n)])); pos += fsz
        o = np.zeros((n, 1500), np.uint8); lens = np.zeros(n, np.int32); rng = np.zeros(n, np.uint32)
        assert L.opusgpu_encode_batch(b, pcm.ctypes.data, fsz, o.ctypes.data, 1500, mdb, lens.ctypes.data, rng.ctypes.data) == 0
        for s in range(n):
            pk, ln, fr = refs[s].encode(pcm[s], fsz, mdb)
            assert ln == int(lens[s]) and (ln < 0 or (pk == bytes(o[s, :ln]) and fr == int(rng[s]))), ("schedule", j, s, fsz, mdb, ctl, ln, int(lens[s]))
    L.opusgpu_enc_batch_destroy(b)
    print("transient pre-pass: schedule of %d calls with frame size / channel / CBR / budget changes equal to the reference" % len(sched))

def run(which, n=5, frames=14):
    L = capi.load(which)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    L.opusgpu_enc_batch_create.restype = vp; L.opusgpu_enc_batch_create.argtypes = [i32, i32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.opusgpu_enc_batch_ctl.argtypes = [vp, i32, ctypes.c_int, i32]
    L.opusgpu_encode_batch.argtypes = [vp, vp, ctypes.c_int, vp, i32, i32, vp, vp]
    L.opusgpu_enc_batch_destroy.argtypes = [vp]; L.opusgpu_enc_batch_destroy.restype = None
    cases = [(48000, 2, 960, dict(bitrate=128000, complexity=10)), (48000, 1, 960, dict(bitrate=64000, complexity=10)), (48000, 2, 480, dict(bitrate=96000, complexity=5)),
             (48000, 1, 240, dict(bitrate=64000, complexity=10)), (48000, 2, 120, dict(bitrate=160000, complexity=8)),
             (48000, 2, 960, dict(bitrate=24000, complexity=10)),        # low rate: the stereo width fade takes the frame's own path
             (48000, 2, 960, dict(bitrate=96000, complexity=0)),         # complexity 0: no transient analysis at all
             (24000, 2, 480, dict(bitrate=64000, complexity=10))]        # 24 kHz API rate: the pre-pass does not run
    run_schedules(L, n)
    for Fs, ch, fsz, ctl in cases:
        err = ctypes.c_int()
        b = L.opusgpu_enc_batch_create(n, Fs, ch, 2051, 0, ctypes.byref(err)); assert b, err.value
        refs = [capi.Enc("ref", Fs, ch, 2051, **ctl) for _ in range(n)]
        for k, v in ctl.items(): assert L.opusgpu_enc_batch_ctl(b, -1, capi.REQ[k], v) == 0
        assert L.opusgpu_enc_batch_ctl(b, -1, 11900, 0) == 0           # (the fixed-point reference without the float API is the comparison here)
        assert L.opusgpu_enc_batch_ctl(b, -1, 11906, 1) == 0           # the lane pre-pass, whatever the width of the launch
        sig = [signal(fsz, frames, ch, 3 * s + ch, Fs) for s in range(n)]
        for f in range(frames):
            pcm = np.ascontiguousarray(np.stack([sig[s][f * fsz:(f + 1) * fsz] for s in range(n)]))
            o = np.zeros((n, 1500), np.uint8); lens = np.zeros(n, np.int32); rng = np.zeros(n, np.uint32)
            assert L.opusgpu_encode_batch(b, pcm.ctypes.data, fsz, o.ctypes.data, 1500, 1275, lens.ctypes.data, rng.ctypes.data) == 0
            for s in range(n):
                pk, ln, fr = refs[s].encode(pcm[s], fsz, 1275)
                assert ln == int(lens[s]) and pk == bytes(o[s, :ln]) and fr == int(rng[s]), (Fs, ch, fsz, ctl, f, s, ln, int(lens[s]))
        L.opusgpu_enc_batch_destroy(b)
    print("transient pre-pass: %d cases x %d streams x %d frames equal to the reference" % (len(cases), n, frames))

if __name__ == "__main__":
    run(sys.argv[1] if len(sys.argv) > 1 else "emu")
