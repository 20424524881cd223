#!/usr/bin/env python3
"""tools/split_check.py — the SILK-capable encoder's split path (front / quantiser / back kernels, opus_amd/csrc/opus_sh_split.h) against its one-kernel path:
the same batches through OPUS_AMD_SH_SPLIT = 0 .. 4 (one subprocess each: the process-wide default of OPUS_AMD_SET_KERNEL_PIPELINE) must give identical packets, final ranges AND
identical stream records, byte for byte.  usage: split_check.py [emu|gpu]      (child: split_check.py <lib> <mode> <out.pkl>)"""
import os, sys, ctypes, pickle, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)

def speech(fs, secs, ch, seed):
    rng = np.random.default_rng(seed); t = np.arange(int(fs * secs)) / fs; outs = []
    for c in range(ch):
        f0 = 120 + 30 * np.sin(2 * np.pi * 0.7 * t + c + seed) + 15 * c + (seed % 7) * 9
        ph = 2 * np.pi * np.cumsum(f0) / fs
        s = sum(np.sin(k * ph) / k for k in range(1, 25) if k * 220 < fs / 2) * (np.sin(2 * np.pi * 1.5 * t + 0.3 * c + seed) > -0.3) * 6000 + rng.normal(0, 60 + 400 * (t > secs * 0.7), len(t))
        outs.append(s)
    return np.clip(np.stack(outs, 1), -32768, 32767).astype(np.int16)

CASES = {   # name: (Fs, channels, application, streams, frames, ms, {ctl: value}, per-stream ctl overrides)
    "config3":      (16000, 1, 2048, 37, 12, 20, {4002: 24000, 4010: 10, 11002: 1000, 11900: 0}, {}),
    "config4":      (48000, 2, 2049, 19, 8, 20, {4002: 128000, 4010: 10, 11002: 1001, 11900: 0}, {}),
    "config3_fec":  (16000, 1, 2048, 9, 12, 20, {4002: 24000, 4010: 10, 11002: 1000, 11900: 0, 4012: 1, 4014: 10}, {1: {4014: 25}, 2: {4002: 40000}, 3: {4010: 5}}),      # in-band FEC through the pipeline (round 6): the LBRR pass in the quantiser kernel, the side stream at the head of the next packet
    "stereo_fec":   (48000, 2, 2049, 6, 12, 20, {4002: 36000, 4010: 10, 4012: 1, 4014: 15}, {1: {4002: 20000}, 2: {4012: 2}, 3: {4006: 0}}),
    "fec_10ms":     (16000, 1, 2048, 5, 16, 10, {4002: 20000, 4010: 8, 11002: 1000, 4012: 1, 4014: 20}, {}),
    "silk_60ms":    (16000, 1, 2048, 7, 8, 60, {4002: 24000, 4010: 10, 11002: 1000, 11900: 0}, {1: {4002: 14000}, 2: {4006: 0}, 3: {4010: 4}, 4: {4016: 1}}),      # 40 / 60 ms SILK packets through the pipeline (round 6): the front -> pred -> quantiser relay once per 20 ms frame
    "silk_40ms_fec": (16000, 1, 2048, 6, 10, 40, {4002: 28000, 4010: 10, 11002: 1000, 4012: 1, 4014: 15}, {1: {4014: 30}, 2: {4002: 16000}}),
    "stereo_60ms":  (24000, 2, 2048, 5, 6, 60, {4002: 30000, 4010: 9, 11002: 1000, 4008: 1103}, {1: {4012: 1, 4014: 10}, 2: {4002: 18000}}),
    "audio_40ms":   (48000, 2, 2049, 5, 8, 40, {4002: 24000, 4010: 10}, {1: {4002: 64000}, 2: {11002: 1000, 4008: 1103}}),
    "voip_auto":    (16000, 1, 2048, 5, 10, 20, {4002: 16000, 4010: 10}, {}),
    "cbr_12k":      (16000, 1, 2048, 6, 10, 20, {4002: 12000, 4010: 8, 4006: 0, 11002: 1000, 11900: 0}, {}),
    "tight_cvbr":   (16000, 1, 2048, 6, 10, 20, {4002: 9000, 4010: 6, 11002: 1000, 11900: 0}, {}),
    "10ms_nb_mb":   (48000, 1, 2049, 6, 14, 10, {4002: 14000, 4010: 5, 11002: 1000}, {1: {4008: 1101}, 2: {4008: 1102}, 3: {4010: 2}, 4: {4010: 0}}),
    "audio_auto":   (48000, 2, 2049, 6, 14, 20, {4002: 40000, 4010: 10}, {1: {4002: 20000}, 2: {4002: 96000}, 3: {4012: 1, 4014: 10}, 4: {4016: 1}}),
    "stereo_silk":  (24000, 2, 2048, 5, 10, 20, {4002: 30000, 4010: 9, 11002: 1000}, {1: {4022: 1}, 2: {4002: 14000}}),
    # CELT-only frames of the SILK-capable applications: kept by the front kernel since round 4's last day (no quantiser job, the back kernel codes them whole)
    "audio_celt":   (48000, 2, 2049, 7, 12, 20, {4002: 96000, 4010: 10}, {1: {4002: 160000}, 2: {4006: 0}, 3: {4010: 5}, 4: {4022: 1}, 5: {4020: 0}, 6: {4008: 1104}}),
    "celt_10ms":    (48000, 1, 2049, 6, 16, 10, {4002: 64000, 4010: 10}, {1: {4006: 0}, 2: {4046: 1}}),
    "celt_5ms":     (48000, 2, 2049, 5, 20, 5, {4002: 96000, 4010: 8}, {}),
    "celt_2_5ms":   (24000, 1, 2048, 5, 24, 2.5, {4002: 40000, 4010: 10}, {}),
    "voip_celt":    (16000, 1, 2048, 5, 10, 20, {4002: 40000, 4010: 10, 11002: 1002}, {}),
    # mode switches in both directions inside the run (redundancy frames, CELT prefill, SILK prefill -> those calls go to the one-kernel path, their neighbours stay): SCHEDULE below
    "switching":    (48000, 2, 2049, 6, 18, 20, {4002: 24000, 4010: 10}, {1: {4022: 1}, 2: {4006: 0}}),
}
SCHEDULE = {"config3_fec": {6: {4012: 0}, 9: {4012: 1}}, "switching": {4: {4002: 96000}, 8: {4002: 20000}, 11: {11002: 1002}, 14: {11002: -1000, 4002: 32000}}}     # frame -> {ctl: value} for every stream

def selected_cases():
    """SPLIT_CHECK_CASES=name,name,...: a subset (the default CPU suite runs the quick one: tests/test_hostemu_split.py)"""
    want = os.environ.get("SPLIT_CHECK_CASES")
    return CASES if not want else {k: v for k, v in CASES.items() if k in want.split(",")}

def run_child(libpath, out):
    L = ctypes.CDLL(libpath)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    L.opusgpu_enc_batch_create.restype = vp; L.opusgpu_enc_batch_create.argtypes = [i32, i32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.opusgpu_enc_batch_ctl.argtypes = [vp, i32, ctypes.c_int, i32]
    L.opusgpu_encode_batch.argtypes = [vp, vp, ctypes.c_int, vp, i32, i32, vp, vp]
    L.opusgpu_enc_batch_export_state.argtypes = [vp, i32, vp]
    L.opusgpu_enc_batch_split_stats.argtypes = [vp, vp, vp]; L.opusgpu_enc_batch_pvq_stage_stats.argtypes = [vp, vp]
    L.opusgpu_enc_batch_destroy.argtypes = [vp]; L.opusgpu_enc_batch_destroy.restype = None
    res = {}
    for name, (Fs, ch, app, n, frames, ms, ctl, per) in selected_cases().items():
        err = ctypes.c_int()
        b = L.opusgpu_enc_batch_create(n, Fs, ch, app, 0, ctypes.byref(err)); assert b, err.value
        for k, v in ctl.items(): assert L.opusgpu_enc_batch_ctl(b, -1, k, v) == 0, (name, k, v)
        for s, d in per.items():
            for k, v in d.items(): assert L.opusgpu_enc_batch_ctl(b, s, k, v) == 0, (name, s, k, v)
        fsz = int(Fs * ms // 1000)
        sig = [speech(Fs, frames * ms / 1000 + 0.1, ch, s) for s in range(n)]
        pk = []; pvq_calls = 0
        for f in range(frames):
            pcm = np.stack([np.ascontiguousarray(sig[s][f * fsz:(f + 1) * fsz]).reshape(-1) for s in range(n)]).astype(np.int16)
            if f == frames // 2: pcm[0] = 0                      # a frame of digital silence
            for k, v in SCHEDULE.get(name, {}).get(f, {}).items(): assert L.opusgpu_enc_batch_ctl(b, -1, k, v) == 0, (name, f, k, v)
            o = np.zeros((n, 1500), np.uint8); lens = np.zeros(n, np.int32); rng = np.zeros(n, np.uint32)
            r = L.opusgpu_encode_batch(b, pcm.ctypes.data, fsz, o.ctypes.data, 1500, 1275, lens.ctypes.data, rng.ctypes.data); assert r == 0, (name, r)
            pv = ctypes.c_uint32(); L.opusgpu_enc_batch_pvq_stage_stats(b, ctypes.byref(pv)); pvq_calls += pv.value
            pk.append((lens.copy(), rng.copy(), [bytes(o[s, :max(lens[s], 0)]) for s in range(n)]))
        sz = L.opusgpu_enc_sh_state_size(); blobs = []
        for s in range(n):
            bl = np.zeros(sz, np.uint8); assert L.opusgpu_enc_batch_export_state(b, s, bl.ctypes.data) == 0; blobs.append(bl)
        k = ctypes.c_uint32(); d = ctypes.c_uint32(); L.opusgpu_enc_batch_split_stats(b, ctypes.byref(k), ctypes.byref(d))
        L.opusgpu_enc_batch_destroy(b)
        res[name] = (pk, blobs, (k.value, d.value), pvq_calls)
    pickle.dump(res, open(out, "wb"))

def compare(which="emu", tmpdir="/tmp", verbose=True):
    if which == "emu":
        import hostemu; lib = hostemu.build_emu_lib()
    else: lib = os.path.join(ROOT, "opus_amd/libopus_amd.so")
    r = {}
    for mode in "01234":
        out = os.path.join(tmpdir, "split_check_%s_%s_%d.pkl" % (which, mode, os.getpid()))
        env = dict(os.environ, OPUS_AMD_SH_SPLIT=mode)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), lib, mode, out], env=env)
        r[mode] = pickle.load(open(out, "rb")); os.unlink(out)
    bad = []
    for name in selected_cases():
        a = r["0"][name]
        assert a[2] == (0, 0), "the one-kernel run went through the split path?"
        for mode in "1234":
            b = r[mode][name]
            okp = all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2] for x, y in zip(a[0], b[0]))
            nd = [int((x != y).sum()) for x, y in zip(a[1], b[1])]
            if verbose: print("%-12s mode %s: packets %s, state bytes differing per stream %s, calls kept / handed back %s, CELT frames through the PVQ stage %d" % (name, mode, "equal" if okp else "DIFFER", nd, b[2], b[3]))
            if not okp or any(nd): bad.append((name, mode))
    return bad, {name: r["1"][name][2] for name in selected_cases()}

if __name__ == "__main__":
    if len(sys.argv) == 4: run_child(sys.argv[1], sys.argv[3])
    else:
        bad, stats = compare(sys.argv[1] if len(sys.argv) > 1 else "emu")
        print("FAIL: %s" % bad if bad else "split path == one-kernel path on every case")
        sys.exit(1 if bad else 0)
