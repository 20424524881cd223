#!/usr/bin/env python3
"""tools/dec_fast_check.py — the decoder's steady-state kernels (oa_decode_fast_kernel: CELT-only; oa_sdec_lane_kernel: SILK-only and the SILK layer of hybrid packets, one lane
per stream; oa_decode_hyb_kernel: the hybrid packets' CELT layer) in front of the general kernel against the general kernel alone: the same packet
sequences through OPUS_AMD_DEC_FAST = 0 and 1 (one subprocess each: the switch is read once per process) must give identical PCM, sample counts, final ranges AND identical
stream records, byte for byte.  The sequences mix what the fast kernel takes (CELT-only, one frame) with what it hands over (SILK / hybrid packets, multi-frame packets, lost
packets and the frames after them), so that streams move between the kernels.  usage: dec_fast_check.py [emu|gpu]      (child: dec_fast_check.py <lib> <out.pkl>)"""
import os, sys, ctypes, pickle, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)

CASES = {   # name: (Fs, channels, application, encoder ctls, frame ms, frames, loss pattern period (0 = none))
    "celt_stereo":   (48000, 2, 2051, {4002: 96000, 4010: 5}, 20, 14, 5),
    "celt_mono_10":  (48000, 1, 2051, {4002: 48000, 4010: 5}, 10, 16, 0),
    "celt_24k":      (24000, 2, 2051, {4002: 64000, 4010: 5}, 20, 10, 4),
    "audio_auto":    (48000, 2, 2049, {4002: 32000, 4010: 5}, 20, 16, 6),          # SILK / hybrid / CELT packets as the encoder decides
    "voip_16k":      (16000, 1, 2048, {4002: 20000, 4010: 5}, 20, 10, 0),
    "celt_40ms":     (48000, 2, 2051, {4002: 96000, 4010: 5}, 40, 6, 0),            # multi-frame packets: general kernel
    "celt_2_5ms":    (48000, 2, 2051, {4002: 128000, 4010: 5}, 2.5, 24, 7),         # LM = 0: one post-filter segment
    "celt_5ms_12k":  (12000, 1, 2051, {4002: 24000, 4010: 5}, 5, 16, 0),            # downsampling by 4 on the way out
    "mono_coded":    (48000, 2, 2051, {4002: 48000, 4010: 5, 4022: 1}, 20, 10, 4),  # mono packets into a stereo decoder: one spectrum, two syntheses
    "stereo_to_mono": (48000, 2, 2051, {4002: 96000, 4010: 5}, 20, 10, 0, 1),       # stereo packets into a mono decoder: the spectra are mixed before the synthesis
    "celt_510k":     (48000, 2, 2051, {4002: 510000, 4010: 5}, 20, 8, 0),           # the highest rate: leaves with more pulses than a 16-entry window of U(n, k) holds (celt_dec_pvq4.h: p4d_cwrsi's table reads on the spot)
    "celt_mono_256k_10": (48000, 1, 2051, {4002: 256000, 4010: 5}, 10, 12, 5),
    # the SILK steady state: oa_sdec_lane_kernel (one lane per stream) takes these after each stream's first packet, and after a loss the general kernel has them back for a packet
    "silk_wb_20":    (16000, 1, 2048, {11002: 1000, 4002: 24000, 4010: 5}, 20, 12, 0),
    "silk_stereo":   (48000, 2, 2048, {11002: 1000, 4004: 1103, 4002: 40000, 4010: 5}, 20, 14, 6),            # mid/side, side frames that come and go, 16 -> 48 kHz on the way out
    "silk_nb_10":    (8000, 1, 2048, {11002: 1000, 4008: 1101, 4002: 12000, 4010: 5}, 10, 16, 0),
    "silk_mb_60":    (12000, 1, 2048, {11002: 1000, 4008: 1102, 4002: 16000, 4010: 5}, 60, 6, 0),  # three SILK frames per packet
    "silk_fec":      (16000, 1, 2048, {11002: 1000, 4002: 28000, 4010: 5, 4012: 1, 4014: 20}, 20, 14, 5),   # packets that carry LBRR data (skipped: no FEC request)
    "silk_24k_st40": (24000, 2, 2048, {11002: 1000, 4004: 1102, 4002: 36000, 4010: 5}, 40, 8, 0),
    "silk_cbr":      (16000, 1, 2048, {11002: 1000, 4002: 20000, 4010: 5, 4006: 0}, 20, 10, 0),    # padded packets (code 3): the general kernel
    "silk_dtx":      (16000, 1, 2048, {11002: 1000, 4002: 20000, 4010: 5, 4016: 1}, 20, 30, 0, 1, "quiet"),
    "silk_mono_in_stereo": (16000, 1, 2048, {11002: 1000, 4002: 20000, 4010: 5}, 20, 8, 0, 2),     # mono packets into a stereo decoder: not the lane kernel's
    "silk_celt_switch": (16000, 1, 2049, {11002: 1000, 4002: 24000, 4010: 5}, 20, 20, 0, 1, "mode"),  # SILK-only <-> CELT-only every five frames: the last SILK packet before a switch carries a redundant CELT frame -- the lane hands it on
    # the hybrid steady state: the SILK layer on the lane kernel, the CELT layer (bands 17..) by oa_decode_hyb_kernel
    "hyb_stereo":    (48000, 2, 2049, {11002: 1001, 4002: 48000, 4010: 5}, 20, 14, 6),
    "hyb_mono_10":   (48000, 1, 2048, {11002: 1001, 4004: 1104, 4002: 32000, 4010: 5}, 10, 16, 0),           # super-wideband: bands 17..18
    "hyb_24k_out":   (48000, 2, 2049, {11002: 1001, 4002: 64000, 4010: 5}, 20, 10, 0, 2, "", 24000),         # decoded at 24 kHz: the SILK layer resampled 16 -> 24, the CELT layer decimated
    "hyb_celt_switch": (48000, 2, 2049, {11002: 1001, 4002: 48000, 4010: 5}, 20, 20, 0, 2, "hmode"),           # hybrid <-> CELT-only: redundant frames, transitions
    "silk_down_12k": (16000, 1, 2048, {11002: 1000, 4002: 24000, 4010: 5}, 20, 12, 5, 1, "", 12000),          # decoded below the internal rate: the resampler's AR2 + FIR path on the lanes
    "silk_st_down_8k": (16000, 2, 2048, {11002: 1000, 4002: 36000, 4010: 5}, 40, 8, 0, 2, "", 8000),
    "silk_nb_up_24k": (8000, 1, 2048, {11002: 1000, 4002: 12000, 4010: 5}, 60, 6, 0, 1, "", 24000),
    "hyb_cbr_mono_in_stereo": (48000, 1, 2049, {11002: 1001, 4002: 32000, 4010: 5, 4006: 0}, 20, 10, 0, 2),   # padded hybrid packets (code 3, one frame), mono into a stereo decoder
    "silk_bw_switch": (16000, 1, 2048, {11002: 1000, 4002: 20000, 4010: 5}, 20, 16, 0, 1, "bw"),   # the bandwidth changes in mid-stream: the general kernel re-initialises, then the lanes again
}
if os.environ.get("DEC_FAST_CASES"): CASES = {k: v for k, v in CASES.items() if any(x in k for x in os.environ["DEC_FAST_CASES"].split(","))}
def dec_channels(name): return CASES[name][7] if len(CASES[name]) > 7 else CASES[name][1]

def make_packets(name):
    from reflib import ref_fx
    L = ref_fx(); Fs, ch, app, ctl, ms, frames, loss = CASES[name][:7]
    L.opus_encoder_create.restype = ctypes.c_void_p; L.opus_encoder_create.argtypes = [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int)]
    L.opus_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    n = int(Fs * ms // 1000); S = 5; out = (ctypes.c_ubyte * 4000)(); err = ctypes.c_int(); seqs = []
    for s in range(S):
        e = L.opus_encoder_create(Fs, ch, app, ctypes.byref(err))
        for k, v in ctl.items(): L.opus_encoder_ctl(e, k, v)
        if name == "audio_auto" and s % 2: L.opus_encoder_ctl(e, 4002, 96000)          # half the streams at a rate where the encoder picks CELT
        rng = np.random.default_rng(100 * s + len(name)); t = np.arange(n * frames) / Fs
        x = 6000 * np.sin(2 * np.pi * (180 + 40 * s) * t)[:, None] * (np.sin(2 * np.pi * 1.3 * t + s) > -0.2)[:, None] + rng.normal(0, 300, (n * frames, ch))
        kind = CASES[name][8] if len(CASES[name]) > 8 else ""
        if kind == "quiet": x[n * 6:n * 22] = x[n * 6:n * 22] * 0.0005                    # a long silence: DTX frames (one-byte packets)
        x = np.clip(x, -32768, 32767).astype(np.int16)
        pk = []
        for f in range(frames):
            if kind == "hmode" and f and f % 5 == 0: L.opus_encoder_ctl(e, 11002, 1002 if (f // 5) % 2 else 1001)
            if kind == "mode" and f and f % 5 == 0: L.opus_encoder_ctl(e, 11002, 1002 if (f // 5) % 2 else 1000)
            if kind == "bw" and f in (5, 10): L.opus_encoder_ctl(e, 4008, 1101 if f == 5 else 1103)
            k = L.opus_encode(e, np.ascontiguousarray(x[f * n:(f + 1) * n]).ctypes.data, n, out, 1500); assert k > 0
            pk.append(b"" if loss and (f + s) % loss == loss - 1 else bytes(out[:k]))
        seqs.append(pk)
    return seqs

def run_child(libpath, outp):
    import opus_amd
    opus_amd.LIB_PATH = libpath
    res = {}
    for name in CASES:
        Fs, ch, app, ctl, ms, frames, loss = CASES[name][:7]
        seqs = make_packets(name); S = len(seqs)
        if len(CASES[name]) > 9: Fs = CASES[name][9]
        n = int(Fs * ms // 1000)
        b = opus_amd.DecoderBatch(S, channels=dec_channels(name), Fs=Fs)
        steps = []; lane = [0, 0]
        for f in range(frames):
            pcm, ns, rng = b.decode([seqs[s][f] for s in range(S)], n)
            steps.append((pcm.copy(), ns.copy(), rng.copy()))
            if hasattr(b, "lane_stats") and os.environ.get("OPUS_AMD_DEC_FAST", "1") != "0":
                t, h = b.lane_stats(); lane[0] += t; lane[1] += h
        res[name] = (steps, [b.export_state(s) for s in range(S)], lane); b.close()
    pickle.dump(res, open(outp, "wb"))

def compare(which="emu", tmpdir="/tmp", verbose=True, with_stats=False):
    if which == "emu":
        import hostemu; lib = hostemu.build_emu_lib()
    else: lib = os.path.join(ROOT, "opus_amd/libopus_amd.so")
    r = {}
    for mode in "01":
        outp = os.path.join(tmpdir, "dec_fast_%s_%s_%d.pkl" % (which, mode, os.getpid()))
        subprocess.check_call([sys.executable, os.path.abspath(__file__), lib, outp], env=dict(os.environ, OPUS_AMD_DEC_FAST=mode))
        r[mode] = pickle.load(open(outp, "rb")); os.unlink(outp)
    bad = []
    for name in CASES:
        a, b = r["0"][name], r["1"][name]
        ok = all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]) for x, y in zip(a[0], b[0]))
        nd = [sum(p != q for p, q in zip(x, y)) for x, y in zip(a[1], b[1])]
        if verbose: print("%-13s output %s, state bytes differing per stream %s; lane kernel took %d packets, handed on %d" % (name, "equal" if ok else "DIFFERS", nd, b[2][0], b[2][1]))
        if not ok or any(nd): bad.append(name)
    return (bad, {name: r["1"][name][2] for name in CASES}) if with_stats else bad

if __name__ == "__main__":
    if len(sys.argv) == 3: run_child(sys.argv[1], sys.argv[2])
    else:
        bad = compare(sys.argv[1] if len(sys.argv) > 1 else "emu")
        print("FAIL: %s" % bad if bad else "fast + general kernels == general kernel alone on every case")
        sys.exit(1 if bad else 0)
