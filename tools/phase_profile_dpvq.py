#!/usr/bin/env python3
"""tools/phase_profile_dpvq.py — the -DOA_PHASE_TIMERS build on the decoder leg of config 2 through the decoder's kernel pipeline: shader-clock ticks of the sections of
oa_celt_dpvq_kernel (celt_dec_pvq4.h: four streams per wave; ticks per FRAME = wave ticks / 4) with their lane fill, and of the front / back kernels' frame sections (one wave
per stream).  Profiling aid only; the product library has no timers.   usage: phase_profile_dpvq.py [streams]"""
import ctypes, os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
P4 = {0: "band begin (rows, budget, folding masks)", 1: "theta symbol (stereo, band level) + folding source staged", 22: "quant_band pre / post (haar, hadamard, lowband out)", 3: "theta symbol (partition splits)",
      2: "tree: way down (incl. the split thetas)", 4: "tree: leaf budget (bits2pulses ..)", 5: "leaf: ec_dec_uint (the PVQ index)", 6: "leaf: cwrsi (index -> pulses)", 7: "leaf: scale + rotation back + store",
      8: "tree: leaf total (incl. 4 - 7 and the leaves without pulses)", 9: "tree: way up", 20: "stereo merge + band out", 24: "(tree total, nested)"}
FB = {10: "front: header symbols, coarse energy, tf, dynalloc (lane 0)", 11: "front: bit allocation + fine energy + clear X", 13: "back: anti-collapse, energy finalise", 14: "back: denormalise_bands (+ overlap load)",
      15: "back: IMDCT", 16: "back: comb filter + history / overlap store", 17: "back: de-emphasis", 18: "back: PCM out"}
def main():
    so = os.path.join(ROOT, "opus_amd/libopus_amd_prof.so")
    if os.environ.get("OPUS_AMD_PROF_PREBUILT") != "1":
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wl,-Bsymbolic", "-DOA_PHASE_TIMERS",
                               "-I" + os.path.join(ROOT, "opus_amd/csrc"), "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "opus_amd/csrc/opus_amd.hip"), "-o", so])
    if os.environ.get("OPUS_AMD_PROF_BUILD_ONLY") == "1": return
    import opus_amd, signals
    opus_amd.LIB_PATH = so
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    e = opus_amd.EncoderBatch(S, channels=2)
    e.ctl(opus_amd.OPUS_SET_BITRATE_REQUEST, 128000); e.ctl(opus_amd.OPUS_SET_COMPLEXITY_REQUEST, 10)
    d = opus_amd.DecoderBatch(S, channels=2, Fs=48000); d.set_pvq_stage(1)
    sig = [signals.music(10, seed=s) if s % 4 else signals.noise_bursts(10, seed=s) for s in range(64)]
    pks = [e.encode(np.stack([sig[s % 64][i * 960:(i + 1) * 960].reshape(-1) for s in range(S)]), 960)[0] for i in range(8)]      # (the encoder's PVQ kernel shares the counters: all of it first)
    L = opus_amd.lib(); ticks = (ctypes.c_ulonglong * 32)(); lanes = (ctypes.c_ulonglong * 32)()
    took = 0
    for i in range(8):
        if i == 3: L.opusgpu_debug_p4_ticks(ticks, lanes, 1)
        d.decode(pks[i], 960)
        if i >= 3: took += d.pvq_stats()
    L.opusgpu_debug_p4_ticks(ticks, lanes, 0)
    t = np.array(list(ticks), dtype=np.float64); ln = np.array(list(lanes), dtype=np.float64); frames = 5 * S
    print("decoder kernel pipeline, config 2 packets, %d frames (%d through oa_celt_dpvq_kernel)" % (frames, took))
    print("oa_celt_dpvq_kernel: ticks per FRAME = wave ticks / 4 streams")
    for k in P4: print("  %-62s %9.0f ticks   lanes %4.1f / 64" % (P4[k], t[k] / frames, 64 * ln[k] / max(t[k], 1)))
    print("front (oa_decode_fast_kernel) and back (oa_celt_dback_kernel) kernels: ticks per frame, one wave per stream")
    for k in FB: print("  %-62s %9.0f ticks" % (FB[k], t[k] / frames))
if __name__ == "__main__": main()
