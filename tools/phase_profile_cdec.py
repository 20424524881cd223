#!/usr/bin/env python3
"""tools/phase_profile_cdec.py — the -DOA_PHASE_TIMERS build on the decoder leg of config 2: shader-clock share of the sections of the CELT-only fast kernel's frame function
(celt_dec_frame.h: celt_decode_frame_wave<true>).  Profiling aid only; the product library has no timers.   usage: phase_profile_cdec.py [streams]"""
import ctypes, os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
NAMES = {10: "header symbols, coarse energy, tf, dynalloc (lane 0)", 11: "bit allocation + fine energy + clear X", 12: "quant_all_bands (PVQ decode, band tree)", 13: "anti-collapse, energy finalise", 
         14: "denormalise_bands (+ overlap load)", 15: "IMDCT", 16: "comb filter + history / overlap store", 17: "de-emphasis (lane 0)", 18: "PCM out"}
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
    d = opus_amd.DecoderBatch(S, channels=2, Fs=48000)
    sig = [signals.music(10, seed=s) if s % 4 else signals.noise_bursts(10, seed=s) for s in range(64)]
    L = opus_amd.lib(); ticks = (ctypes.c_ulonglong * 32)(); lanes = (ctypes.c_ulonglong * 32)()
    for i in range(8):
        pk = e.encode(np.stack([sig[s % 64][i * 960:(i + 1) * 960].reshape(-1) for s in range(S)]), 960)[0]
        if i == 3: L.opusgpu_debug_p4_ticks(ticks, lanes, 1)
        d.decode(pk, 960)
    L.opusgpu_debug_p4_ticks(ticks, lanes, 0)
    t = np.array(list(ticks), dtype=np.float64); frames = 5 * S
    tot = sum(t[k] for k in NAMES)
    print("CELT-only fast decoder kernel, config 2 packets, %d frames; ticks per frame (one wave per stream)" % frames)
    for k in NAMES: print("  %-58s %9.0f ticks  %5.1f %%" % (NAMES[k], t[k] / frames, 100 * t[k] / tot))
    print("  %-58s %9.0f ticks" % ("sum", tot / frames))
if __name__ == "__main__": main()
