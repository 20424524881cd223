#!/usr/bin/env python3
"""tools/phase_profile_sdec.py — the -DOA_PHASE_TIMERS build on the decoder legs of configs 3 / 4: shader-clock share of every section of the lane = stream SILK decoder
(silk_dec_lane.h: oa_sdec_lane_packet).  ticks = wave time in the section, lanes = how many of the 64 lanes were in it.  Profiling aid only; the product library has no timers.
usage: phase_profile_sdec.py [config] [streams]"""
import ctypes, os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
NAMES = {0: "stream record -> lane (load)", 1: "frame head / flags / stereo / glue code", 2: "decode_indices", 3: "decode_pulses", 4: "decode_parameters", 5: "decode_core (excitation, LTP, LPC synthesis)",
         6: "frame back (history, PLC / CNG update)", 7: "MS->LR, resampler, PCM out", 8: "commit (lane -> stream record)"}
def main():
    so = os.path.join(ROOT, "opus_amd/libopus_amd_prof.so")
    if os.environ.get("OPUS_AMD_PROF_PREBUILT") != "1":
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wl,-Bsymbolic", "-DOA_PHASE_TIMERS",
                               "-I" + os.path.join(ROOT, "opus_amd/csrc"), "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "opus_amd/csrc/opus_amd.hip"), "-o", so])
    if os.environ.get("OPUS_AMD_PROF_BUILD_ONLY") == "1": return
    os.environ.setdefault("OPUS_AMD_SDEC_TILE", "64")              # full waves whatever the number of streams: ticks per 64-stream wave
    import opus_amd, signals
    opus_amd.LIB_PATH = so
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    Fs, ch, app, ctl = {3: (16000, 1, 2048, {11002: 1000, 4008: 1103, 4002: 24000, 4010: 10}), 4: (48000, 2, 2049, {11002: 1001, 4008: 1105, 4006: 1, 4002: 128000, 4010: 10})}[cfg]
    n = Fs // 50
    e = opus_amd.EncoderBatch(S, channels=ch, Fs=Fs, application=app)
    for k, v in ctl.items(): e.ctl(k, v)
    d = opus_amd.DecoderBatch(S, channels=ch, Fs=Fs)
    sig = [signals.music(10, seed=s) if s % 4 else signals.noise_bursts(10, seed=s) for s in range(64)]
    def pcm_of(i):
        x = np.stack([sig[s % 64][i * 960:(i + 1) * 960] for s in range(S)])          # (S, 960, 2) at 48 kHz
        if Fs == 16000: x = x[:, ::3]
        return np.ascontiguousarray(x[:, :, :ch]).reshape(S, -1)
    L = opus_amd.lib(); ticks = (ctypes.c_ulonglong * 32)(); lanes = (ctypes.c_ulonglong * 32)()
    frames = 0
    for i in range(8):
        pk = e.encode(pcm_of(i), n)[0]
        if i == 3: L.opusgpu_debug_p4_ticks(ticks, lanes, 1)
        d.decode(pk, n)
        if i >= 3: frames += d.lane_stats()[0]
    L.opusgpu_debug_p4_ticks(ticks, lanes, 0)
    t = np.array(list(ticks), dtype=np.float64); l = np.array(list(lanes), dtype=np.float64)
    print("lane = stream SILK decoder, config %d decode, %d packets through the lane kernel; ticks per 64-stream WAVE and packet" % (cfg, frames))
    tot = sum(t[k] for k in NAMES)
    for k in NAMES:
        if t[k]: print("  %-52s %9.0f ticks  %5.1f %%   lanes %4.1f / 64" % (NAMES[k], t[k] / (frames / 64.0), 100 * t[k] / tot, 64 * l[k] / t[k]))
    print("  %-52s %9.0f ticks" % ("sum", tot / (frames / 64.0)))
if __name__ == "__main__": main()
