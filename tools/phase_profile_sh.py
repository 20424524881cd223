#!/usr/bin/env python3
"""tools/phase_profile_sh.py — build the -DOA_PHASE_TIMERS variant and print the shader-clock share of every stage of the SILK-capable encoder kernel
(lane 0 of every wave) on the config-3 workload.  Profiling aid only; the product library has no timers."""
import ctypes, os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
PH = ["load state", "silence+decide+highpass", "silk_Encode front (control, resample, VAD)", "find_pitch_lags", "noise_shape_analysis", "find_pred_coefs", "process_gains", "NSQ", "encode indices+pulses",
      "silk_Encode tail", "finalise+store", "  pred: LTP corr + VQ + analysis filter", "  pred: Burg x2 + A2NLSF(2nd half)", "  pred: NLSF interpolation search", "  pred: final A2NLSF", "  pred: NLSF quantiser + NLSF2A", "CELT layer of a hybrid frame (+ store)",
      "  shape: sparseness + control", "  shape: window + (warped) autocorrelation", "  shape: schur64 + k2a", "  pitch: window + autocorr + schur + k2a", "  pitch: LPC analysis filter", "  shape: lane-0 gain / bwexpand / limit_warped_coefs", "tonality analysis (own clock, not in the total)"]
def main():
    so = os.path.join(ROOT, "opus_amd/libopus_amd_prof.so")
    hd = os.path.join(ROOT, "opus_amd/csrc")
    if not os.path.exists(so) or (os.environ.get("OPUS_AMD_PROF_PREBUILT") != "1" and os.path.getmtime(so) < max(os.path.getmtime(os.path.join(hd, f)) for f in os.listdir(hd))):
      subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wl,-Bsymbolic", "-DOA_PHASE_TIMERS",
                           "-I" + os.path.join(ROOT, "opus_amd/csrc"), "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "opus_amd/csrc/opus_amd.hip"), "-o", so])
    import opus_amd
    from silk_enc_bench import speech
    opus_amd.LIB_PATH = so
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    cx = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    hyb = len(sys.argv) > 3 and sys.argv[3] == "hybrid"
    Fs, ch, n = (48000, 2, 960) if hyb else (16000, 1, 320)
    b = opus_amd.EncoderBatch(S, channels=ch, application=2049 if hyb else 2048, Fs=Fs)
    for req, v in (((11002, 1001), (4008, 1105), (4002, 128000), (4010, cx)) if hyb else ((11002, 1000), (4008, 1103), (4002, 24000), (4010, cx))): b.ctl(req, v)
    sig = [speech(Fs, 8 * n, 100 + s) if ch == 1 else np.stack([speech(Fs, 8 * n, 100 + s), speech(Fs, 8 * n, 900 + s)], 1).reshape(-1) for s in range(64)]
    L = opus_amd.lib()
    ticks = (ctypes.c_ulonglong * 24)()
    for i in range(8):
        pcm = np.stack([sig[s % 64][i * n * ch:(i + 1) * n * ch] for s in range(S)])
        if i == 3: L.opusgpu_debug_sh_phase_ticks(ticks, 1)
        b.encode(pcm, n)
    L.opusgpu_debug_sh_phase_ticks(ticks, 0)
    t = np.array(list(ticks)[:24], dtype=np.float64)
    t[5] += t[11:16].sum(); tot = t[:11].sum() + t[16]     # the sub-marks consume find_pred_coefs' clock: give the total back to the parent row
    print("oa_sh_encode_kernel, complexity %d: stage shares over %d frames (shader clock ticks per frame: %.0f)" % (cx, 5 * S, tot / (5 * S)))
    for n, v in zip(PH, t): print("  %-44s %6.2f %%  %9.0f ticks/frame" % (n, 100 * v / tot, v / (5 * S)))
if __name__ == "__main__": main()
