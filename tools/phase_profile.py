#!/usr/bin/env python3
"""tools/phase_profile.py — build the -DOA_PHASE_TIMERS variant of the library and print the shader-clock share of every
encoder phase (lane 0 of every wave) on the config-2 workload.  Profiling aid only; the product library has no timers."""
import ctypes, os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
PHASES = ["dc_reject+prologue", "preemphasis", "tone_detect", "transient", "prefilter(pitch+comb)", "mdct+bandE",
          "tvbr/patch", "normalise+dynalloc", "tf_analysis", "coarse+tf_encode", "spread+dynalloc_sig+stereo+trim", "vbr+alloc+fine", "pvq", "finalise", "store", "state load + analysis + call decisions", "  mdct: fold + pre-rotation", "  mdct: FFT", "  mdct: post-rotation", "  mdct: (whole transform, per channel)",
          "  mdct: band energies", "  mdct: spectrum store", "  alloc: clt_compute_allocation", "  alloc: fine energy", "  (unused)",
          "  pf: pitch_downsample", "  pf: pitch_search", "  pf: remove_doubling", "  pf: before/comb/after", "  pf: history store", "  (pf nested)", "  analysis: decimator + window", "  analysis: FFT + bins", "  analysis: bands + network",
          "    an: downmix load", "    an: decimator chains", "    an: decimator out + energy", "    an: silence check + bookkeeping", "    an: window + bit reversal", "    an: inmem move", "    an: FFT", "    an: bins (phases)",
          "    an: bin store + smoothing", "    an: bands (18 lanes)", "    an: distances + cepstrum", "    an: lane-0 chain (features, bandwidth)", "    an: leak boost + network + info", "    an: get_info ring load", "    an: get_info lane-0", "    (unused)"]
def main():
    so = os.path.join(ROOT, "opus_amd/libopus_amd_prof.so")
    srcs = [os.path.join(ROOT, "opus_amd/csrc", f) for f in os.listdir(os.path.join(ROOT, "opus_amd/csrc"))]
    if not (os.path.exists(so) and (os.environ.get("OPUS_AMD_PROF_PREBUILT") == "1" or os.path.getmtime(so) >= max(os.path.getmtime(f) for f in srcs))):      # (prebuilt in the container: the GPU box's minutes are for measuring)
      subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wl,-Bsymbolic", "-DOA_PHASE_TIMERS",
                           "-I" + os.path.join(ROOT, "opus_amd/csrc"), "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "opus_amd/csrc/opus_amd.hip"), "-o", so])
    import opus_amd, signals
    opus_amd.LIB_PATH = so
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    b = opus_amd.EncoderBatch(S, channels=2)
    b.ctl(opus_amd.OPUS_SET_BITRATE_REQUEST, 128000); b.ctl(opus_amd.OPUS_SET_COMPLEXITY_REQUEST, 10)
    sig = [signals.music(8, seed=s) if s % 4 else signals.noise_bursts(8, seed=s) for s in range(64)]
    L = opus_amd.lib()
    ticks = (ctypes.c_ulonglong * 50)()
    for i in range(8):
        pcm = np.stack([sig[s % 64][i * 960:(i + 1) * 960].reshape(-1) for s in range(S)])
        if i == 3: L.opusgpu_debug_phase_ticks(ticks, 1)
        b.encode(pcm, 960)
    L.opusgpu_debug_phase_ticks(ticks, 0)
    t = np.array(list(ticks)[:50], dtype=np.float64)
    tot = t[:16].sum()
    print("phase shares over %d frames (shader clock ticks per frame: %.0f)" % (5 * S, tot / (5 * S)))
    for n, v in zip(PHASES, t): print("  %-34s %6.2f %%  %9.0f ticks/frame" % (n, 100 * v / tot, v / (5 * S)))
if __name__ == "__main__": main()
