#!/usr/bin/env python3
"""tools/phase_profile_p4.py — the -DOA_PHASE_TIMERS build on the config-2 workload through the CELT kernel pipeline: shader-clock share of every section of the
four-streams-per-wave PVQ kernel (celt_enc_pvq4.h).  ticks = wave time in the section (first active lane), lanes = ticks x active lanes / 64 = how full the wave was.
Profiling aid only; the product library has no timers.   usage: phase_profile_p4.py [streams] [same]   (same: every stream gets the same signal -- four aligned groups)"""
import ctypes, os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
NAMES = {1: "leaf: entry (reg path: load)", 2: "leaf: LDS-path rotation fwd", 3: "leaf: reg-path rotation set-up", 0: "band begin (rows, budget, folding masks)", 16: "leaf: exp_rotation fwd", 17: "leaf: pulse search", 18: "leaf: mask + icwrs + ec_enc_uint", 19: "leaf: resynthesis (+ rotation back)",
         20: "compute_theta (partition splits)", 31: "compute_theta (stereo, band level)", 21: "trial set-up: staging, RDO switch / decision", 22: "quant_band pre / post (haar, hadamard, lowband out)",
         23: "stereo_merge", 24: "(tree total, nested)", 25: "tree: way down incl. theta (nested)", 26: "tree: leaf budget (bits2pulses ..)", 30: "tree: leaf incl. alg_quant (nested)", 27: "tree: way up"}
def main():
    so = os.path.join(ROOT, "opus_amd/libopus_amd_prof.so")
    srcs = [os.path.join(ROOT, "opus_amd/csrc", f) for f in os.listdir(os.path.join(ROOT, "opus_amd/csrc"))]
    if not (os.path.exists(so) and (os.environ.get("OPUS_AMD_PROF_PREBUILT") == "1" or os.path.getmtime(so) >= max(os.path.getmtime(f) for f in srcs))):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wl,-Bsymbolic", "-DOA_PHASE_TIMERS",
                               "-I" + os.path.join(ROOT, "opus_amd/csrc"), "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "opus_amd/csrc/opus_amd.hip"), "-o", so])
    if os.environ.get("OPUS_AMD_PROF_BUILD_ONLY") == "1": return
    import opus_amd, signals
    opus_amd.LIB_PATH = so
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    same = len(sys.argv) > 2 and sys.argv[2] == "same"
    b = opus_amd.EncoderBatch(S, channels=2)
    b.ctl(opus_amd.OPUS_SET_BITRATE_REQUEST, 128000); b.ctl(opus_amd.OPUS_SET_COMPLEXITY_REQUEST, 10); b.ctl(11902, 1)
    sig = [signals.music(8, seed=(0 if same else s)) if (same or s % 4) else signals.noise_bursts(8, seed=s) for s in range(64)]
    L = opus_amd.lib()
    ticks = (ctypes.c_ulonglong * 32)(); lanes = (ctypes.c_ulonglong * 32)()
    for i in range(8):
        pcm = np.stack([sig[s % 64][i * 960:(i + 1) * 960].reshape(-1) for s in range(S)])
        if i == 3: L.opusgpu_debug_p4_ticks(ticks, lanes, 1)
        b.encode(pcm, 960)
    L.opusgpu_debug_p4_ticks(ticks, lanes, 0)
    t = np.array(list(ticks), dtype=np.float64); l = np.array(list(lanes), dtype=np.float64)
    frames = 5 * S
    excl = [0, 1, 2, 3, 16, 17, 18, 19, 20, 31, 21, 22, 23, 26, 27]
    print("PVQ kernel sections over %d frames (%s signals); ticks per FRAME = wave ticks / 4 streams" % (frames, "identical" if same else "mixed"))
    for k in sorted(NAMES, key=lambda k: (k not in excl, k)):
        if t[k] == 0: continue
        print("  %-52s %9.0f ticks/frame   lanes %4.1f / 64" % (NAMES[k], t[k] / frames, 64 * l[k] / t[k]))
    way_down_excl = t[25] - t[20]; leaf_excl = t[30] - t[16] - t[17] - t[18] - t[19]
    print("  %-52s %9.0f ticks/frame" % ("tree: way down without theta", way_down_excl / frames))
    print("  %-52s %9.0f ticks/frame" % ("tree: leaf without budget / alg_quant (fill, fold)", leaf_excl / frames))
    tot = sum(t[k] for k in [0, 21, 22, 23, 31]) + t[25] + t[26] + t[30] + t[27]
    print("  %-52s %9.0f ticks/frame" % ("sum of the sections", tot / frames))
if __name__ == "__main__": main()
