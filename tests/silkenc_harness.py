"""tests/silkenc_harness.py — TEST INFRASTRUCTURE: drive the SILK encoder body on the CPU wave emulator and the compiled reference's silk_Encode
(oracle/ref_expose/x_silk_enc.c) side by side, below the Opus layer, with per-stage taps to localise the first divergence."""
import ctypes, os, subprocess, numpy as np
from reflib import ref_expose, ROOT
from emu_harness import DumpLog, first_diff, DUMPFN

def build_emu():
    so = os.path.join(ROOT, "tests/emu/libemu_silk_enc.so")
    srcs = [os.path.join(ROOT, "tests/emu", f) for f in ("emu_silk_enc.cpp", "wave_emu.cpp")]
    hd = os.path.join(ROOT, "opus_amd/csrc")
    hdrs = [os.path.join(hd, f) for f in os.listdir(hd) if f.endswith(".h")] + [os.path.join(ROOT, "tests/emu/wave_emu.h")]
    import fcntl
    with open(so + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in srcs + hdrs):
            subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-rdynamic", "-I" + os.path.join(ROOT, "tests/emu"), "-I" + hd] + srcs + ["-o", so + ".tmp"])
            os.replace(so + ".tmp", so)
    return ctypes.CDLL(so)

CTL = ["nChannelsAPI", "nChannelsInternal", "API_sampleRate", "maxInternalSampleRate", "minInternalSampleRate", "desiredInternalSampleRate", "payloadSize_ms", "bitRate",
       "packetLossPercentage", "complexity", "useInBandFEC", "LBRR_coded", "useDTX", "useCBR", "maxBits", "toMono", "opusCanSwitch", "reducedDependency",
       "internalSampleRate", "allowBandwidthSwitch", "inWBmodeWithoutVariableLP", "stereoWidth_Q14", "switchReady", "signalType", "offset"]
def make_ctl(**kw):
    d = dict(nChannelsAPI=1, nChannelsInternal=1, API_sampleRate=16000, maxInternalSampleRate=16000, minInternalSampleRate=8000, desiredInternalSampleRate=16000, payloadSize_ms=20,
             bitRate=24000, packetLossPercentage=0, complexity=10, useInBandFEC=0, LBRR_coded=0, useDTX=0, useCBR=0, maxBits=1275 * 8, toMono=0, opusCanSwitch=0, reducedDependency=0)
    d.update(kw)
    c = np.zeros(len(CTL), np.int32)
    for i, k in enumerate(CTL): c[i] = d.get(k, 0)
    return c
def P(a): return a.ctypes.data_as(ctypes.c_void_p)

class Pair:
    """one reference encoder and one emulated encoder fed identically"""
    def __init__(self, channels=1):
        self.R = ref_expose(); self.E = build_emu()
        self.rst = np.zeros(self.R.refx_silk_enc_size() + 64, np.uint8); self.R.refx_silk_enc_init(P(self.rst), channels)
        self.est = np.zeros(self.E.emu_silk_enc_size() + 64, np.uint8); self.E.emu_silk_enc_init(P(self.est))
    def step(self, ctl, pcm, activity=-1, cap=1275, taps=True):
        nS = len(pcm) // int(ctl[0])
        out = []
        logs = []
        for which in (0, 1):
            c = ctl.copy(); o = np.zeros(cap + 8, np.uint8); res = np.zeros(4, np.int32); log = DumpLog(); cb = log.cb()
            if which == 0:
                self.R.refx_silk_set_dump(cb if taps else DUMPFN(0))
                ret = self.R.refx_silk_encode(P(self.rst), P(c), P(pcm), nS, P(o), cap, P(res), activity)
                self.R.refx_silk_set_dump(DUMPFN(0))
            else:
                self.E.emu_set_dump(cb if taps else DUMPFN(0))
                ret = self.E.emu_silk_encode(P(self.est), P(c), P(pcm), nS, P(o), cap, P(res), activity)
                self.E.emu_set_dump(DUMPFN(0))
            out.append((ret, c, o, res)); logs.append(log.items)
        return out, logs

def compare(out, logs):
    """None when the two encoders agree on return code, byte count, range-coder state, payload bytes and control read-back; otherwise a description that
    starts with the first diverging stage tap (unused words of the control block are stack garbage in the reference, so taps are only consulted on failure)"""
    (r0, c0, o0, s0), (r1, c1, o1, s1) = out
    n = (int(s0[1]) + 7) >> 3
    why = None
    if r0 != r1: why = "ret %d vs %d" % (r0, r1)
    elif not np.array_equal(s0, s1): why = "res %s vs %s" % (s0, s1)
    elif not np.array_equal(o0[:n], o1[:n]): why = "bytes differ (first at %d of %d)" % (int(np.nonzero(o0[:n] != o1[:n])[0][0]), n)
    elif not np.array_equal(c0, c1): why = "ctl %s vs %s" % (c0, c1)
    if why is None: return None
    d = first_diff(logs[0], logs[1])
    return why + (" | first tap difference: " + d if d else "")
