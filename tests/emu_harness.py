"""tests/emu_harness.py — TEST INFRASTRUCTURE: drive the kernel body on the CPU wave emulator and the oracle side by
side, with optional per-stage dumps, to localise the first divergence."""
import ctypes, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMPFN = ctypes.CFUNCTYPE(None, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int)

def load_emu():
    return ctypes.CDLL(os.path.join(ROOT, "tests/emu/libemu_encoder.so"))

CFG_FIELDS = ["channels", "application", "user_bitrate_bps", "use_vbr", "vbr_constraint", "complexity",
              "force_channels", "user_bandwidth", "max_bandwidth", "lsb_depth", "disable_inv", "packet_loss_perc"]

def new_stream(E, channels, bitrate=-1000, complexity=9, application=2051, use_vbr=1, vbr_constraint=1, force_channels=-1000,
               user_bandwidth=-1000, max_bandwidth=1105, lsb_depth=24, disable_inv=0):
    """Host-side image of OaStream after opus_encoder_init (mirrors opus_amd host init)."""
    n = E.emu_sizeof_stream() // 4
    s = np.zeros(n, np.int32)
    cfg = dict(channels=channels, application=application, user_bitrate_bps=bitrate, use_vbr=use_vbr, vbr_constraint=vbr_constraint,
               complexity=complexity, force_channels=force_channels, user_bandwidth=user_bandwidth, max_bandwidth=max_bandwidth,
               lsb_depth=lsb_depth, disable_inv=disable_inv, packet_loss_perc=0)
    for i, k in enumerate(CFG_FIELDS): s[i] = cfg[k]
    st = 16
    # OaEncScalars order: stream_channels, bandwidth, auto_bandwidth, first, prev_mode, hybrid_stereo_width_Q14, hp_mem[4], rangeFinal,
    # rng, spread_decision, delayedIntra, tonal_average, lastCodedBands, hf_average, tapset_decision, prefilter_period, prefilter_gain,
    # prefilter_tapset, consec_transient, preemph_memE[2], vbr_reservoir, vbr_drift, vbr_offset, vbr_count, overlap_max, stereo_saving, intensity, spec_avg, pad[4]
    s[st + 0] = channels; s[st + 1] = 1105; s[st + 3] = 1; s[st + 5] = 1 << 14
    s[st + 12] = 2      # spread_decision = SPREAD_NORMAL
    s[st + 13] = 1      # delayedIntra
    s[st + 14] = 256    # tonal_average
    s[st + 34] = -1     # voice_ratio
    arr = st + 36
    s[arr + 42: arr + 42 + 84] = -(28 << 24)   # oldLogE, oldLogE2 = -28.0 (Q24)
    return s

class DumpLog:
    def __init__(self): self.items = []
    def cb(self):
        def f(tag, p, n):
            self.items.append((tag.decode(), ctypes.string_at(p, n)))
        self._f = DUMPFN(f)
        return self._f

def first_diff(a, b):
    """a, b: lists of (tag, bytes). Returns description of first mismatch or None."""
    for i, (x, y) in enumerate(zip(a, b)):
        if x[0] != y[0]: return "tag order differs at %d: %s vs %s" % (i, x[0], y[0])
        if x[1] != y[1]:
            xa = np.frombuffer(x[1], np.int32 if len(x[1]) % 4 == 0 else np.uint8); ya = np.frombuffer(y[1], xa.dtype)
            d = np.nonzero(xa != ya)[0]
            return "%s (#%d): %d/%d words differ, first at %d: %s vs %s" % (x[0], i, len(d), len(xa), d[0], xa[d[0]:d[0] + 4], ya[d[0]:d[0] + 4])
    if len(a) != len(b): return "dump count differs %d vs %d" % (len(a), len(b))
    return None
