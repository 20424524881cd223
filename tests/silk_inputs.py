"""tests/silk_inputs.py — synthetic but plausible inputs for the SILK noise-shaping quantisers (TEST INFRASTRUCTURE).

The value ranges follow what the reference's analysis produces (silk/fixed/noise_shape_analysis_FIX.c, process_gains_FIX.c,
find_pitch_lags_FIX.c): stable LPC/shaping filters, pitch lags in 2..18 ms, gains 2^16..2^24, Lambda 0.5..3.
"""
import numpy as np

NSQ_STATE = np.dtype([("xq", "<i2", 640), ("sLTP_shp_Q14", "<i4", 640), ("sLPC_Q14", "<i4", 96), ("sAR2_Q14", "<i4", 24),
                      ("sLF_AR_shp_Q14", "<i4"), ("sDiff_shp_Q14", "<i4"), ("lagPrev", "<i4"), ("sLTP_buf_idx", "<i4"),
                      ("sLTP_shp_buf_idx", "<i4"), ("rand_seed", "<i4"), ("prev_gain_Q16", "<i4"), ("rewhite_flag", "<i4")], align=True)
NSQ_FRAME = np.dtype([("signalType", "i1"), ("quantOffsetType", "i1"), ("NLSFInterpCoef_Q2", "i1"), ("Seed", "i1"),
                      ("PredCoef_Q12", "<i2", 32), ("LTPCoef_Q14", "<i2", 20), ("AR_Q13", "<i2", 96),
                      ("HarmShapeGain_Q14", "<i4", 4), ("Tilt_Q14", "<i4", 4), ("LF_shp_Q14", "<i4", 4), ("Gains_Q16", "<i4", 4),
                      ("pitchL", "<i4", 4), ("Lambda_Q10", "<i4"), ("LTP_scale_Q14", "<i4")], align=True)
assert NSQ_STATE.itemsize == 4352 and NSQ_FRAME.itemsize == 388

def make_cfg(fs_kHz=16, nb_subfr=4, shaping=24, states=4, warping=True):
    return np.array([fs_kHz, nb_subfr, 16 if fs_kHz == 16 else 10, shaping, states,
                     int(fs_kHz * 0.015 * 65536) if warping else 0], dtype=np.int32)

def fresh_state(n=1):
    st = np.zeros(n, dtype=NSQ_STATE)
    st["prev_gain_Q16"] = 65536          # silk/init_encoder.c / control_codec.c reset values
    st["lagPrev"] = 100
    return st

def _stable(rng, order, scale, q):
    a = rng.standard_normal(order) * (0.7 ** np.arange(order))
    a *= scale / max(1e-9, np.abs(a).sum())
    return np.round(a * (1 << q)).astype(np.int16)

def make_frame(rng, cfg, voiced=None, interp=None, max_lag_ms=18):
    fs, nb, P, S = int(cfg[0]), int(cfg[1]), int(cfg[2]), int(cfg[3])
    f = np.zeros(1, dtype=NSQ_FRAME)[0]
    voiced = bool(rng.integers(0, 2)) if voiced is None else voiced
    f["signalType"] = 2 if voiced else int(rng.integers(0, 2))
    f["quantOffsetType"] = int(rng.integers(0, 2))
    interp = int(rng.integers(0, 5)) if interp is None else interp
    f["NLSFInterpCoef_Q2"] = interp if nb == 4 else 4
    f["Seed"] = int(rng.integers(0, 4))
    pc = np.zeros(32, np.int16)
    pc[:P] = _stable(rng, P, 0.9, 12); pc[16:16 + P] = _stable(rng, P, 0.9, 12)
    f["PredCoef_Q12"] = pc
    ar = np.zeros(96, np.int16); ltp = np.zeros(20, np.int16)
    lag0 = int(rng.integers(2 * fs, int(max_lag_ms * fs) + 1))
    for k in range(nb):
        ar[k * 24:k * 24 + S] = _stable(rng, S, 0.8, 13)
        if voiced:
            ltp[k * 5:k * 5 + 5] = np.round(np.array([0.05, 0.2, 0.45, 0.2, 0.05]) * rng.uniform(0.3, 1.0) * 16384 + rng.integers(-300, 300, 5)).astype(np.int16)
        f["HarmShapeGain_Q14"][k] = int(rng.integers(0, 6000)) if voiced else 0
        f["Tilt_Q14"][k] = -int(rng.integers(0, 5000))
        b = int(rng.integers(200, 1200))
        f["LF_shp_Q14"][k] = np.int32(np.uint32(((16384 - b - int(rng.integers(0, 300))) << 16) | ((b - 16384) & 0xFFFF)))
        f["Gains_Q16"][k] = int(2 ** rng.uniform(16, 24))
        f["pitchL"][k] = int(np.clip(lag0 + rng.integers(-3, 4), 2 * fs, 18 * fs)) if voiced else 0
    if rng.integers(0, 4) == 0:
        f["Gains_Q16"][1:] = f["Gains_Q16"][0]            # exercise the "gain unchanged" path
    f["AR_Q13"] = ar; f["LTPCoef_Q14"] = ltp
    f["Lambda_Q10"] = int(rng.integers(400, 3100))
    f["LTP_scale_Q14"] = int(rng.choice([15565, 12288, 8192]))
    return f

def make_input(rng, cfg, gains):
    """int16 input at roughly the level the subframe gains imply (so pulses are not all zero / all clipped)."""
    fs, nb = int(cfg[0]), int(cfg[1]); L = 5 * fs
    x = np.zeros(nb * L, np.int16)
    for k in range(nb):
        amp = min(30000.0, gains[k] / 65536.0 * rng.uniform(0.5, 6.0))
        x[k * L:(k + 1) * L] = np.clip(np.round(rng.standard_normal(L) * amp), -32768, 32767).astype(np.int16)
    return x

def make_pitch_frame(rng, fs_kHz, nb_subfr, kind=None):
    """an LPC-residual-like analysis buffer of (20 + 5*nb_subfr) ms: a jittered pulse train + noise (voiced), noise (unvoiced), or silence"""
    n = (20 + 5 * nb_subfr) * fs_kHz
    kind = kind or rng.choice(["voiced", "voiced", "voiced", "weak", "noise", "loud", "silence", "tiny"])
    if kind == "silence": return np.zeros(n, np.int16), kind
    x = rng.standard_normal(n) * 200.0
    if kind in ("voiced", "weak", "loud", "tiny"):
        lag = rng.uniform(2.2, 17.5) * fs_kHz; drift = rng.uniform(-0.004, 0.004)
        t = rng.uniform(0, lag); amp = 3000.0 if kind != "weak" else 500.0
        shape = rng.standard_normal(6) * np.array([1, .7, .5, .3, .2, .1])
        while t < n - 8:
            i = int(t); x[i:i + 6] += amp * shape * rng.uniform(0.8, 1.2)
            lag *= (1 + drift); t += lag
    if kind == "loud": x *= 12.0
    if kind == "tiny": x *= 0.01
    return np.clip(np.round(x), -32768, 32767).astype(np.int16), kind
