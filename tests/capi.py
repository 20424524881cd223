"""tests/capi.py — TEST INFRASTRUCTURE: the classic libopus API (opus_encoder_* / opus_decoder_* / multistream) of ANY library that exports it, through
ctypes: the compiled reference (oracle/_ref/libopus_ref_fx.so), the product on the MI355X (opus_amd/libopus_amd.so) or the product's C ABI on the CPU
wave emulator (tests/emu/libopus_amd_emu.so).  Parity tests are written once against this interface and run on (reference, emulated) here and on
(reference, product) on the GPU box."""
import ctypes, os, functools, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQ = dict(application=4000, bitrate=4002, max_bandwidth=4004, vbr=4006, bandwidth=4008, complexity=4010, inband_fec=4012, packet_loss=4014, dtx=4016,
           vbr_constraint=4020, force_channels=4022, signal=4024, lsb_depth=4036, expert_frame_duration=4040, prediction_disabled=4042, phase_inv_disabled=4046,
           force_mode=11002, gain=4034)

@functools.lru_cache(None)
def load(which):
    if which == "ref":
        from reflib import ref_fx
        return ref_fx()
    if which == "ref_fl":
        from reflib import ref_fl
        return ref_fl()
    if which == "ref_fxa":
        from reflib import ref_fxa
        return ref_fxa()
    if which == "emu":
        import hostemu
        return ctypes.CDLL(hostemu.build_emu_lib(), mode=ctypes.RTLD_LOCAL)
    if which == "gpu":
        return ctypes.CDLL(os.path.join(ROOT, "opus_amd/libopus_amd.so"), mode=ctypes.RTLD_LOCAL)
    raise ValueError(which)

def _proto(L):
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.opus_encoder_create.restype = vp; L.opus_encoder_create.argtypes = [ci, ci, ci, ctypes.POINTER(ci)]
    L.opus_encoder_destroy.argtypes = [vp]; L.opus_encoder_destroy.restype = None
    L.opus_encode.argtypes = [vp, vp, ci, vp, ci]
    L.opus_decoder_create.restype = vp; L.opus_decoder_create.argtypes = [ci, ci, ctypes.POINTER(ci)]
    L.opus_decoder_destroy.argtypes = [vp]; L.opus_decoder_destroy.restype = None
    L.opus_decode.argtypes = [vp, ctypes.c_char_p, ci, vp, ci, ci]
    return L

class Enc:
    def __init__(self, which, Fs, channels, application, **ctl):
        L = self.L = _proto(load(which)); err = ctypes.c_int()
        self.st = L.opus_encoder_create(Fs, channels, application, ctypes.byref(err))
        assert err.value == 0 and self.st, (which, Fs, channels, application, err.value)
        self.ch = channels; self.out = (ctypes.c_ubyte * 8000)()
        for k, v in ctl.items(): assert self.set(k, v) == 0, (k, v)
    def set(self, name, value):
        self.L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        return self.L.opus_encoder_ctl(self.st, REQ[name], int(value))
    def get(self, req):
        v = ctypes.c_int32(); self.L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        r = self.L.opus_encoder_ctl(self.st, req, ctypes.byref(v)); assert r == 0, (req, r)
        return v.value
    def encode(self, pcm, frame, maxb=1276):
        pcm = np.ascontiguousarray(pcm, np.int16)
        n = self.L.opus_encode(self.st, pcm.ctypes.data, frame, self.out, maxb)
        return bytes(self.out[:max(n, 0)]), n, self.get(4031) & 0xffffffff
    def __del__(self):
        try: self.L.opus_encoder_destroy(self.st)
        except Exception: pass

class Dec:
    def __init__(self, which, Fs, channels):
        L = self.L = _proto(load(which)); err = ctypes.c_int()
        self.st = L.opus_decoder_create(Fs, channels, ctypes.byref(err))
        assert err.value == 0 and self.st, (which, Fs, channels, err.value)
        self.ch = channels; self.Fs = Fs
    def get(self, req):
        v = ctypes.c_int32(); self.L.opus_decoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        r = self.L.opus_decoder_ctl(self.st, req, ctypes.byref(v)); assert r == 0, (req, r)
        return v.value
    def set(self, name, value):
        self.L.opus_decoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        return self.L.opus_decoder_ctl(self.st, REQ[name], int(value))
    def decode(self, pkt, max_frame=None, fec=0):
        if max_frame is None: max_frame = self.Fs // 25 * 3
        pcm = np.zeros((max_frame, self.ch), np.int16)
        n = self.L.opus_decode(self.st, pkt if pkt else None, len(pkt) if pkt else 0, pcm.ctypes.data, max_frame, int(fec))
        return n, pcm[:max(n, 0)].copy(), self.get(4031) & 0xffffffff
    def __del__(self):
        try: self.L.opus_decoder_destroy(self.st)
        except Exception: pass
