"""Every ctl request number of the libopus surface on every object type, legal and illegal arguments, against the compiled reference (libopus_ref_fxa.so: the fixed-point
build with the float API): return code of every call, then the read-back of every GET.  The eleventh fuzzer of the suite -- the others compare packets and PCM, this one the
control surface (src/opus_encoder.c:2754-3300, src/opus_decoder.c:1083-1250, src/opus_multistream_encoder.c:1143-1330, src/opus_multistream_decoder.c:430-540,
src/opus_projection_encoder.c:440-480, src/opus_projection_decoder.c:240-260).  Here on the wave emulator (ctl calls never launch a kernel: the same host code as the product)."""
import ctypes, numpy as np, pytest
import capi
from reflib import ref_fxa
pytestmark = pytest.mark.skipif(ref_fxa() is None, reason="oracle/_ref not built")
WHICH = "emu"
vp, ci = ctypes.c_void_p, ctypes.c_int

REQS = list(range(4000, 4064)) + [5120, 5121, 5122, 6001, 6002, 6003, 6004, 6005] + list(range(10000, 10030)) + [11002, 11003, 11018, 11019]
PTR_EVEN = {10022, 10026, 10028}                                        # CELT_SET_ANALYSIS, OPUS_SET_ENERGY_MASK, CELT_SET_SILK_INFO take a pointer
SKIP = {4052, 4053}                                                     # OPUS_SET / GET_DNN_BLOB (pointer + length; builds with the DNN tools only)
NO_VALUE = {10015, 5120, 5122}                                          # these hand back addresses
VALUES = [-1000, -1, 0, 1, 2, 3, 5, 8, 10, 11, 16, 24, 25, 100, 101, 500, 999, 1000, 1001, 1002, 1003, 1100, 1101, 1102, 1103, 1104, 1105, 1106, 2047, 2048, 2049, 2050, 2051, 2052, 2053, 2054,
          3001, 3002, 3003, 4999, 5000, 5001, 5002, 5003, 5004, 5005, 5006, 5007, 5008, 5009, 5010, 6000, 32767, 32768, -32768, -32769, 64000, 510000, 512000, 750000, 1275 * 8 * 50, 2147483647, -2147483647]

class Obj:
    def __init__(self, L, kind):
        self.L, self.kind = L, kind; err = ci()
        if kind == "enc":
            L.opus_encoder_create.restype = vp; L.opus_encoder_create.argtypes = [ci, ci, ci, ctypes.POINTER(ci)]
            self.st = L.opus_encoder_create(48000, 2, 2049, ctypes.byref(err)); self.fn = L.opus_encoder_ctl
        elif kind == "dec":
            L.opus_decoder_create.restype = vp; L.opus_decoder_create.argtypes = [ci, ci, ctypes.POINTER(ci)]
            self.st = L.opus_decoder_create(48000, 2, ctypes.byref(err)); self.fn = L.opus_decoder_ctl
        elif kind == "msenc":
            L.opus_multistream_encoder_create.restype = vp; L.opus_multistream_encoder_create.argtypes = [ci, ci, ci, ci, ctypes.c_char_p, ci, ctypes.POINTER(ci)]
            self.st = L.opus_multistream_encoder_create(48000, 3, 2, 1, bytes([0, 1, 2]), 2049, ctypes.byref(err)); self.fn = L.opus_multistream_encoder_ctl
        elif kind == "surround":
            L.opus_multistream_surround_encoder_create.restype = vp; L.opus_multistream_surround_encoder_create.argtypes = [ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.c_char_p, ci, ctypes.POINTER(ci)]
            s, c = ci(), ci(); m = ctypes.create_string_buffer(8)
            self.st = L.opus_multistream_surround_encoder_create(48000, 6, 1, ctypes.byref(s), ctypes.byref(c), m, 2049, ctypes.byref(err)); self.fn = L.opus_multistream_encoder_ctl
        elif kind == "msdec":
            L.opus_multistream_decoder_create.restype = vp; L.opus_multistream_decoder_create.argtypes = [ci, ci, ci, ci, ctypes.c_char_p, ctypes.POINTER(ci)]
            self.st = L.opus_multistream_decoder_create(48000, 3, 2, 1, bytes([0, 1, 2]), ctypes.byref(err)); self.fn = L.opus_multistream_decoder_ctl
        elif kind == "projenc":
            L.opus_projection_ambisonics_encoder_create.restype = vp; L.opus_projection_ambisonics_encoder_create.argtypes = [ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ci, ctypes.POINTER(ci)]
            s, c = ci(), ci()
            self.st = L.opus_projection_ambisonics_encoder_create(48000, 4, 3, ctypes.byref(s), ctypes.byref(c), 2049, ctypes.byref(err)); self.fn = L.opus_projection_encoder_ctl
            self.streams, self.coupled = s.value, c.value
        elif kind == "projdec":
            # the demixing matrix comes from an encoder of the same library (bytes compared elsewhere: tests/test_hostemu_projection.py)
            e = Obj(L, "projenc"); size = e.get(6003)[1]
            buf = (ctypes.c_ubyte * size)(); e.fn.argtypes = [vp, ci, vp, ci]; assert e.fn(e.st, 6005, buf, size) == 0
            L.opus_projection_decoder_create.restype = vp; L.opus_projection_decoder_create.argtypes = [ci, ci, ci, ci, vp, ci, ctypes.POINTER(ci)]
            self.st = L.opus_projection_decoder_create(48000, 4, e.streams, e.coupled, buf, size, ctypes.byref(err)); self.fn = L.opus_projection_decoder_ctl
        assert self.st and err.value == 0, (kind, err.value)
        self.fn.restype = ci
    def set(self, req, v):
        self.fn.argtypes = [vp, ci, ci]; return self.fn(self.st, req, v)
    def setp(self, req, p):
        self.fn.argtypes = [vp, ci, vp]; return self.fn(self.st, req, p)
    def get(self, req):
        buf = (ctypes.c_int32 * 8)(*([0x5A5A5A5A] * 8)); self.fn.argtypes = [vp, ci, vp]
        r = self.fn(self.st, req, buf); return r, buf[0]
    def get2(self, req, sid):
        out = ctypes.c_void_p(); self.fn.argtypes = [vp, ci, ci, vp]
        return self.fn(self.st, req, sid, ctypes.byref(out))
    def noarg(self, req):
        self.fn.argtypes = [vp, ci]; return self.fn(self.st, req)

def _gets(a, b, tag, bad):
    for req in REQS:
        if req % 2 == 0 or req in SKIP or req == 6005: continue
        ra, rb = a.get(req), b.get(req)
        if ra[0] != rb[0] or (ra[0] == 0 and req not in NO_VALUE and ra[1] != rb[1]): bad.append((tag, "get", req, ra, rb))

@pytest.mark.parametrize("kind", ["enc", "dec", "msenc", "surround", "msdec", "projenc", "projdec"])
def test_ctl_sweep(kind):
    R, E = capi.load("ref_fxa"), capi.load(WHICH)
    a, b = Obj(R, kind), Obj(E, kind)
    bad = []
    _gets(a, b, "fresh", bad)
    for req in REQS:
        if req in SKIP or req % 2 == 1: continue
        if req == 4028: continue
        if req in (5120, 5122):
            for sid in (-1, 0, 1, 2, 3, 1000):
                ra, rb = a.get2(req, sid), b.get2(req, sid)
                if ra != rb: bad.append((kind, "state", req, sid, ra, rb))
            continue
        if req in PTR_EVEN:
            ra, rb = a.setp(req, None), b.setp(req, None)
            if ra != rb: bad.append((kind, "setp", req, ra, rb))
            continue
        changed = False
        for v in VALUES:
            ra, rb = a.set(req, v), b.set(req, v)
            if ra != rb: bad.append((kind, "set", req, v, ra, rb))
            changed = changed or ra == 0
        if changed: _gets(a, b, "%s after %d" % (kind, req), bad)
    for req in (6005,):
        buf = (ctypes.c_ubyte * 4096)()
        for size in (0, 1, 31, 32, 4096):
            for o in (a, b): o.fn.argtypes = [vp, ci, vp, ci]
            ra, rb = a.fn(a.st, req, buf, size), b.fn(b.st, req, buf, size)
            if ra != rb: bad.append((kind, "matrix", size, ra, rb))
    ra, rb = a.noarg(4028), b.noarg(4028)
    if ra != rb: bad.append((kind, "reset", ra, rb))
    _gets(a, b, "after reset", bad)
    assert not bad, "%d ctl differences, first: %s" % (len(bad), bad[:12])
