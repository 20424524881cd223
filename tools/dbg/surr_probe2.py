import ctypes, sys, numpy as np, os
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.environ.get("PROBE_LIB") or os.path.join(here, "../../opus_amd/libopus_amd.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
stage = sys.argv[1] if len(sys.argv) > 1 else "both"
L.opus_multistream_surround_encoder_create.restype = vp
L.opus_multistream_surround_encoder_create.argtypes = [ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), vp, ci, ctypes.POINTER(ci)]
L.opus_multistream_encoder_ctl.argtypes = [vp, ci, ci]
L.opus_multistream_encode.argtypes = [vp, vp, ci, vp, ci]
s, c, err = ci(), ci(), ci()
mapping = (ctypes.c_ubyte * 8)()
e = L.opus_multistream_surround_encoder_create(24000, 3, 1, ctypes.byref(s), ctypes.byref(c), mapping, 2049, ctypes.byref(err))
for req, v in ((4024, 3001), (4006, 1), (4020, 1), (4042, 0), (4022, -1000), (4046, 0), (4016, 0), (4010, 0), (4004, 1101), (4008, 1101), (4036, 8), (4012, 1), (4002, 84315)): L.opus_multistream_encoder_ctl(e, req, v)
p0 = np.array(open(os.path.join(here, "surr_pcm0.txt")).read().split(), np.int16); p1 = np.zeros(1440 * 3, np.int16); _v = np.array(open(os.path.join(here, "surr_pcm1.txt")).read().split(), np.int16); p1[:len(_v)] = _v
buf = (ctypes.c_ubyte * 7380)()
print("call 1 ->", L.opus_multistream_encode(e, p0.ctypes.data, 960, buf, 7380), flush=True)
if stage == "first": sys.exit(0)
ctls = [(4024, 3002), (4006, 1), (4020, 0), (4042, 1), (4022, -1000), (4046, 1), (4016, 1), (4010, 6), (4004, 1101), (4008, -1000), (4036, 9), (4012, 1), (4014, 5), (4002, 775410)]
skip = sys.argv[2].split(",") if len(sys.argv) > 2 else []
for req, v in ctls:
    if str(req) in skip: continue
    L.opus_multistream_encoder_ctl(e, req, v)
print("call 2 ->", L.opus_multistream_encode(e, p1.ctypes.data, 1440, buf, 7380), flush=True)
