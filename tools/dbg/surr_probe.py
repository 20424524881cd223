import ctypes, sys, numpy as np, os
L = ctypes.CDLL(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opus_amd/libopus_amd.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
variant = sys.argv[1]
Fs, nch, frame = 24000, 3, 960
fam = 1
cpx, fec, app = 0, 1, 2049
if variant == "f20": frame = 480
if variant == "nofec": fec = 0
if variant == "fam255": fam = 255
if variant == "c5": cpx = 5
if variant == "48k": Fs = 48000; frame = 1920
if variant == "48k20": Fs = 48000; frame = 960
if variant == "lowdelay": app = 2051
L.opus_multistream_surround_encoder_create.restype = vp
L.opus_multistream_surround_encoder_create.argtypes = [ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), vp, ci, ctypes.POINTER(ci)]
L.opus_multistream_encoder_ctl.argtypes = [vp, ci, ci]
L.opus_multistream_encode.argtypes = [vp, vp, ci, vp, ci]
s, c, err = ci(), ci(), ci()
mapping = (ctypes.c_ubyte * 8)()
e = L.opus_multistream_surround_encoder_create(Fs, nch, fam, ctypes.byref(s), ctypes.byref(c), mapping, app, ctypes.byref(err))
assert e, err.value
for req, v in ((4024, 3001), (4006, 1), (4020, 1), (4010, cpx), (4004, 1101), (4008, 1101), (4036, 8), (4012, fec), (4002, 84315)): L.opus_multistream_encoder_ctl(e, req, v)
rng = np.random.default_rng(1)
pcm = (rng.standard_normal((frame, nch)) * 3000).astype(np.int16)
buf = (ctypes.c_ubyte * 7380)()
for k in range(3):
    n = L.opus_multistream_encode(e, pcm.ctypes.data, frame, buf, 7380)
    print(variant, "frame", k, "->", n, flush=True)
