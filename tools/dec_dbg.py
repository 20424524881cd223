"""bring-up aid: three SILK-only packets through a DecoderBatch (the fast kernel hands them to the general kernel)"""
import sys, os, time, ctypes, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
print("start", flush=True)
import opus_amd
L = ctypes.CDLL(os.path.join(ROOT, "oracle/_ref/libopus_ref_fx.so"))
L.opus_encoder_create.restype = ctypes.c_void_p; L.opus_encoder_create.argtypes = [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int)]
L.opus_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
Fs, ch, n = 16000, 1, 320
err = ctypes.c_int(); e = L.opus_encoder_create(Fs, ch, 2048, ctypes.byref(err)); L.opus_encoder_ctl(e, 4002, 24000)
t = np.arange(n * 6) / Fs
sig = (8000 * np.sin(2 * np.pi * 220 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t))).astype(np.int16)
out = (ctypes.c_ubyte * 1500)(); pk = []
for i in range(4):
    x = np.ascontiguousarray(sig[i * n:(i + 1) * n]); k = L.opus_encode(e, x.ctypes.data, n, out, 1276); pk.append(bytes(out[:k]))
print("encoded", [len(p) for p in pk], hex(pk[0][0]), flush=True)
b = opus_amd.DecoderBatch(3, channels=ch, Fs=Fs)
print("batch", flush=True)
t0 = time.time()
for p in pk:
    pcm, ns, rng = b.decode([p, p, p], n)
    print("ok", ns, round(time.time() - t0, 3), flush=True)
