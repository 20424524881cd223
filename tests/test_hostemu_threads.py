"""Concurrent callers of the classic API (include/opus.h:425-429: any number of threads, each on its own state): calls that arrive while a launch is in flight share
the next launch, one wave per call (opus_amd/csrc/opus_call_combiner.h).  Whatever the grouping turns out to be, every thread must see exactly what the reference gives
it for its own stream: threads with different signals, settings, frame sizes, applications (both kernels), byte budgets and entry points run against one library at
once, then the same streams are encoded / decoded by the compiled reference, one after the other.  Here on the wave emulator (the queueing logic is host code and is the
same in the product); tests/test_gpu_classic_api.py runs this file's tests on the MI355X, where a launch really is shared."""
import ctypes, threading, numpy as np, pytest
import capi, signals
from reflib import ref_fx
from test_kernel_emu_silkdec import speechy
pytestmark = pytest.mark.skipif(ref_fx() is None, reason="oracle/_ref not built")
WHICH = "emu"

def _stats():
    L = capi.load(WHICH); out = (ctypes.c_longlong * 4)()
    L.opusgpu_classic_call_stats.argtypes = [ctypes.c_void_p]; L.opusgpu_classic_call_stats.restype = None
    L.opusgpu_classic_call_stats(out)
    return list(out)

# (Fs, channels, application, frame samples @48k, max bytes, ctls)
SHAPES = [
    (48000, 2, 2051, 960, 1276, dict(bitrate=128000, complexity=10)),
    (48000, 2, 2051, 960, 1276, dict(bitrate=64000, complexity=5)),
    (48000, 2, 2051, 480, 1276, dict(bitrate=96000)),
    (48000, 1, 2051, 960, 400, dict(bitrate=48000, vbr=0)),
    (16000, 1, 2048, 960, 1276, dict(bitrate=20000, complexity=10)),
    (16000, 1, 2048, 960, 1276, dict(bitrate=16000, inband_fec=1, packet_loss=10)),
    (48000, 2, 2049, 960, 1276, dict(bitrate=40000)),
    (48000, 2, 2049, 1920, 1276, dict(bitrate=32000)),
]

def _signal(k, Fs, ch, nframes, frame48):
    x = signals.music(nframes * frame48 // 960 + 1, seed=50 + k) if k % 3 else speechy(nframes * frame48 // 960 + 2, 2, 300 + k, 960)
    x = x[::48000 // Fs]
    return np.ascontiguousarray(x if ch == 2 else x[:, 0])

def _run_threads(fns):
    errs = []
    def wrap(f):
        try: f()
        except BaseException as e: errs.append(e)
    th = [threading.Thread(target=wrap, args=(f,)) for f in fns]
    for t in th: t.start()
    for t in th: t.join()
    if errs: raise errs[0]

def test_concurrent_encoders_of_many_shapes(nthreads=16, nframes=8):
    jobs = []
    for k in range(nthreads):
        Fs, ch, app, f48, maxb, ctl = SHAPES[k % len(SHAPES)]
        jobs.append((Fs, ch, app, f48 * Fs // 48000, maxb, ctl, _signal(k, Fs, ch, nframes, f48)))
    got = [None] * nthreads
    encs = [capi.Enc(WHICH, j[0], j[1], j[2], **j[5]) for j in jobs]
    s0 = _stats()
    def work(k):
        Fs, ch, app, n, maxb, ctl, x = jobs[k]
        got[k] = [encs[k].encode(x[i * n:(i + 1) * n], n, maxb) for i in range(nframes)]
    _run_threads([lambda k=k: work(k) for k in range(nthreads)])
    s1 = _stats()
    assert s1[0] - s0[0] == nthreads * nframes and 0 < s1[1] - s0[1] <= nthreads * nframes
    for k, (Fs, ch, app, n, maxb, ctl, x) in enumerate(jobs):
        r = capi.Enc("ref", Fs, ch, app, **ctl)
        want = [r.encode(x[i * n:(i + 1) * n], n, maxb) for i in range(nframes)]
        assert got[k] == want, (k, SHAPES[k % len(SHAPES)])

def test_concurrent_encoders_of_one_shape_share_launches(nthreads=12, nframes=6):
    """the serving case: identical settings, different signals; also the 24-bit entry point next to the 16-bit one (they never share a launch: the analysis input differs)"""
    L = capi.load(WHICH)
    L.opus_encode24.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int32]
    xs = [_signal(k, 48000, 2, nframes, 960) for k in range(nthreads)]
    encs = [capi.Enc(WHICH, 48000, 2, 2051, bitrate=96000, complexity=10) for _ in range(nthreads)]
    got = [None] * nthreads
    def work(k):
        out = []
        for i in range(nframes):
            fr = xs[k][i * 960:(i + 1) * 960]
            if k % 4 == 3:
                x24 = np.ascontiguousarray(fr.astype(np.int32) * 256); buf = (ctypes.c_ubyte * 1276)()
                n = L.opus_encode24(encs[k].st, x24.ctypes.data, 960, buf, 1276)
                out.append((bytes(buf[:max(n, 0)]), n, encs[k].get(4031) & 0xffffffff))
            else: out.append(encs[k].encode(fr, 960))
        got[k] = out
    _run_threads([lambda k=k: work(k) for k in range(nthreads)])
    for k in range(nthreads):
        r = capi.Enc("ref", 48000, 2, 2051, bitrate=96000, complexity=10)
        assert got[k] == [r.encode(xs[k][i * 960:(i + 1) * 960], 960) for i in range(nframes)], k

def test_concurrent_decoders(nthreads=12, nframes=8):
    """decoders at different output rates / channel counts / packet modes and lengths at once, with losses and an FEC recovery in some of the threads"""
    jobs = []
    for k in range(nthreads):
        Fs, ch, app, f48, maxb, ctl = SHAPES[k % len(SHAPES)]
        n = f48 * Fs // 48000; x = _signal(k, Fs, ch, nframes, f48)
        e = capi.Enc("ref", Fs, ch, app, **ctl)
        pk = [e.encode(x[i * n:(i + 1) * n], n, maxb)[0] for i in range(nframes)]
        outFs = (48000, 16000, 24000)[k % 3]; outch = 1 + (k // 2) % 2
        jobs.append((outFs, outch, pk, f48 * outFs // 48000, k % 4 == 1))
    decs = [capi.Dec(WHICH, j[0], j[1]) for j in jobs]
    got = [None] * nthreads
    def run(d, job):
        outFs, outch, pk, n, lossy = job; res = []
        for i, p in enumerate(pk):
            if lossy and i == 3: r = d.decode(b"", n)                       # a lost packet: concealment
            elif lossy and i == 4: r = d.decode(p, n, fec=1); res.append((r[0], r[1].tobytes(), r[2])); r = d.decode(p, n)
            else: r = d.decode(p, n)
            res.append((r[0], r[1].tobytes(), r[2]))
        return res
    s0 = _stats()
    def work(k): got[k] = run(decs[k], jobs[k])
    _run_threads([lambda k=k: work(k) for k in range(nthreads)])
    s1 = _stats()
    assert s1[2] - s0[2] >= nthreads * nframes and 0 < s1[3] - s0[3] <= s1[2] - s0[2]
    for k, job in enumerate(jobs):
        assert got[k] == run(capi.Dec("ref", job[0], job[1]), job), k
