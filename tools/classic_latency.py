"""tools/classic_latency.py — per-call latency of the classic (one object, one frame per call) API on the MI355X: opus_encode / opus_decode through
opus_amd/libopus_amd.so next to the compiled reference on one host core.  One wave per call: this is the boundary's worst case and is reported, not hidden."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import capi

def run(which, Fs, ch, app, frame, n, **ctl):
    rs = np.random.RandomState(7)
    t = np.arange(frame * n * ch) / (Fs * ch)
    pcm = (6000 * np.sin(2 * np.pi * 440 * t) + 1500 * rs.randn(t.size)).astype(np.int16).reshape(n, frame * ch)
    e = capi.Enc(which, Fs, ch, app, **ctl)
    pk = []
    e.encode(pcm[0], frame)
    t0 = time.perf_counter()
    for i in range(n): pk.append(e.encode(pcm[i], frame)[0])
    te = (time.perf_counter() - t0) / n
    d = capi.Dec(which, Fs, ch)
    d.decode(pk[0], frame)
    t0 = time.perf_counter()
    for p in pk: d.decode(p, frame)
    td = (time.perf_counter() - t0) / n
    return te * 1e6, td * 1e6

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    cases = [("CELT 48k stereo 20 ms 128k (lowdelay)", 48000, 2, 2051, 960, dict(bitrate=128000)),
             ("hybrid/auto 48k stereo 20 ms 64k (audio)", 48000, 2, 2049, 960, dict(bitrate=64000)),
             ("SILK 16k mono 20 ms 16k (voip)", 16000, 1, 2048, 320, dict(bitrate=16000)),
             ("CELT 48k mono 2.5 ms (lowdelay)", 48000, 1, 2051, 120, dict(bitrate=64000))]
    for name, Fs, ch, app, frame, ctl in cases:
        g = run("gpu", Fs, ch, app, frame, n, **ctl)
        try: r = run("ref", Fs, ch, app, frame, n, **ctl)
        except Exception as ex: r = (float("nan"), float("nan"))
        print("%-44s gpu enc %8.1f us dec %8.1f us | reference (1 core) enc %7.1f us dec %7.1f us" % (name, g[0], g[1], r[0], r[1]), flush=True)
