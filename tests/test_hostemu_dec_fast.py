"""The decoder's steady-state kernels (the CELT-only fast kernel; the lane = stream SILK kernel and the hybrid packets' CELT-layer kernel behind it) in front of the general
kernel against the general kernel alone, on the CPU wave emulator: identical PCM, sample counts, final ranges and stream records over sequences that move streams between the
kernels (tools/dec_fast_check.py).  The emulator watches the fast kernel's dynamic LDS: it must not touch the A arena."""
import os, sys, pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from reflib import ref_fx
pytestmark = pytest.mark.skipif(ref_fx() is None, reason="oracle/_ref not built")

LONG = os.environ.get("OPUS_AMD_LONG_TESTS") == "1"          # the default CPU suite: the cases that reach every kernel and every hand-over; OPUS_AMD_LONG_TESTS=1 (and the GPU test): all of them
QUICK = "celt_stereo,celt_5ms_12k,mono_coded,audio_auto,voip_16k,silk_stereo,silk_mb_60,silk_fec,silk_celt_switch,silk_bw_switch,hyb_stereo,hyb_24k_out,hyb_celt_switch"

def test_emu_fast_decoder_equals_general_decoder(tmp_path, monkeypatch):
    if not LONG: monkeypatch.setenv("DEC_FAST_CASES", QUICK)
    import importlib, dec_fast_check
    importlib.reload(dec_fast_check)                             # (the case filter is read at import)
    bad, lane = dec_fast_check.compare("emu", tmpdir=str(tmp_path), verbose=False, with_stats=True)
    assert not bad, bad
    assert lane["voip_16k"][0] > 0 and lane["silk_stereo"][0] > 0 and lane["hyb_stereo"][0] > 0           # the lane kernel took packets ...
    assert lane["silk_celt_switch"][1] > 0 and lane["hyb_celt_switch"][1] > 0                             # ... and handed the ones with a redundant CELT frame on
    assert lane["celt_stereo"] == [0, 0] or lane["celt_stereo"] == (0, 0)

def test_emu_lane_kernel_switch_per_batch():
    """opusgpu_dec_batch_set_lane_kernel(b, 0) == the default, in one process"""
    import numpy as np, hostemu, opus_amd, dec_fast_check
    saved = (opus_amd.LIB_PATH, opus_amd._lib)
    opus_amd.LIB_PATH = hostemu.build_emu_lib(); opus_amd._lib = None
    try:
        for name in ("voip_16k", "hyb_stereo"):
            import importlib; os.environ.pop("DEC_FAST_CASES", None); importlib.reload(dec_fast_check)
            Fs, ch, app, ctl, ms, frames, loss = dec_fast_check.CASES[name][:7]
            seqs = dec_fast_check.make_packets(name); S = len(seqs); n = int(Fs * ms // 1000)
            a = opus_amd.DecoderBatch(S, channels=ch, Fs=Fs); b = opus_amd.DecoderBatch(S, channels=ch, Fs=Fs); b.set_lane_kernel(False)
            took = 0
            for f in range(frames):
                pk = [seqs[s][f] for s in range(S)]
                x, y = a.decode(pk, n), b.decode(pk, n)
                assert all(np.array_equal(p, q) for p, q in zip(x, y)), (name, f)
                took += a.lane_stats()[0]; assert b.lane_stats() == (0, 0)
            assert took > 0 and all(a.export_state(s) == b.export_state(s) for s in range(S)), name
            a.close(); b.close()
    finally:
        opus_amd.LIB_PATH, opus_amd._lib = saved

def test_emu_fast_kernel_switch_per_batch():
    """opusgpu_dec_batch_set_fast_kernel(b, 0) == the default, in one process: PCM, sample counts, final ranges and the stream records of two batches fed the same packets"""
    import numpy as np, hostemu, opus_amd, dec_fast_check
    saved = (opus_amd.LIB_PATH, opus_amd._lib)
    opus_amd.LIB_PATH = hostemu.build_emu_lib(); opus_amd._lib = None
    try:
        for name in ("celt_stereo", "audio_auto"):
            Fs, ch, app, ctl, ms, frames, loss = dec_fast_check.CASES[name][:7]
            seqs = dec_fast_check.make_packets(name); S = len(seqs); n = int(Fs * ms // 1000)
            a = opus_amd.DecoderBatch(S, channels=ch, Fs=Fs); b = opus_amd.DecoderBatch(S, channels=ch, Fs=Fs); b.set_fast_kernel(False)
            for f in range(frames):
                pk = [seqs[s][f] for s in range(S)]
                x, y = a.decode(pk, n), b.decode(pk, n)
                assert all(np.array_equal(p, q) for p, q in zip(x, y)), (name, f)
            assert all(a.export_state(s) == b.export_state(s) for s in range(S)), name
            a.close(); b.close()
    finally:
        opus_amd.LIB_PATH, opus_amd._lib = saved                  # (this process's other tests -- and opus_amd.build() -- see the product library again)

PVQ_CASES = ("celt_stereo", "celt_mono_10", "hyb_stereo", "hyb_mono_10", "mono_coded", "stereo_to_mono", "audio_auto")

def pvq_stage_switch(lib, cases=PVQ_CASES):
    """opusgpu_dec_batch_set_pvq_stage(b, 1) == (b, 0), in one process: the bands of the steady-state CELT frames on oa_celt_dpvq_kernel (four streams per wave,
    opus_amd/csrc/celt_dec_pvq4.h) against the one-wave-per-stream band decoder -- PCM, sample counts, final ranges and stream records"""
    import numpy as np, opus_amd, dec_fast_check, importlib
    saved = (opus_amd.LIB_PATH, opus_amd._lib)
    opus_amd.LIB_PATH = lib; opus_amd._lib = None
    try:
        os.environ.pop("DEC_FAST_CASES", None); importlib.reload(dec_fast_check)
        for name in cases:
            Fs, ch, app, ctl, ms, frames, loss = dec_fast_check.CASES[name][:7]
            seqs = dec_fast_check.make_packets(name); S = len(seqs)
            if len(dec_fast_check.CASES[name]) > 9: Fs = dec_fast_check.CASES[name][9]
            n = int(Fs * ms // 1000); dch = dec_fast_check.dec_channels(name)
            a = opus_amd.DecoderBatch(S, channels=dch, Fs=Fs); b = opus_amd.DecoderBatch(S, channels=dch, Fs=Fs); a.set_pvq_stage(1); b.set_pvq_stage(0)
            took = 0
            for f in range(frames):
                pk = [seqs[s][f] for s in range(S)]
                x, y = a.decode(pk, n), b.decode(pk, n)
                assert all(np.array_equal(p, q) for p, q in zip(x, y)), (name, f)
                took += a.pvq_stats(); assert b.pvq_stats() == 0
            assert took > 0 and all(a.export_state(s) == b.export_state(s) for s in range(S)), name
            a.close(); b.close()
    finally:
        opus_amd.LIB_PATH, opus_amd._lib = saved

def test_emu_pvq_stage_switch_per_batch():
    import hostemu
    pvq_stage_switch(hostemu.build_emu_lib())

def test_emu_fast_decoder_with_the_pvq_stage_forced_equals_general_decoder(tmp_path, monkeypatch):
    """the whole check of the first test once more with OPUS_AMD_DEC_PVQ4=1 (the process default the narrow test batches would otherwise never reach)"""
    monkeypatch.setenv("OPUS_AMD_DEC_PVQ4", "1")
    monkeypatch.setenv("DEC_FAST_CASES", "celt_,hyb_,mono_coded,stereo_to_mono,audio_auto,silk_celt_switch" if LONG else "celt_stereo,celt_24k,celt_5ms_12k,hyb_stereo,hyb_24k_out,hyb_celt_switch,mono_coded,stereo_to_mono")
    import importlib, dec_fast_check
    importlib.reload(dec_fast_check)
    assert not dec_fast_check.compare("emu", tmpdir=str(tmp_path), verbose=False)

def pvq_stage_vs_reference(lib, cases=("celt_stereo", "celt_mono_10", "celt_510k", "celt_24k", "hyb_stereo", "mono_coded")):
    """the PVQ stage forced, against the COMPILED REFERENCE decoder (oracle/_ref), packet by packet: PCM, sample counts, final ranges -- lost packets included (the stream leaves
    the pipeline for the general kernel and comes back)"""
    import numpy as np, opus_amd, dec_fast_check, importlib, capi
    saved = (opus_amd.LIB_PATH, opus_amd._lib)
    opus_amd.LIB_PATH = lib; opus_amd._lib = None
    try:
        os.environ.pop("DEC_FAST_CASES", None); importlib.reload(dec_fast_check)
        for name in cases:
            Fs, ch, app, ctl, ms, frames, loss = dec_fast_check.CASES[name][:7]
            seqs = dec_fast_check.make_packets(name); S = len(seqs)
            if len(dec_fast_check.CASES[name]) > 9: Fs = dec_fast_check.CASES[name][9]
            n = int(Fs * ms // 1000); dch = dec_fast_check.dec_channels(name)
            a = opus_amd.DecoderBatch(S, channels=dch, Fs=Fs); a.set_pvq_stage(1)
            refs = [capi.Dec("ref", Fs, dch) for _ in range(S)]
            took = 0
            for f in range(frames):
                pk = [seqs[s][f] for s in range(S)]
                pcm, ns, rng = a.decode(pk, n); took += a.pvq_stats()
                for s in range(S):
                    x = refs[s].decode(pk[s], n)
                    assert x[0] == int(ns[s]) == n and (x[2] == int(rng[s]) or not pk[s]), (name, f, s, x[0], int(ns[s]), hex(x[2]), hex(int(rng[s])))
                    assert np.array_equal(x[1], pcm[s, :n]), (name, f, s, np.nonzero(x[1] != pcm[s, :n])[0][:6])
            assert took > 0, name
            a.close()
    finally:
        opus_amd.LIB_PATH, opus_amd._lib = saved

def test_emu_pvq_stage_vs_reference():
    import hostemu
    pvq_stage_vs_reference(hostemu.build_emu_lib())
