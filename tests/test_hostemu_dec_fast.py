"""The decoder's CELT-only fast kernel in front of the general kernel against the general kernel alone, on the CPU wave emulator: identical PCM, sample counts, final ranges and
stream records over sequences that move streams between the two kernels (tools/dec_fast_check.py).  The emulator watches the fast kernel's dynamic LDS: it must not touch the A arena."""
import os, sys, pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from reflib import ref_fx
pytestmark = pytest.mark.skipif(ref_fx() is None, reason="oracle/_ref not built")

def test_emu_fast_decoder_equals_general_decoder(tmp_path):
    import dec_fast_check
    assert not dec_fast_check.compare("emu", tmpdir=str(tmp_path), verbose=False)

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
