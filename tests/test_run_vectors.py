"""BASELINE config 1 plumbing: the committed `.bit` files (opus_demo framing, produced by the compiled reference encoder: tools/gen_bit_vectors.py) decoded as one batch
through tools/run_vectors_gpu.py -- final range vs the range stored in the file, PCM byte for byte vs the compiled reference decoder -- at several output rates / channel
counts; on the emulated C ABI here, on the product library on the MI355X."""
import glob, os, sys, pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from reflib import ref_fx
FILES = sorted(glob.glob(os.path.join(ROOT, "tests/golden/bitstreams/*.bit")))
pytestmark = pytest.mark.skipif(ref_fx() is None or not FILES, reason="compiled reference or vectors missing")

def _go(which, Fs, ch, files=FILES):
    import run_vectors_gpu
    npk, bad, _ = run_vectors_gpu.run(files, Fs, ch, which)
    assert npk > 100 and bad == 0, (npk, bad)

HAND = [f for f in FILES if not os.path.basename(f).startswith("mx_")]      # the 13 hand-picked files; mx_*: the 3 x 13 rows of the Encode+Decode matrix of tests/test_opus_encode.c:420-512 (tools/gen_bit_vectors.py)
def test_emu_vectors_48k_stereo(): _go("emu", 48000, 2, HAND)
def test_emu_vectors_16k_mono(): _go("emu", 16000, 1, HAND[::3])
def test_emu_vectors_reference_matrix(): _go("emu", 48000, 2, [f for f in FILES if os.path.basename(f).startswith("mx_")])
@pytest.mark.gpu
@pytest.mark.parametrize("Fs,ch", [(48000, 2), (48000, 1), (24000, 2), (16000, 1), (8000, 2)])
def test_gpu_vectors(Fs, ch): _go("gpu", Fs, ch)
