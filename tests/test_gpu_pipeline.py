"""MI355X: the settings matrix and the settings fuzzers ONCE MORE with every 10 / 20 ms call of the SILK-capable encoder forced through the kernel
pipeline (OPUS_AMD_SET_KERNEL_PIPELINE(4): front / pred lane + wave kernels / quantiser / back, opus_amd/csrc/opus_sh_split.h -- the path a >= 64-stream launch, i.e. the bench,
takes by itself --, (3): the pred stage as one wave-per-channel kernel, and (1): the pipeline without a pred stage of its own), against the compiled reference:
tests/test_gpu_silkenc.py's matrix (SILK, hybrid, CELT-only and automatic modes, complexities, hard CBR, tight buffers, DTX, in-band FEC), the encoder / sparse / batch-ABI /
multistream fuzzers of tests/test_hostemu_fuzz.py, the pinned seeds and the deterministic case of the round-4 review (the LBRR side stream owed after FEC goes 1 -> 0).
The small launches of the other GPU test files take the one-kernel path; here the same cases meet the reference through the pipeline."""
import pytest
import test_gpu_silkenc as G, test_hostemu_fuzz as Z
pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True, params=[4, 3, 1], ids=["front-pred(lanes)-quant-back", "front-pred-quant-back", "front-quant-back"])
def _through_the_pipeline(request, monkeypatch):
    monkeypatch.setattr(G, "PIPELINE", request.param); monkeypatch.setattr(Z, "WHICH", "gpu"); monkeypatch.setattr(Z, "PIPELINE", request.param)

from test_gpu_silkenc import (test_gpu_config3_silk_voip_16k, test_gpu_silk_complexities, test_gpu_silk_matrix, test_gpu_hybrid_matrix, test_gpu_silk_small_buffer,
    test_gpu_celt_only_and_auto_modes_in_audio_voip, test_gpu_dtx, test_gpu_inband_fec)
from test_hostemu_fuzz import test_pending_lbrr_after_fec_is_switched_off

def test_settings_fuzz_through_the_pipeline_on_the_device():
    for seed in [248, 300, 346] + list(range(2000, 2016)): Z.fuzz(seed)

def test_sparse_settings_fuzz_through_the_pipeline_on_the_device():
    for seed in [7770080, 102, 5179] + list(range(2600, 2616)): Z.fuzz_sparse(seed)

def test_batch_abi_fuzz_through_the_pipeline_on_the_device():
    for seed in [7770010, 5, 118] + list(range(2420, 2432)): Z.fuzz_batch(seed)

def test_multistream_settings_fuzz_through_the_pipeline_on_the_device():
    for seed in [101, 197] + list(range(2400, 2408)): Z.fuzz_ms(seed)

def test_pipeline_is_a_per_batch_switch():
    """two batches of one process on different settings of the switch, the same input: identical packets, and the pipeline batch did keep its calls (split statistics)"""
    import ctypes, numpy as np, opus_amd as oa
    L = oa.lib(); L.opusgpu_enc_batch_split_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    out = []
    for mode in (0, 4):
        b = oa.EncoderBatch(6, channels=1, application=2048, Fs=16000); b.ctl(11902, mode); assert b.get(11903, 0) == mode
        for k, v in dict(force_mode=1000, bitrate=24000, complexity=10).items(): b.ctl(G.REQ[k], v)
        sig = [G.speech(16000, 0.4, 1, 50 + s) for s in range(6)]
        pk = [b.encode(np.stack([sig[s][f * 320:(f + 1) * 320].reshape(-1) for s in range(6)]), 320)[0] for f in range(12)]
        kept, declined = ctypes.c_uint32(), ctypes.c_uint32(); assert L.opusgpu_enc_batch_split_stats(b._b, ctypes.byref(kept), ctypes.byref(declined)) == 0
        out.append((pk, kept.value, declined.value)); b.close()
    assert out[0][0] == out[1][0]
    assert out[0][1] == 0 and out[1][1] == 72 and out[1][2] == 0, (out[0][1:], out[1][1:])
