"""MI355X: the encoder with its tonality / music analysis (the default at complexity 10) against the reference built with the float API (oracle/_ref/libopus_ref_fxa.so):
tests/test_hostemu_analysis.py with the product library opus_amd/libopus_amd.so in place of the emulated C ABI -- here the analysis' IEEE arithmetic is the GPU's own
(no fused multiply-adds in it, correctly rounded division and square root, the device's double-precision log)."""
import pytest
import test_hostemu_analysis as A
pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True)
def _product_library(monkeypatch):
    monkeypatch.setattr(A, "WHICH", "gpu")

from test_hostemu_analysis import (test_config_2_with_analysis, test_lowdelay_rates_and_sizes, test_audio_unforced_mode_decisions, test_speech_music_speech,
    test_forced_modes_and_hybrid, test_dtx_with_activity_probability, test_cbr_and_constrained, test_controls_midstream, test_signal_type_steers_the_lowdelay_application,
    test_24_bit_and_float_entry_points_feed_the_analysis_unrounded_samples, test_multistream_with_analysis,
    test_lowdelay_dtx_is_only_taken_on_analysed_or_silent_frames, test_lowdelay_dtx_sees_the_peak_tracked_before_it_was_switched_on)


@pytest.mark.parametrize("name,Fs,ch,app,ctl", [
    ("config 2", 48000, 2, 2051, ((4002, 128000), (4010, 10))),
    ("config 3", 16000, 1, 2048, ((11002, 1000), (4008, 1103), (4002, 24000), (4010, 10))),
    ("config 4", 48000, 2, 2049, ((11002, 1001), (4008, 1105), (4002, 128000), (4010, 10))),
    ("audio, nothing forced", 48000, 2, 2049, ((4002, 48000), (4010, 10)))])
def test_full_width_batch_with_analysis(name, Fs, ch, app, ctl):
    """the batch entry at BASELINE width (65,536 streams) with the analysis running on every wavefront, as bench.py times it: 32 distinct signals tiled over the streams,
    25 consecutive frames with the detector's state carried in HBM; every replica must agree with the reference built with the float API in length and final range,
    sampled replicas byte for byte (the last row leaves the mode, bandwidth and channel decisions to the detector's output)"""
    import numpy as np, capi, signals
    from test_kernel_emu_silkdec import speechy
    from test_gpu_parity import _oa
    oa = _oa()
    S, U, n, F = 65536, 32, Fs // 50, 25
    b = oa.EncoderBatch(S, channels=ch, application=app, Fs=Fs)
    b.ctl(oa.OPUS_AMD_SET_FLOAT_ANALYSIS_REQUEST, 1)
    for req, v in ctl: b.ctl(req, v)
    step = 48000 // Fs
    base = []
    for k in range(U):
        x = signals.music(F, seed=900 + k) if k % 2 else speechy(F + 1, 2, 700 + k, 960)
        x = x[:F * 960:step]
        base.append(np.ascontiguousarray(x if ch == 2 else x[:, 0]))
    req_name = {11002: "force_mode", 4008: "bandwidth", 4002: "bitrate", 4010: "complexity"}
    refs = [capi.Enc("ref_fxa", Fs, ch, app, **{req_name[r]: v for r, v in ctl}) for _ in range(U)]
    distinct = set()
    for i in range(F):
        fr = np.stack([base[k][i * n:(i + 1) * n].reshape(-1) for k in range(U)])
        pk, lens, rng = b.encode(np.tile(fr, (S // U, 1)), n)
        for k in range(U):
            a = refs[k].encode(np.ascontiguousarray(base[k][i * n:(i + 1) * n]), n)
            idx = np.arange(k, S, U)
            assert np.all(lens[idx] == a[1]) and np.all(rng[idx] == a[2]), (name, i, k, a[1], int(lens[k]))
            for s in (k, k + U * 1000, k + U * 2047): assert pk[s] == a[0], (name, i, k, s)
            distinct.add(a[0][0] >> 3)
    if name == "audio, nothing forced": assert len(distinct) >= 2, distinct
    b.close()
