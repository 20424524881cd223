"""MI355X: the classic-API parity tests of tests/test_hostemu_encoder_modes.py and tests/test_hostemu_decoder_rates.py (reference vs product, packet bytes / PCM and
final range call by call) with the product library opus_amd/libopus_amd.so in place of the emulated C ABI: mode switches with redundancy and prefills, multi-frame
packets, API rates below 48 kHz in both directions, CBR padding, settings fuzz."""
import pytest
import test_hostemu_encoder_modes as M, test_hostemu_decoder_rates as D, test_hostemu_api_limits as A, test_hostemu_threads as T, test_hostemu_fuzz as Z
pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True)
def _product_library(monkeypatch):
    monkeypatch.setattr(M, "WHICH", "gpu"); monkeypatch.setattr(D, "WHICH", "gpu"); monkeypatch.setattr(A, "WHICH", "gpu"); monkeypatch.setattr(T, "WHICH", "gpu"); monkeypatch.setattr(Z, "WHICH", "gpu")

from test_hostemu_encoder_modes import (test_silk_celt_switches_mono, test_hybrid_celt_switches_stereo, test_switches_10ms_and_short_frames, test_auto_mode_rate_sweep,
    test_silk_bandwidth_switch, test_long_frames_celt_and_hybrid, test_long_frames_silk, test_celt_below_48k, test_cbr_padding_and_tiny_buffers, test_settings_fuzz)
from test_hostemu_decoder_rates import test_celt_rates, test_silk_rates, test_hybrid_and_switches
from test_hostemu_api_limits import (test_hard_cbr_above_510_kbps, test_bitrate_max_pads_a_long_call_to_the_whole_buffer, test_packet_pad_to_tens_of_kilobytes,
    test_ms_encoder_batch_follows_the_encoder_in_use, test_ms_hard_cbr_above_the_frame_cap, test_ms_decoder_gain_and_fec_frame_size_check,
    test_ms_decode_sub_packet_longer_than_six_frames)
from test_hostemu_threads import test_concurrent_encoders_of_many_shapes, test_concurrent_encoders_of_one_shape_share_launches, test_concurrent_decoders

def test_concurrent_callers_share_launches_on_the_device():
    """64 threads, each with its own encoder of BASELINE config 2 and 25 frames to encode through opus_encode(): the calls must coalesce (far fewer launches than
    calls), the wall time must be far below 64 x the single-caller time, and every thread's packets must be the reference's"""
    import time, numpy as np, capi, signals
    nt, nf = 64, 25
    xs = [signals.music(nf, seed=800 + k) for k in range(nt)]
    encs = [capi.Enc("gpu", 48000, 2, 2051, bitrate=128000, complexity=10) for _ in range(nt)]
    solo = capi.Enc("gpu", 48000, 2, 2051, bitrate=128000, complexity=10)
    solo.encode(xs[0][:960], 960)                                            # first call: device arrays of the shape
    t0 = time.time()
    for i in range(nf): solo.encode(xs[0][i * 960:(i + 1) * 960], 960)
    t_solo = time.time() - t0
    got = [None] * nt
    def work(k): got[k] = [encs[k].encode(xs[k][i * 960:(i + 1) * 960], 960) for i in range(nf)]
    s0 = T._stats(); t0 = time.time()
    T._run_threads([lambda k=k: work(k) for k in range(nt)])
    t_all = time.time() - t0; s1 = T._stats()
    calls, launches = s1[0] - s0[0], s1[1] - s0[1]
    print("classic opus_encode: 1 caller %.1f calls/s; %d callers %.1f calls/s, %.1f calls per launch" % (nf / t_solo, nt, calls / t_all, calls / launches))
    assert calls == nt * nf and launches * 4 <= calls, (calls, launches)
    assert t_all < 0.25 * nt * t_solo, (t_all, t_solo)
    for k in range(0, nt, 7):
        r = capi.Enc("ref", 48000, 2, 2051, bitrate=128000, complexity=10)
        assert got[k] == [r.encode(xs[k][i * 960:(i + 1) * 960], 960) for i in range(nf)], k

@pytest.mark.parametrize("block", range(2))
def test_settings_fuzz_more_seeds_on_the_device(block):
    """tests/test_hostemu_fuzz.py's fuzz, 34 seeds per block (the emulator runs 20 in the CPU suite; 248 seeds ran here once: profiles/r03_o): one encoder per seed through ten
    random setting changes, every packet against the reference (even seeds: float API with the analysis; odd: without)"""
    for seed in [248, 252, 300, 304, 306, 337, 346, 423][4 * block:4 * block + 4] + list(range(1000 + 30 * block, 1030 + 30 * block)): Z.fuzz(seed)

def test_multistream_settings_fuzz_on_the_device():
    for seed in [101, 197] + list(range(400, 430)): Z.fuzz_ms(seed)

def test_sparse_settings_fuzz_on_the_device():
    for seed in [102, 134, 172] + list(range(600, 630)): Z.fuzz_sparse(seed)

def test_decoder_fuzz_on_the_device():
    for seed in [6, 16, 39, 115, 143, 150, 182] + list(range(300, 330)): Z.fuzz_dec(seed)

def test_batch_abi_fuzz_on_the_device():
    for seed in [5, 118] + list(range(420, 440)): Z.fuzz_batch(seed)

def test_multistream_decoder_fuzz_on_the_device():
    for seed in range(200, 220): Z.fuzz_ms_dec(seed)

def test_entry_point_fuzz_on_the_device():
    for seed in [0, 3, 25, 26] + list(range(80, 96)): Z.fuzz_entry(seed)

def test_projection_fuzz_on_the_device():
    for seed in range(120, 130): Z.fuzz_proj(seed)

def test_packet_toolkit_fuzz_with_the_product_library():
    for seed in range(200, 210): Z.fuzz_packets(seed)

@pytest.mark.timeout(300)
def test_float_output_fuzz_with_the_product_library():
    for seed in range(80, 90): Z.fuzz_float_out(seed)
