"""The CPU wave emulator schedules its 64 fibers in lane order between two rendezvous, so a read of another lane's LDS write without a wv_sync in between -- a race
on the lockstep hardware -- cannot be seen under one order.  OA_EMU_REVERSE=1 runs lane 63 first; OA_EMU_CHECK_UNI=1 additionally checks that every wv_uni argument
(v_readfirstlane on the GPU) is the same in all lanes and every wv_bcast lane index (v_readlane) is uniform and < 64 (tests/emu/wave_emu.h).  The whole suite is meant
to be run that way by hand (`OA_EMU_REVERSE=1 OA_EMU_CHECK_UNI=1 pytest -m "not gpu"`); this file keeps a slice of it in every run: opus_demo's hybrid, CELT, SILK and
random-frame-size / FEC / loss schedules through the complete encoder and decoder, PCM identical to the reference-linked run."""
import os, pytest
import hostemu
import test_zz_reference_programs as R

@pytest.mark.skipif(not os.path.isdir(hostemu.REF) and not os.path.exists(os.path.join(R.ROOT, "oracle/_ref/reftests/ref/opus_demo")), reason="no reference tree")
@pytest.mark.parametrize("case", [0, 3, 4, 7, 12])
def test_emu_reversed_lane_order_and_uniformity(case, tmp_path, monkeypatch):
    monkeypatch.setenv("OA_EMU_REVERSE", "1"); monkeypatch.setenv("OA_EMU_CHECK_UNI", "1")
    R._opus_demo_codec("emu", tmp_path, *R.DEMO_MODES[case], seconds=1.2)
