"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/opus_amd.h declares,
keeps libopus' error/argument behaviour on the no-compute entry points, and fails loudly (no CPU fallback) when
there is no HIP device.  No kernel is launched here."""
import ctypes, os, re, subprocess
import pytest
import opus_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "opus_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = src.replace("#define OPUS_AMD_EXPORT", "")
    names = re.findall(r"OPUS_AMD_EXPORT\s+[^;(]*?\b(\w+)\s*\(", src)
    assert len(names) >= 20
    return names


@pytest.fixture(scope="module")
def lib():
    opus_amd.build()
    return ctypes.CDLL(opus_amd.LIB_PATH)


def test_every_declared_symbol_is_exported(lib):
    dyn = subprocess.run(["nm", "-D", "--defined-only", opus_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in dyn.splitlines() if " T " in l}
    for n in _declared():
        assert n in exported, n
        getattr(lib, n)
    # nothing else leaks out of the library (hidden visibility): only the declared ABI and toolchain symbols
    extra = {e for e in exported if not e.startswith("_") and e not in set(_declared())}
    assert not extra, extra


def _reference_exports():
    """every OPUS_EXPORT function the reference's public headers declare (custom modes excluded: CUSTOM_MODES is not part of the build that is the oracle).
    Parsed from /root/reference when it is there (and the committed list refreshed), else read from the committed list."""
    fixture = os.path.join(ROOT, "tests/golden/opus_exports.txt")
    inc = "/root/reference/include"
    if os.path.isdir(inc):
        names = []
        for h in ["opus.h", "opus_multistream.h", "opus_projection.h", "opus_defines.h"]:
            src = re.sub(r"/\*.*?\*/", "", open(os.path.join(inc, h)).read(), flags=re.S)
            src = re.sub(r"^\s*#.*$", "", src, flags=re.M)                     # (the macro's own definition lines)
            names += [(m.group(1), h) for m in re.finditer(r"OPUS_EXPORT\b[^;{]*?\b(\w+)\s*\(", src)]
        text = "# OPUS_EXPORT functions of the reference's public headers, refreshed by tests/test_abi_symbols.py whenever /root/reference is present\n" + "".join("%s %s\n" % n for n in names)
        if not os.path.exists(fixture) or open(fixture).read() != text: open(fixture, "w").write(text)
    return [l.split()[0] for l in open(fixture) if l.strip() and not l.startswith("#")]


def test_every_reference_export_is_exported(lib):
    """the drop-in claim, symbol by symbol: nm -D of the product against the reference's own headers"""
    dyn = subprocess.run(["nm", "-D", "--defined-only", opus_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in dyn.splitlines() if " T " in l}
    ref = _reference_exports()
    assert len(ref) >= 85
    missing = [n for n in ref if n not in exported]
    assert not missing, missing


def test_sizes_and_strings(lib):
    assert lib.opus_encoder_get_size(1) == lib.opus_encoder_get_size(2) > lib.opusgpu_enc_state_size() > 0
    assert lib.opus_encoder_get_size(0) == 0 and lib.opus_encoder_get_size(3) == 0          # opus_encoder.c:194-202
    lib.opus_strerror.restype = ctypes.c_char_p
    assert lib.opus_strerror(0) == b"success" and lib.opus_strerror(-1) == b"invalid argument"   # celt/celt.c:342
    assert lib.opus_strerror(-5) == b"request not implemented" and lib.opus_strerror(-99) == b"unknown error"
    lib.opus_get_version_string.restype = ctypes.c_char_p
    assert b"opus" in lib.opus_get_version_string()


def test_create_argument_errors(lib):
    err = ctypes.c_int(7)
    lib.opus_encoder_create.restype = ctypes.c_void_p
    lib.opus_encoder_destroy.argtypes = [ctypes.c_void_p]
    for Fs, ch, app, want in [(44100, 2, 2051, -1), (48000, 3, 2051, -1), (48000, 2, 1234, -1), (44100, 1, 2048, -1)]:
        p = lib.opus_encoder_create(Fs, ch, app, ctypes.byref(err))
        assert p is None and err.value == want, (Fs, ch, app, err.value)
    for Fs, ch, app in [(48000, 2, 2049), (16000, 1, 2048), (8000, 1, 2052), (16000, 1, 2051), (24000, 2, 2053)]:  # the SILK-capable encoder: AUDIO / VOIP / RESTRICTED_SILK at any API rate
        p = lib.opus_encoder_create(Fs, ch, app, ctypes.byref(err))
        assert p and err.value == 0
        v = ctypes.c_int32(0)
        assert lib.opus_encoder_ctl(ctypes.c_void_p(p), 11002, ctypes.c_int32(1000)) == 0 and lib.opus_encoder_ctl(ctypes.c_void_p(p), 11002, ctypes.c_int32(7)) == -1
        assert lib.opus_encoder_ctl(ctypes.c_void_p(p), 4029, ctypes.byref(v)) == 0 and v.value == Fs
        lib.opus_encoder_destroy(ctypes.c_void_p(p))
    p = lib.opus_encoder_create(48000, 2, 2051, ctypes.byref(err))
    assert p and err.value == 0
    v = ctypes.c_int32(0)
    lib.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.opus_encoder_ctl(p, 4002, ctypes.c_int32(96000)) == 0
    assert lib.opus_encoder_ctl(p, 4003, ctypes.byref(v)) == 0 and v.value == 96000
    assert lib.opus_encoder_ctl(p, 4010, ctypes.c_int32(11)) == -1                       # complexity out of range
    assert lib.opus_encoder_ctl(p, 4029, ctypes.byref(v)) == 0 and v.value == 48000
    assert lib.opus_encoder_ctl(p, 4998, ctypes.c_int32(0)) == -5                         # unknown request (opus_encoder.c:3352)
    lib.opus_encoder_destroy(p)


def test_no_cpu_fallback(lib):
    """Without a HIP device the batch constructor must fail (OPUS_INTERNAL_ERROR), never silently run on the host."""
    if lib.opusgpu_device_count() > 0:
        pytest.skip("a GPU is visible")
    err = ctypes.c_int(0)
    lib.opusgpu_enc_batch_create.restype = ctypes.c_void_p
    assert lib.opusgpu_enc_batch_create(4, 48000, 2, 2051, 0, ctypes.byref(err)) is None and err.value == -3
    with pytest.raises(opus_amd.OpusError):
        opus_amd.EncoderBatch(4, channels=2)
    # the product package never imports the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "opus_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle/" not in txt.replace("oracle/oc_", "").replace("generated from oracle", "") or f in ("fx.h", "celt_tables.h"), f
                assert "libcelt_oracle" not in txt and "libopus_ref" not in txt, f
