"""tests/hostemu.py — TEST INFRASTRUCTURE: builds tests/emu/libopus_amd_emu.so (the product's whole C ABI on the CPU wave emulator) and the
reference's unmodified C test programs against either that library (CPU, here) or the real opus_amd/libopus_amd.so (MI355X)."""
import os, subprocess, fcntl
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, "tests/emu/libopus_amd_emu.so")
REF = "/root/reference"

def build_emu_lib(opt="-O1"):
    srcs = [os.path.join(ROOT, "tests/emu", f) for f in ("emu_host.cpp", "wave_emu.cpp")]
    csrc = os.path.join(ROOT, "opus_amd/csrc")
    deps = srcs + [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(ROOT, "tests/emu", f) for f in ("wave_emu.h", "hip_stub.h")] + [os.path.join(ROOT, "include/opus_amd.h")]
    with open(EMU_SO + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not os.path.exists(EMU_SO) or os.path.getmtime(EMU_SO) < max(os.path.getmtime(p) for p in deps):
            subprocess.check_call(["g++", opt, "-g", "-std=c++17", "-fPIC", "-shared", "-rdynamic", "-fvisibility=hidden", "-Wno-attributes",
                                   "-I" + os.path.join(ROOT, "tests/emu"), "-I" + csrc, "-I" + os.path.join(ROOT, "include")] + srcs + ["-o", EMU_SO + ".tmp", "-lpthread"])
            os.replace(EMU_SO + ".tmp", EMU_SO)
    return EMU_SO

REFTEST_FLAGS = ["-O2", "-std=gnu99", "-w", "-DOPUS_BUILD", "-DVAR_ARRAYS", "-DHAVE_LRINT", "-DHAVE_LRINTF", "-DFIXED_POINT=1", "-DDISABLE_FLOAT_API",
                 '-DPACKAGE_VERSION="reference"', "-I" + REF + "/include", "-I" + REF + "/celt", "-I" + REF + "/silk", "-I" + REF + "/src", "-I" + REF]
REFTESTS = {"test_opus_api": ["tests/test_opus_api.c"], "test_opus_decode": ["tests/test_opus_decode.c"], "test_opus_padding": ["tests/test_opus_padding.c"],
            "test_opus_encode": ["tests/test_opus_encode.c", "tests/opus_encode_regressions.c"], "opus_demo": ["src/opus_demo.c"],
            # its first section exercises the reference's internal mapping_matrix_* helpers, which no libopus exports: that one source file is compiled into the program
            "test_opus_projection": ["tests/test_opus_projection.c", "src/mapping_matrix.c"]}

def build_reftests(flavour):
    """Compile the reference's UNMODIFIED test programs (sources where they lie under /root/reference) against the emulated library (flavour 'emu')
    or the product library (flavour 'gpu', rpath relative so that the binaries travel to the GPU box).  Outputs under oracle/_ref/reftests/."""
    out = os.path.join(ROOT, "oracle/_ref/reftests", flavour); os.makedirs(out, exist_ok=True)
    if flavour == "emu": lib = ["-L" + os.path.join(ROOT, "tests/emu"), "-lopus_amd_emu", "-Wl,-rpath," + os.path.join(ROOT, "tests/emu")]; build_emu_lib()
    elif flavour == "ref": lib = ["-L" + os.path.join(ROOT, "oracle/_ref"), "-l:libopus_ref_fxa.so", "-Wl,-rpath,$ORIGIN/../.."]         # the reference's own library (fixed-point, float API on: the default build): the comparison side of the opus_demo tests
    else: lib = ["-L" + os.path.join(ROOT, "opus_amd"), "-lopus_amd", "-Wl,-rpath,$ORIGIN/../../../../opus_amd"]
    if not os.path.isdir(REF): return out
    for name, srcs in REFTESTS.items():
        if flavour == "ref" and name != "opus_demo": continue
        exe = os.path.join(out, name)
        s = [os.path.join(REF, x) for x in srcs]
        extra = []
        if name == "opus_demo":                # its rand() must not be shared with the ROCm runtime's threads: tests/emu/demo_rand.c
            s.append(os.path.join(ROOT, "tests/emu/demo_rand.c")); extra = ["-Drand=oa_demo_rand", "-Dsrand=oa_demo_srand"]
        if os.path.exists(exe) and os.path.getmtime(exe) >= max(os.path.getmtime(x) for x in s): continue
        subprocess.check_call(["gcc"] + REFTEST_FLAGS + extra + s + ["-o", exe] + lib + ["-lm"])
    if flavour != "ref":
        # test_opus_api once more WITHOUT -DDISABLE_FLOAT_API: the library models the reference's default build (float API on), so the program's float-entry checks
        # (opus_decode_float / opus_encode_float / multistream float paths) run too
        exe = os.path.join(out, "test_opus_api_fl"); s = [os.path.join(REF, x) for x in REFTESTS["test_opus_api"]]
        if not (os.path.exists(exe) and os.path.getmtime(exe) >= max(os.path.getmtime(x) for x in s)):
            subprocess.check_call(["gcc"] + [f for f in REFTEST_FLAGS if f != "-DDISABLE_FLOAT_API"] + s + ["-o", exe] + lib + ["-lm"])
    return out

def build_trace_shim():
    """tools/encode_trace_shim.c -> oracle/_ref/enc_trace_shim.so (LD_PRELOAD logger of the encode calls an unmodified program makes; travels to the GPU box)"""
    so = os.path.join(ROOT, "oracle/_ref/enc_trace_shim.so"); src = os.path.join(ROOT, "tools/encode_trace_shim.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", src, "-o", so, "-ldl"])
    return so

if __name__ == "__main__":
    import sys
    print(build_reftests(sys.argv[1] if len(sys.argv) > 1 else "emu"))
