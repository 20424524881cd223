"""DIAGNOSTIC: run one DEMO_MODES case of tests/test_zz_reference_programs.py with opus_demo linked to <flavour>, under tools/api_trace_shim.c, and leave
<out>/<flavour>.log (one line per opus_encode24 / opus_decode24 call), <out>/<flavour>.dump (packets and PCM) and <out>/<flavour>.pcm behind.
   python tools/demo_trace.py <case> <flavour: gpu|emu|ref> <out dir> [seconds]"""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_kernel_emu_silkdec import speechy
from test_zz_reference_programs import DEMO_MODES
case, fl, out = int(sys.argv[1]), sys.argv[2], sys.argv[3]
seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
Fs, ch, args = DEMO_MODES[case]
os.makedirs(out, exist_ok=True)
shim = os.path.join(ROOT, "tools/_trace.so")
if not os.path.exists(shim): subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools/api_trace_shim.c"), "-o", shim, "-ldl"])
n = int(Fs * seconds)
sig = np.ascontiguousarray(speechy(n * (48000 // Fs) // 960 + 2, ch, len(args), 960)[::48000 // Fs][:n]).astype("<i2")
pcm = os.path.join(out, "in.pcm"); sig.tofile(pcm)
env = dict(os.environ, LD_PRELOAD=shim, OPUS_TRACE_FILE=os.path.join(out, fl + ".log"), OPUS_TRACE_DUMP=os.path.join(out, fl + ".dump"), OPUS_TRACE_CH=str(ch))
p = subprocess.run([os.path.join(ROOT, "oracle/_ref/reftests", fl, "opus_demo")] + list(args) + [pcm, os.path.join(out, fl + ".pcm")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=900)
print(fl, "rc", p.returncode, p.stdout.decode(errors="replace")[-300:])
