"""tools/fuzz_sweep.py — TEST INFRASTRUCTURE: fresh-seed sweeps of the settings fuzzers (tests/test_hostemu_fuzz.py) against the compiled reference, one process per job
(an abort of the wave emulator's LDS watch ends only that job), on the CPU wave emulator here or on the MI355X (--which gpu).

  python tools/fuzz_sweep.py --fuzzers fuzz,fuzz_sparse,fuzz_batch --first 8800000 --count 3000 --pipeline 1 --workers 6 --log profiles/r05_fuzz_pipeline.log

Every failing job is logged with the tail of its output; the summary line is `jobs J, failed F`.  --pipeline N = OPUS_AMD_SET_KERNEL_PIPELINE(N) on every encoder under
test (1: every 10 / 20 ms call through the front / quantiser / back kernels)."""
import argparse, os, subprocess, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

JOB = """
import sys
sys.path.insert(0, %r)
import test_hostemu_fuzz as t
t.WHICH = %r
getattr(t, %r)(%d)
"""

def run(job):
    fz, seed, which, pipeline, timeout = job
    env = dict(os.environ, OPUS_AMD_TEST_PIPELINE=str(pipeline), OPUS_AMD_TEST_TRPRE=os.environ.get("OPUS_AMD_TEST_TRPRE", "-1"))
    t0 = time.time()
    try:
        p = subprocess.run([sys.executable, "-c", JOB % (os.path.join(ROOT, "tests"), which, fz, seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, cwd=ROOT, timeout=timeout)
        rc, out = p.returncode, p.stdout.decode(errors="replace")
    except subprocess.TimeoutExpired as e:
        rc, out = -999, "TIMEOUT\n" + (e.stdout or b"").decode(errors="replace")
    return fz, seed, rc, out, time.time() - t0

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fuzzers", default="fuzz,fuzz_sparse,fuzz_batch")
    ap.add_argument("--first", type=int, default=8800000); ap.add_argument("--count", type=int, default=100)
    ap.add_argument("--pipeline", type=int, default=-1); ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--rerun-failures-of", default=None, help="a previous sweep's log: run exactly the jobs it logged as FAIL")
    ap.add_argument("--which", default="emu"); ap.add_argument("--log", default=None); ap.add_argument("--timeout", type=int, default=1800)
    a = ap.parse_args()
    if a.rerun_failures_of:
        # the (fuzzer, seed) pairs a previous sweep logged as FAIL (jobs that died while the sources were being edited under a running sweep, or real finds since fixed), once more
        pairs = sorted({(l.split()[1], int(l.split()[2])) for l in open(a.rerun_failures_of) if l.startswith("FAIL ")})
        jobs = [(fz, seed, a.which, a.pipeline, a.timeout) for fz, seed in pairs]
    else:
        jobs = [(fz, a.first + i, a.which, a.pipeline, a.timeout) for i in range(a.count) for fz in a.fuzzers.split(",")]
    log = open(a.log, "a") if a.log else None
    def say(s):
        print(s, flush=True)
        if log: log.write(s + "\n"); log.flush()
    say("# fuzz_sweep: %s seeds %d..%d pipeline=%d which=%s (%d jobs)" % (a.fuzzers, a.first, a.first + a.count - 1, a.pipeline, a.which, len(jobs)))
    done = failed = 0; t0 = time.time()
    with ThreadPoolExecutor(a.workers) as ex:
        for fz, seed, rc, out, dt in ex.map(run, jobs):
            done += 1
            if rc != 0:
                failed += 1
                say("FAIL %s %d rc=%d (%.0f s)\n%s" % (fz, seed, rc, dt, "\n".join(out.strip().splitlines()[-6:])))
            if done % 50 == 0: say("# %d/%d jobs, %d failed, %.0f s" % (done, len(jobs), failed, time.time() - t0))
    say("# jobs %d, failed %d, %.0f s" % (done, failed, time.time() - t0))
    return 1 if failed else 0

if __name__ == "__main__":
    sys.exit(main())
