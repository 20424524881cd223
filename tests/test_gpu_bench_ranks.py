"""MI355X (1 GPU): bench.py's N > 1 code path executed end to end with two ranks -- torch.distributed.run, rank / world bookkeeping, stream sharding, barrier + max-over-ranks
timing, the compacted final gather of opus_amd/shard.py (device packing kernel, table collective, point-to-point payload) -- with both ranks on GPU 0 and the gloo
backend standing in for RCCL (two RCCL ranks cannot share one device); and the N = 1 line's contract fields."""
import json, os, subprocess, sys, socket, pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p

def _last_json(txt):
    for line in reversed(txt.strip().splitlines()):
        if line.startswith("{"): return json.loads(line)
    raise AssertionError(txt[-2000:])

def test_bench_two_ranks_one_gpu():
    env = dict(os.environ, OPUS_AMD_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--streams", "2048"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    r = _last_json(p.stdout.decode(errors="replace"))
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak" and r["value"] > 0
    assert r["config"]["frames_per_step"] == 2 * 2048 and r["config"]["all_packets_valid"]

def test_bench_single_rank_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--streams", "4096", "--no-extra-configs", "--frames-per-launch", "3"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    r = _last_json(p.stdout.decode(errors="replace"))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["roofline"]["bound"] == "hbm" and 0 < r["roofline"]["frac"] < 1 and r["roofline"]["peak_measured"] > 1000
    assert r["cpu_baseline"]["kind"] == "reference" and r["cpu_baseline"]["cores"] == 1 and r["cpu_baseline"]["host_nproc"] >= 1
    assert r["frames_per_launch"]["T"] == 3 and r["frames_per_launch"]["frames_per_s"] > 0
