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

def _run_ranks(n, extra, backend):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend: env["OPUS_AMD_BENCH_BACKEND"] = backend
    else: env.pop("OPUS_AMD_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + extra
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    return _last_json(p.stdout.decode(errors="replace"))

def _check_ranks(r, n, S, K):
    assert r["n_gpus"] == n and r["steps"] == K and r["scaling"] == "weak" and r["value"] > 0
    assert r["config"]["frames_per_step"] == n * S and r["config"]["all_packets_valid"]
    assert r["ranks_seen"] == n and sorted(x["rank"] for x in r["per_rank"]) == list(range(n)) and all(x["ms_per_step"] > 0 for x in r["per_rank"])

def test_bench_two_ranks_one_gpu():
    """the N > 1 path end to end on one GPU (gloo stands in for RCCL), with and without the final gather: the gather of step t is issued on a side stream / from the host
    while step t + 1 encodes, so it must not show in the step time beyond noise (two processes time-slice ONE GPU here, hence the generous bound)"""
    S, K = 4096, 6
    r = _run_ranks(2, ["--steps", str(K), "--warmup", "2", "--streams", str(S)], "gloo")
    _check_ranks(r, 2, S, K)
    g = r["gather"]
    assert g["steps"] == K + 2 and not g["overflow"] and g["cap_bytes_per_stream"] < g["slot_bytes_per_stream"]
    r0 = _run_ranks(2, ["--steps", str(K), "--warmup", "2", "--streams", str(S), "--no-gather"], "gloo")
    _check_ranks(r0, 2, S, K)
    assert "gather" not in r0
    print("ms_per_step with the gather %.3f, without %.3f" % (r["ms_per_step"], r0["ms_per_step"]))
    assert r["ms_per_step"] < 1.5 * r0["ms_per_step"] + 2.0

@pytest.mark.parametrize("n,S", [(2, 4096), (8, 4096)])
def test_bench_ranks_one_gpu_p2p_gather(n, S):
    """the gather's second transport (--gather p2p: every rank copies its wire record into rank 0's double buffer through an IPC mapping, no collective, control traffic over
    gloo) executed for real: n processes on GPU 0, IPC handles opened across them, device-to-device copies on side streams, the sticky overflow flag reduced over the ranks;
    world size 8 at 8 x 4,096 streams is the shape of the driver's scaling run"""
    K = 3
    r = _run_ranks(n, ["--steps", str(K), "--warmup", "1", "--streams", str(S), "--gather", "p2p"], "gloo")
    _check_ranks(r, n, S, K)
    g = r["gather"]
    assert g["steps"] == K + 1 and not g["overflow"] and g["transport"].startswith("p2p") and g["cap_bytes_per_stream"] == 2 * 320 + 64

def test_bench_ranks_gather_falls_back_to_p2p_when_the_collective_fails():
    """--gather auto (the default): the RCCL gather's warm-up fails on every rank (test hook) -> all ranks switch to the point-to-point transport, the exchange stays inside
    the timed region and the line says so; 8 ranks on the one GPU = the shape of the driver's scaling run"""
    S, K = 4096, 3
    env_was = os.environ.get("OPUS_AMD_BENCH_FAIL_RCCL")
    os.environ["OPUS_AMD_BENCH_FAIL_RCCL"] = "1"
    try: r = _run_ranks(8, ["--steps", str(K), "--warmup", "1", "--streams", str(S)], "gloo")
    finally:
        if env_was is None: os.environ.pop("OPUS_AMD_BENCH_FAIL_RCCL", None)
        else: os.environ["OPUS_AMD_BENCH_FAIL_RCCL"] = env_was
    _check_ranks(r, 8, S, K)
    g = r["gather"]
    assert g["in_timed_region"] and g["transport"].startswith("p2p") and "rccl" in g["fallback_from"] and g["steps"] == K + 1 and not g["overflow"]
    assert "p2p" in r["config"]["parallelism"]

def test_bench_two_ranks_rccl():
    """two ranks, two GPUs, RCCL: only where the box has them (the driver's scaling run is the real measurement)"""
    import torch
    if torch.cuda.device_count() < 2: pytest.skip("one GPU on this box")
    S, K = 8192, 4
    r = _run_ranks(2, ["--steps", str(K), "--warmup", "2", "--streams", str(S)], None)
    _check_ranks(r, 2, S, K)
    assert sorted(x["device"] for x in r["per_rank"]) == [0, 1] and not r["gather"]["overflow"] and r["gather"]["transport"].startswith("RCCL")

def test_bench_single_rank_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--streams", "4096", "--no-extra-configs", "--frames-per-launch", "3"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    r = _last_json(p.stdout.decode(errors="replace"))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["roofline"]["bound"] == "hbm" and 0 < r["roofline"]["frac"] < 1 and r["roofline"]["peak_measured"] > 1000
    assert r["cpu_baseline"]["kind"] == "reference" and r["cpu_baseline"]["cores"] == 1 and r["cpu_baseline"]["host_nproc"] >= 1
    assert r["frames_per_launch"]["T"] == 3 and r["frames_per_launch"]["frames_per_s"] > 0 and r["frames_per_launch"]["equals_step_by_step"]
    assert r["config"]["parity_sample_ok"] is True and r["config"]["parity_sample"]["frames"] == 64 * 4 and r["config"]["lib_matches_sources"] is True

def test_bench_default_line_has_every_configuration():
    """the default N = 1 line (what the driver records) at a reduced stream count: configs 3 / 4 / 5 and the decoder legs, each with value, roofline and parity sample"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--streams", "4096", "--frames-per-launch", "6", "--steady-state", "40"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200, cwd=ROOT)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    r = _last_json(p.stdout.decode(errors="replace"))
    assert set(r["configs"]) == {"decode_2", "config_3", "decode_3", "config_4", "decode_4", "config_5", "config_3_fec", "config_3_60ms"}
    for k, e in r["configs"].items():
        assert e["value"] > 0 and e["valid"] and e["roofline"]["kernel_ms"] > 0, k
        assert e["parity_ok"] is True and e["cpu"]["value"] > 0, k                       # every leg, config 5 included (opus_multistream_encode of the reference on two encoders)
    assert r["roofline"]["dominant"]["kernel"] in ("oa_encode_kernel", "oa_celt_pvq_kernel") and r["roofline"]["dominant"]["ms"] > 0      # HIP events between the launches, inside the library (4,096 streams fit one round of the chip: one kernel)
    ss = r["steady_state"]
    assert ss["consecutive_frames"] == 40 and ss["parity_sample_ok"] is True and ss["parity_frames"] == 4 * 40 and ss["all_packets_valid"]
    assert ss["full_width"]["value"] > 0 and ss["full_width"]["all_packets_valid"] and ss["full_width"]["replicas_agree"]
    assert len(p.stdout.decode(errors="replace").strip().splitlines()[-1]) < 8000          # the driver keeps the last 8 KB of the line: every leg must be in it
    assert all(k in r["notes"] for k in ("legs", "roofline", "cpu", "parity", "steady_state"))
