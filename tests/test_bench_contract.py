"""bench.py prints ONE JSON line with the keys the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--grid", "24", "--steps", "2",
                          "--warmup", "1", "--cpu-sample-m", "12", "--check"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    assert len(lines[0]) < 4096, len(lines[0])          # (round-5 review: a 25.9 KB line left the driver's record unparsed)
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert os.path.exists(os.path.join(ROOT, "bench_detail.json"))
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "GFLOP/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - d["config"]["fl"] / d["ms_per_step"] / 1e6) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    assert d["residual_2norm"] < 1e-11


@pytest.mark.gpu
def test_bench_two_ranks_over_gloo():
    """The N > 1 launch line of the driver (torch.distributed.run, one rank per GPU);
    on the 1-GPU box the two ranks share GPU 0 and exchange through gloo."""
    from ports import free_port
    port = free_port()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "32", "--steps", "2", "--warmup", "1",
                          "--dist-backend", "gloo", "--check"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["cpu_baseline"] is None
    assert d["exchange"]["allreduce_calls_per_factorization"] > 0
    assert d["residual_2norm"] < 1e-11
    # the invariants of the factor as it lies distributed over the ranks (no gathered copy)
    fc = d["factor_checks_distributed"]
    assert fc["logdet_rel_err"] < 1e-11 and fc["trace_rel_err"] < 1e-11, fc
    assert fc["upper_nonzeros"] == 0 and fc["nonfinite"] == 0 and fc["nonpositive_diag"] == 0, fc


@pytest.mark.gpu
def test_bench_keeps_its_line_when_the_gathered_factor_does_not_fit():
    """A rank that has no room for the gathered factor (two ranks at the headline size hold 181.6 + 117 GB with 8 GB to
    spare): the line must still be printed -- without the residual, with the note and the distributed invariants.  Here rank 1
    is told that it has no room (test hook)."""
    from ports import free_port
    port = free_port()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "24", "--steps", "1", "--warmup", "1",
                          "--dist-backend", "gloo", "--check"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, CHOLMOD_HIP_TEST_FAIL_GATHER="1", SSAMD_TEST_HOOKS_LIB="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "residual_2norm" not in d and "skipped" in d["residual_2norm_note"]
    assert d["factor_checks_distributed"]["logdet_rel_err"] < 1e-11


@pytest.mark.gpu
def test_bench_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's 1-GPU command line with N = 2):
    bench.py starts its two ranks itself and rank 0 prints the one line with n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--grid", "32",
                          "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong"
    assert d["residual_2norm"] < 1e-11
    assert d["factor_checks_distributed"]["logdet_rel_err"] < 1e-11


@pytest.mark.gpu
def test_bench_watchdog_turns_a_hung_collective_into_an_error_line():
    """First hardware contact must not be able to fail silently (round-4 review, item 1b): rank 1 of 2 stops issuing its
    collectives in the middle of the second timed step (test hook: its host thread sleeps before exchange 3; nothing hangs
    on the device), rank 0 waits for it in the exchange.  The per-step deadline expires: rank 0 prints ONE JSON line with
    "error", the phase, the completed step's time and the exchange its device has entered and not left, and the job exits
    non-zero -- instead of sitting there until the driver's 1800 s kill."""
    from ports import free_port
    port = free_port()
    env = dict(os.environ, SSAMD_TEST_HOOKS_LIB="1", CHOLMOD_HIP_TEST_HANG_EXCHANGE="1:3:3")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "32", "--steps", "2", "--warmup", "1",
                          "--dist-backend", "gloo", "--step-deadline", "15"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode != 0, out.stdout[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (out.stdout, out.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["value"] is None and "deadline" in d["error"] and d["n_gpus"] == 2, d
    assert d["phase"] == "timed step 2 of 2" and d["steps_completed"] == 1 and d["last_completed_step_ms"] > 0, d
    pg = d["progress"]
    assert pg["exchange_entered_on_device"] == 3 and pg["exchange_left_on_device"] == 2, pg
    assert pg["pending_exchange"]["sequence"] == 3 and pg["pending_exchange"]["rank_group"] == [0, 1], pg


def test_bench_gpus_flag_fails_loudly_without_the_devices():
    """More ranks than visible devices over RCCL: an error message and a non-zero exit code, not a run on fewer GPUs
    (this container has no GPU at all; a 1-GPU box has fewer than 8)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--grid", "16"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0
    assert "needs 64 visible HIP devices" in out.stderr, out.stderr[-2000:]
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    # under a launcher --gpus must agree with WORLD_SIZE
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "16"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and "must agree" in out.stderr


@pytest.mark.gpu
def test_bench_exchange_selftest_at_100_cubed_over_rccl():
    """The multi-GPU path at a BASELINE size on the one GPU there is: CHOLMOD_HIP_SHARE_AS_WORLD=8 marks the fronts an 8-rank
    run of Poisson 100^3 would share (the 10 000-column root among them: 1024 / 2048-wide outer blocks, several windows,
    negative window bases), the engine's native exchange runs over the real RCCL with its one rank (reduce-scatter, D
    broadcast skipped at g = 1, all-gather, agreement), the shared fronts' chain through k_chainf -- and the factor must
    pass the size-independent checks: residual, closed-form log det, dead triangles, finite entries."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CHOLMOD_HIP_SHARE_AS_WORLD"] = "8"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--grid", "100", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["exchange"]["self_test_share_as_world"] == 8
    assert d["exchange"]["allreduce_calls_per_factorization"] > 50
    assert d["residual_2norm"] < 1e-11
    fc = d["factor_checks"]
    assert fc["logdet_rel_err"] < 1e-11 and fc["upper_nonzeros"] == 0 and fc["nonfinite"] == 0 and fc["nonpositive_diag"] == 0, fc


@pytest.mark.gpu
@pytest.mark.slow
@pytest.mark.parametrize("grid,extra", [(200, {"CHOLMOD_HIP_NO_CB_PASSTHROUGH": "1"}), (160, {})])
def test_bench_exchange_selftest_at_the_headline_size_over_rccl(grid, extra):
    """The same self test at the metric's own configuration (Poisson 200^3, 8 M dof, L = 181.6 GB): CHOLMOD_HIP_SHARE_AS_WORLD=8
    marks the 14 fronts an 8-rank run shares (the 64 849-column root among them: 4096-wide outer blocks, windows, the 8 / 4 / 2
    group structure), every block column goes through pack -> ncclReduceScatter -> unpack -> chain (k_chainf) -> pack ->
    ncclAllGather -> unpack over the real RCCL with its one rank, ahead of time on the exchange stream where the schedule says
    so -- and the factor must pass the size-independent checks.  One timed step each.
    ONE rank holds what eight would share out: with the default layout (contributions routed past the shared fronts'
    blocks: a contributor's block lives until the root of its tree is factored) the single rank keeps every such block of
    the whole factorization, 230 GB of arena at 200^3 -- so 200^3 runs the layout of the first half of round 4 (full squares
    of partial sums, pulled level by level: 98 GB, CHOLMOD_HIP_NO_CB_PASSTHROUGH=1, as profiles/r04y_* did) and the default
    layout runs at 160^3 (L 87 GB + arena 99 GB)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CHOLMOD_HIP_SHARE_AS_WORLD"] = "8"
    env.update(extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--grid", str(grid), "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline", "--no-secondary", "--no-profile-pass"], capture_output=True, text=True, timeout=1500,
                         cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["config"]["n"] == grid ** 3 and d["n_gpus"] == 1 and d["exchange"]["self_test_share_as_world"] == 8, d["config"]
    assert d["exchange"]["allreduce_calls_per_factorization"] > 150
    assert d["residual_2norm"] < 1e-11
    fc = d["factor_checks"]
    assert fc["logdet_rel_err"] < 1e-10 and fc["upper_nonzeros"] == 0 and fc["nonfinite"] == 0 and fc["nonpositive_diag"] == 0, fc
