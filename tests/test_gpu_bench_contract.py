"""bench.py prints exactly one JSON line with the fields the driver reads, plus the roofline
and CPU-baseline objects (small grid so that it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_schema():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--grid", "512", "--cpu-grid", "256", "--cpu-restarts", "1"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert d["metric"] == base["metric"]
    for key in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "GB/s" and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    rl = d["roofline"]
    assert rl["bound"] == "hbm" and rl["unit"] == "GB/s" and rl["peak"] == 8000.0
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"] and cb["host_stream_triad_GBps"] > 0
    assert d["config"]["tol"] == 1e-8 and "value" in d["config"]["spmv_variant"]
    assert d["kernels"]["matmult"]["csr_model_GBps"] >= d["kernels"]["matmult"]["GBps"]


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bench_two_ranks_under_torchrun():
    """The N > 1 code path of bench.py as a launcher starts it (RANK/WORLD_SIZE in the
    environment): barriers, max over ranks, sums over ranks; two ranks share the one GPU of the
    test box over the gloo transport.  --weak: every rank keeps grid^2 rows."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "ex2", "--weak", "--grid", "256", "--transport", "gloo", "--share-device"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    assert d["config"]["rows_per_gpu"] == 256 * 256 and "2 rank(s)" in d["config"]["partition"]
    assert d["config"]["transport"] == "gloo"
    assert "cpu_baseline" not in d            # rank 0 at N = 1 only
    assert d["value"] > 0


def test_bench_starts_its_own_ranks_strong_scaling():
    """`python bench.py --gpus 2` from a plain shell starts the two rank processes itself; the
    default workload on several GPUs is the 3-D Laplacian on a fixed grid cut into slabs of
    planes (strong scaling)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--grid", "64", "--workload", "lap3d", "--transport", "gloo", "--share-device"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 2
    assert d["config"]["workload"].startswith("3-D 7-point Laplacian 64x64x64 (262144 rows)")
    assert d["config"]["rows_per_gpu"] == 32 * 64 * 64 and "planes" in d["config"]["partition"]
    assert "ghost entries received per product: 4096" in d["config"]["spmv_variant"]
    assert d["value"] > 0


def test_bench_eight_ranks_rehearsal():
    """The rank count the scaling run ends at: eight rank processes started by bench.py itself,
    sharing the one GPU of the test box over gloo, on a 64^3 grid (8 planes per rank, every inner
    rank with two neighbours)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1",
                        "--grid", "64", "--transport", "gloo", "--share-device", "--no-kernel-timing"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong"
    assert d["config"]["rows_per_gpu"] == 8 * 64 * 64 and "8 rank(s)" in d["config"]["partition"]
    assert d["solver_state"]["restarts"] == 2 and d["value"] > 0


def test_bench_refuses_a_transport_other_than_rccl():
    """No silent fallback: two ranks on one device cannot form an RCCL communicator, and the
    run must fail rather than print a line measured over host-staged collectives; asking for
    gloo without --share-device is refused as well."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--grid", "32", "--workload", "lap3d", "--share-device"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.decode().splitlines() if l.strip().startswith("{")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--grid", "32", "--workload", "lap3d", "--transport", "gloo"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert r.returncode != 0 and b"rehearsal" in r.stderr


def test_bench_to_convergence_checks_eigenvalues():
    """--to-convergence times the whole EPSSolve and reports the residuals of the converged pairs
    and the distance of every eigenvalue to the analytic spectrum of the Laplacian (1e-10
    relative is the bar for eigenvalues)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "lap3d", "--grid", "48",
                        "--to-convergence", "--no-cpu-baseline"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    d = json.loads([l for l in r.stdout.decode().splitlines() if l.strip().startswith("{")][0])
    assert d["solver_state"]["nconv"] >= 32 and d["solver_state"]["reason"] > 0
    assert d["steps"] == d["solver_state"]["restarts"] and d["time_to_convergence_s"] > 0
    assert d["max_relative_residual"] < 1e-7
    assert d["max_eigenvalue_error_vs_analytic"] < 1e-10


def test_lap3d_192_to_convergence_against_analytic_spectrum():
    """The 3-D 7-point Laplacian on a 192^3 grid (7.1M rows) solved to convergence with the
    benchmark's settings (nev=32, ncv=64, tol 1e-8): every eigenvalue within 1e-10 relative of
    the analytic spectrum (src/eps/examples/tutorials/ex19.c:30-57), residuals of the class of
    the tolerance.  (profiles/ holds the same run on the 256^3 and 512^3 grids.)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "lap3d", "--grid", "192",
                        "--to-convergence", "--no-cpu-baseline"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    d = json.loads([l for l in r.stdout.decode().splitlines() if l.strip().startswith("{")][0])
    assert d["config"]["workload"].startswith("3-D 7-point Laplacian 192x192x192 (7077888 rows)")
    assert d["config"]["nev"] == 32 and d["config"]["ncv"] == 64 and d["config"]["tol"] == 1e-8
    assert d["solver_state"]["nconv"] >= 32 and d["solver_state"]["reason"] > 0
    assert d["max_eigenvalue_error_vs_analytic"] < 1e-10
    assert d["max_relative_residual"] < 10 * 1e-8
    assert d["frac_of_hbm_peak"] > 0.4            # the whole solve, not a fallback path
