#!/usr/bin/env python3
"""Benchmark of the Krylov-Schur hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A *step* is one Krylov-Schur restart of ``EPSSolve`` (the outer iteration,
``eps->its++``): the Lanczos expansion of the basis from the kept Ritz vectors to
ncv columns (one CSR SpMV and one CGS-with-refinement orthogonalisation per
column), the projected eigenproblem on the host, and the restart ``V <- V Q``.

Workload at N = 1: BASELINE.json configs[1] -- ex2's 2-D 5-point Laplacian on a
4096 x 4096 grid (16.7M rows), symmetric Krylov-Schur, nev = 32 (ncv = 64).  With N
GPUs the grid grows to 4096 x (4096 N) rows split in contiguous row blocks (weak
scaling: every GPU keeps 16.7M rows); reductions and the SpMV halo exchange run on
RCCL inside libksgpu.

``value`` is the achieved HBM traffic rate of the whole job: algorithmic bytes of
the K timed restarts (the byte model of the op table, summed over all ranks)
divided by the wall time of those restarts (max over ranks).  W restarts run
untimed first (the first restart builds the full 64-column basis; later restarts
are the steady state).  The JSON line also carries the roofline of the dominant
kernel (HIP-event time of the launches inside the timed region) and a CPU
baseline: the oracle, i.e. the CPU restatement of the reference path, timed on the
host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0


def cpu_baseline(grid, nev, restarts, threads):
    """The oracle (CPU restatement of the reference path) on a bounded sample:
    `restarts` steady-state restarts of the same solver on a grid x grid Laplacian."""
    from threadpoolctl import threadpool_limits
    from scipy.sparse import csr_matrix
    from oracle import eps as oeps
    from slepc_amd import matgen
    rp, ci, va = matgen.laplacian2d(grid)
    A = csr_matrix((va, ci.astype(np.int32), rp.astype(np.int32)), shape=(grid * grid, grid * grid))
    n = grid * grid
    spmv_bytes = 12.0 * rp[-1] + 8.0 * (n + 1) + 16.0 * n
    marks = []
    state = {"mv": 0}

    def mv(x, y):
        state["mv"] += 1
        y[:] = A @ x

    def monitor(its, nconv, eigr, eigi, errest, nv):
        marks.append((time.perf_counter(), e.V.bytes + state["mv"] * spmv_bytes))

    with threadpool_limits(limits=threads):
        e = oeps.EPS(mv, n, hermitian=True, nev=nev, max_it=1 + restarts, monitor=monitor)
        e.solve()
    (t0, b0), (t1, b1) = marks[0], marks[-1]
    return {"value": (b1 - b0) / (t1 - t0) / 1e9, "unit": "GB/s", "cores": threads, "kind": "port",
            "ms_per_step": 1e3 * (t1 - t0) / restarts,
            "sample": "%d steady-state restarts (nev=%d, ncv=%d) of the oracle (numpy/OpenBLAS gemv + "
                      "scipy CSR) on a %dx%d 2-D Laplacian (%d rows)" % (restarts, nev, e.ncv, grid, grid, n)}


def analytic_laplacian_top(dims, count):
    """The `count` largest eigenvalues (with multiplicity) of the 5/7-point Laplacian on a grid
    with `dims` points per direction: sums of 2 - 2 cos(k pi / (n + 1)), one term per direction
    (src/eps/examples/tutorials/ex19.c:30-57 gives the 3-D formula)."""
    tops = [2.0 - 2.0 * np.cos(np.arange(n, max(0, n - count), -1) * np.pi / (n + 1)) for n in dims]
    acc = tops[0]
    for t in tops[1:]:
        acc = np.sort((acc[:, None] + t[None, :]).ravel())[::-1][:count]
    return np.sort(acc)[::-1][:count]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=4096, help="grid columns; each GPU owns grid x grid rows")
    ap.add_argument("--nev", type=int, default=32)
    ap.add_argument("--eps-type", default="krylovschur", choices=["krylovschur", "arnoldi", "lanczos"],
                    help="driver on the same kernels (default: the Krylov-Schur path of the metric)")
    ap.add_argument("--workload", default="ex2", choices=["ex2", "lap3d", "markov", "rmat"],
                    help="ex2/lap3d: --grid points per side and GPU (weak scaling); markov: --grid is m of "
                         "ex5 (nonsymmetric, largest real); rmat: --grid is the R-MAT scale (graph Laplacian); "
                         "the last two split one fixed matrix over the ranks")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "gloo"])
    ap.add_argument("--mv-grid", type=int, default=0, help="tuning: KSG_OPT_MV_GRID")
    ap.add_argument("--dot-grid", type=int, default=0, help="tuning: KSG_OPT_DOT_GRID")
    ap.add_argument("--no-lazy", action="store_true", help="tuning: KSG_OPT_LAZY_UPDATE off")
    ap.add_argument("--no-fuse", action="store_true", help="tuning: KSG_OPT_FUSE_DOTS off")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank opens device 0 (RCCL refuses two ranks on one "
                         "device, so the run ends up on the gloo transport)")
    ap.add_argument("--cpu-grid", type=int, default=2048)
    ap.add_argument("--cpu-restarts", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--to-convergence", action="store_true",
                    help="time the whole EPSSolve until nev eigenpairs converge (or --max-restarts); "
                         "--steps/--warmup are ignored and `steps` reports the restarts it took")
    ap.add_argument("--max-restarts", type=int, default=20000)
    ap.add_argument("--tol", type=float, default=0.0, help="0 = the default 1e-8")
    args = ap.parse_args()

    from slepc_amd import dist as kdist, ksgpu, kseps as K, matgen
    rank, local_rank, size = kdist.env_rank()
    if size != args.gpus:
        if size == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = size
    dist = kdist.init_process_group() if size > 1 else None

    ctx = ksgpu.Context(0 if args.share_device else local_rank)
    if size > 1:
        _, _, args.transport = kdist.init_comm(ctx, transport=args.transport)
    if args.mv_grid:
        ctx.set_option(ksgpu.OPT_MV_GRID, args.mv_grid)
    if args.dot_grid:
        ctx.set_option(ksgpu.OPT_DOT_GRID, args.dot_grid)
    if args.no_lazy:
        ctx.set_option(ksgpu.OPT_LAZY_UPDATE, 0)
    if args.no_fuse:
        ctx.set_option(ksgpu.OPT_FUSE_DOTS, 0)

    # ---- operator: rows [row0,row1) of the global matrix ----------------------------
    g = args.grid
    t_gen = time.perf_counter()
    if args.workload == "ex2":
        gx, gy = g, g * size
        N = gx * gy
        row0, row1 = rank * g * g, (rank + 1) * g * g
        rp, ci, va = matgen.laplacian2d(gx, gy, row0, row1)
        wname = "ex2 2-D 5-point Laplacian %dx%d (%d rows), symmetric Krylov-Schur nev=%d" % (gx, gy, N, args.nev)
    elif args.workload == "lap3d":
        gz = g * size
        N = g * g * gz
        row0, row1 = rank * g * g * g, (rank + 1) * g * g * g
        rp, ci, va = matgen.laplacian3d(g, g, gz, row0, row1)
        wname = "3-D 7-point Laplacian %dx%dx%d (%d rows), symmetric Krylov-Schur nev=%d" % (g, g, gz, N, args.nev)
    elif args.workload == "markov":
        N = g * (g + 1) // 2
        row0, row1 = matgen.split_rows(N, size, rank)
        rp, ci, va = matgen.markov(g, row0, row1)
        wname = "ex5 Markov chain m=%d (%d rows), nonsymmetric Krylov-Schur largest real nev=%d" % (g, N, args.nev)
    else:
        N = 1 << g
        row0, row1 = matgen.split_rows(N, size, rank)
        rp, ci, va = matgen.rmat_laplacian(g, row0=row0, row1=row1, progress=(lambda t: print(t, file=sys.stderr, flush=True)) if rank == 0 and g > 23 else None)
        wname = "R-MAT scale %d graph Laplacian (%d rows), symmetric Krylov-Schur nev=%d" % (g, N, args.nev)
    if rank == 0:
        print("operator: %s, %d local rows, %d local entries, generated in %.1f s" % (
            wname, row1 - row0, int(rp[-1]), time.perf_counter() - t_gen), file=sys.stderr, flush=True)
    A = K.Mat.create_aij_hip(ctx, rp, ci, va, N=N, row0=row0)
    nnz_local = int(rp[-1])
    del rp, ci, va

    W, Ksteps = args.warmup, args.steps
    if args.to_convergence:
        W, Ksteps = 0, args.max_restarts
    eps = K.EPS(ctx.h)
    eps.set_operators(A)
    if args.workload == "markov":
        eps.set_problem_type(K.EPS_NHEP)
        eps.set_which(K.EPS_LARGEST_REAL)
    else:
        eps.set_problem_type(K.EPS_HEP)
    if args.eps_type != "krylovschur":
        eps.set_type(args.eps_type)
        wname = wname.replace("Krylov-Schur", "explicitly restarted " + args.eps_type.capitalize())
    eps.set_dimensions(args.nev)
    eps.set_tolerances(args.tol, W + Ksteps)     # default tol 1e-8; stop after W+K restarts
    eps.set_bv_type("hip")

    marks = {}
    # Profiler markers: a 3-element k_vec_set (one workgroup) is launched right before
    # and right after the timed region, so that a rocprofv3 trace of this command can be
    # cut to exactly the launches the JSON line reports (tools/prof_summary.py).
    marker = ctx.vec_alloc(8)

    def fence():
        ctx.sync()
        if dist:
            dist.barrier()
        ctx.sync()
        return time.perf_counter()

    def end_timed():
        marks["t1"] = fence()
        marks["stats"] = ctx.stats()
        marks["c1"] = eps.counters()
        ctx.enable_timing(False)
        ctx.vec_set(marker, 1.0, 3)

    def monitor(its, nconv, eigr, eigi, errest, nest):
        if its == W:
            fence()
            ctx.vec_set(marker, 0.0, 3)
            ctx.stats_reset()
            if not args.no_kernel_timing:
                ctx.enable_timing(True)
            marks["t0"] = fence()
            marks["c0"] = eps.counters()
        elif its == W + Ksteps:
            end_timed()
        if args.to_convergence and rank == 0 and its % 50 == 0:
            print("restart %d: nconv %d, first unconverged %.12g errest %.3e, %.1f s" % (
                its, nconv, float(eigr[min(nconv, nest - 1)]), float(errest[min(nconv, nest - 1)]),
                time.perf_counter() - marks.get("t0", time.perf_counter())), file=sys.stderr, flush=True)
        marks["last"] = (its, nconv, float(eigr[0]), float(errest[0]))

    eps.set_monitor(monitor)
    if W == 0:
        monitor(0, 0, np.zeros(1), np.zeros(1), np.zeros(1), 1)
    eps.solve()
    if args.to_convergence and "t1" not in marks:
        # converged before the restart bound: the timed region ends with the solve
        Ksteps = eps.its
        end_timed()
    if "t1" not in marks:
        raise SystemExit("the solver stopped after %d restarts, before the %d requested" % (eps.its, W + Ksteps))

    dt = marks["t1"] - marks["t0"]
    st = marks["stats"]
    nbytes = sum(v["bytes"] for v in st.values())
    # what the reference's op sequence would have moved: there every dotvec is a pass over
    # the basis, 8 n (k+1) bytes, also the ones the fused multvec answered here for free
    fz = st["dotvec_fused"]
    unfused = nbytes + 4.0 * fz["flops"] + 8.0 * (row1 - row0) * fz["calls"]
    if dist:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
        b = torch.tensor([nbytes, unfused], dtype=torch.float64)
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
        nbytes, unfused = float(b[0]), float(b[1])
    value = nbytes / dt / 1e9

    max_res = eig_err = None
    if args.to_convergence and eps.nconv > 0:
        # ||A x - lambda x|| / |lambda| of every converged pair (collective: all ranks take part)
        max_res = max(eps.compute_error(i) for i in range(min(eps.nconv, args.nev)))
        if args.workload in ("ex2", "lap3d"):
            # every computed eigenvalue against the nearest one of the analytic spectrum (multiple
            # eigenvalues are found with a multiplicity that depends on the start vector)
            dims = (gx, gy) if args.workload == "ex2" else (g, g, gz)
            ana = analytic_laplacian_top(dims, 8 * args.nev)
            got = np.array([eps.get_eigenvalue(i)[0] for i in range(min(eps.nconv, args.nev))])
            eig_err = float(np.max(np.min(np.abs(got[:, None] - ana[None, :]), axis=1) / np.abs(got)))
    if rank == 0:
        kernels = {}
        for ev, v in st.items():
            if v["calls"]:
                kernels[ev] = {"calls": v["calls"], "GB": round(v["bytes"] / 1e9, 3),
                               "ms": round(v["ms"], 3),
                               "GBps": round(v["bytes"] / v["ms"] / 1e6, 1) if v["ms"] else None}
        dom = max((e for e in ("dotvec", "multvec", "matmult") if st[e]["ms"] > 0),
                  key=lambda e: st[e]["ms"], default=None)
        roofline = None
        if dom:
            kname = {"dotvec": "k_dotw", "multvec": "k_multvec_dot", "matmult": "k_spmv_tiles"}[dom]
            ach = st[dom]["bytes"] / st[dom]["ms"] / 1e6
            per_launch = st[dom]["bytes"] / st[dom]["calls"]
            roofline = {"bound": "hbm", "kernel": kname,
                        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": None, "launches": st[dom]["calls"],
                        "avg_launch_ms": round(st[dom]["ms"] / st[dom]["calls"], 4),
                        "bytes_per_launch": round(per_launch)}
            # HBM bytes per launch from the PMC passes (cannot be collected from inside this
            # process: they come from the committed rocprofv3 summary of this same command)
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    pmc = json.load(f)
                k = pmc["kernels"].get(kname)
                if k and k.get("traffic_over_algorithmic") and pmc.get("workload") == wname:
                    roofline["traffic"] = round(k["traffic_over_algorithmic"] * per_launch)
                    roofline["traffic_source"] = "profiles/pmc_traffic.json: measured traffic/algorithmic = %.4f over %d launches" % (
                        k["traffic_over_algorithmic"], k["launches"])
            except (OSError, ValueError, KeyError):
                pass
        mv = marks["c1"][0] - marks["c0"][0]
        gs = marks["c1"][1] - marks["c0"][1]
        out = {
            "metric": "EPSSolve wall-time to nev=32 converged eigenpairs; achieved HBM GB/s",
            "value_is": "achieved HBM GB/s: algorithmic bytes of the K timed Krylov-Schur restarts / wall time",
            "value": round(value, 1), "unit": "GB/s", "n_gpus": size, "steps": Ksteps, "warmup": W,
            "ms_per_step": round(1e3 * dt / Ksteps, 3), "higher_is_better": True, "scaling": "weak" if args.workload in ("ex2", "lap3d") else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wname, "rows_per_gpu": row1 - row0, "nnz_per_gpu": nnz_local, "nev": args.nev,
                       "ncv": eps.dimensions()[1], "tol": 1e-8, "partition": "row blocks, %d rank(s)" % size,
                       "transport": args.transport if size > 1 else "none",
                       "spmv_per_step": mv / Ksteps, "gs_passes_per_step": gs / Ksteps},
            "frac_of_hbm_peak": round(value / (HBM_PEAK_GBS * size), 4),
            "unfused_model_GBps": round(unfused / dt / 1e9, 1),
            "roofline": roofline, "kernels": kernels,
            "solver_state": {"restarts": eps.its, "last": marks["last"], "nconv": eps.nconv,
                             "reason": eps.reason},
        }
        if args.to_convergence:
            out["time_to_convergence_s"] = round(dt, 3)
            out["max_relative_residual"] = max_res
            out["max_eigenvalue_error_vs_analytic"] = eig_err
        if not args.no_cpu_baseline and size == 1:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(args.cpu_grid, args.nev, args.cpu_restarts, threads)
        print(json.dumps(out), flush=True)

    eps.destroy()
    A.destroy()
    ctx.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
