import sys
sys.path.insert(0, '.')
import numpy as np
from slepc_amd import ksgpu, matgen
ctx = ksgpu.Context(0)
g = 4096; n = g * g; m = 65
rp, ci, va = matgen.laplacian2d(g)
A = ksgpu.DeviceCSR(ctx, rp, ci, va)
V = ksgpu.DeviceBV(ctx, n, m)
for j in range(m):
    ctx.vec_fill_hash(V.column(j), n, 0, 1000 + j)
ctx.sync(); ctx.enable_timing(True)
J = 49
def run(name, fn, reps=10):
    fn(); ctx.sync(); ctx.stats_reset()
    for _ in range(reps): fn()
    ctx.sync(); s = ctx.stats()
    print(name, {k: round(v["ms"] / v["calls"], 4) for k, v in s.items() if v["calls"]}, flush=True)
V.set_active(0, J + 1)
q = np.full(J, 1e-9)
run("dot alone (no sync between)", lambda: V.dotvec(V.column(J)))
run("spmv then dot", lambda: (A.mult(V.column(J - 1), V.column(J)), V.dotvec(V.column(J))))
def seq3():
    A.mult(V.column(J - 1), V.column(J)); V.dotvec(V.column(J))
    V.set_active(0, J); V.multvec(-1.0, 1.0, V.column(J), q); V.set_active(0, J + 1)
run("spmv, dot, multvec", seq3)
def seq_scale():
    ctx.vec_scale(V.column(J), 1.0000001, n); V.dotvec(V.column(J))
run("scale(y) then dot", seq_scale)
def seq_copy():
    V.copy_column(J - 1, 64); V.dotvec(V.column(J))
run("copy of another column then dot", seq_copy)
