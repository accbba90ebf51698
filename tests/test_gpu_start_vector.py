"""The start vector (the stand-in for BVSetRandomColumn, bvops.c:462-479; EPSGetStartVector,
epssolve.c:675-710): the device fill must produce, bit for bit, the oracle's counter-based
values of the GLOBAL row index, for any row block and seed -- the benchmark, EPSSolve
without an initial space and both SVD drivers start from it."""
import ctypes as C

import numpy as np
import pytest

from oracle import eps as oeps

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("row0", [0, 12345, (1 << 33) + 7])
@pytest.mark.parametrize("seed", [0x12345678, 0xDEADBEEFCAFE])
def test_fill_hash_equals_oracle(gpu_ctx, row0, seed):
    n = 100003
    x = gpu_ctx.vec_alloc(n)
    gpu_ctx.vec_set(x, np.nan, n)
    gpu_ctx.vec_fill_hash(x, n, row0, seed)
    got = gpu_ctx.download(x, n)
    want = oeps.splitmix64_uniform(np.arange(row0, row0 + n, dtype=np.uint64), seed=seed)
    assert np.array_equal(got, want)
    assert got.min() >= -1.0 and got.max() < 1.0 and abs(got.mean()) < 0.02
    gpu_ctx.vec_free(x)


def test_partition_independent(gpu_ctx):
    """Two row blocks filled separately equal one fill of the whole range."""
    n = 50000
    x = gpu_ctx.vec_alloc(n)
    gpu_ctx.vec_fill_hash(x, n, 0, 99)
    whole = gpu_ctx.download(x, n)
    gpu_ctx.vec_fill_hash(x, 20000, 0, 99)
    a = gpu_ctx.download(x, 20000)
    gpu_ctx.vec_fill_hash(x, 30000, 20000, 99)
    b = gpu_ctx.download(x, 30000)
    assert np.array_equal(np.concatenate((a, b)), whole)
    gpu_ctx.vec_free(x)


def test_solver_default_start_is_the_oracle_start(gpu_ctx):
    """BVSetRandomColumn through the host library (what EPSGetStartVector calls when there is no
    initial space): the column is the oracle's default start vector."""
    from slepc_amd import kseps as K
    lib = K.load()
    n, row0 = 4097, 0
    t = K.Vec.create_hip(gpu_ctx, n, n, row0)
    bv = K.BVP()
    K.check(lib.BVCreate(None, C.byref(bv)))
    K.check(lib.BVSetSizesFromVec(bv, t.p, 3))
    K.check(lib.BVSetType(bv, b"hip"))
    K.check(lib.BVSetRandomColumn(bv, 1))
    v = K.VecP()
    K.check(lib.BVGetColumn(bv, 1, C.byref(v)))
    got = np.empty(n)
    K.check(lib.VecGetValuesHost(v, got.ctypes.data_as(K.c_dp)))
    K.check(lib.BVRestoreColumn(bv, 1, C.byref(v)))
    # the first random column of a BV is drawn with the seed 0x12345678 (bvbasic.c:884), the
    # default of the oracle's start vector; the seed then moves on
    assert np.array_equal(got, oeps.splitmix64_uniform(np.arange(n, dtype=np.uint64)))
    assert bv.contents.rand_seed != 0x12345678
    K.check(lib.BVDestroy(C.byref(bv)))
    t.destroy()
