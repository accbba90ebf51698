"""PETSc binary matrix files, restated with numpy (test infrastructure: only the tests
use this module; the product reads the files in C, slepc_amd/csrc/host/mat.c).

The format is the one PetscViewerBinary writes for an AIJ matrix and MatLoad reads
(the reference's src/eps/examples/tutorials/ex4.c:56-60 loads
share/slepc/datafiles/matrices/*.petsc this way): big-endian int32 class id 1211216,
rows, columns, nonzeros; int32 row lengths; int32 global column indices; float64 values.
"""
import numpy as np

MAT_FILE_CLASSID = 1211216


def read(path):
    """-> (M, N, rowptr int64, colidx int32, val float64), rows in ascending column order."""
    raw = open(path, "rb").read()
    hdr = np.frombuffer(raw, dtype=">i4", count=4)
    if hdr[0] != MAT_FILE_CLASSID:
        raise ValueError("%s: not a PETSc binary matrix" % path)
    M, N, nz = int(hdr[1]), int(hdr[2]), int(hdr[3])
    if len(raw) != 16 + 4 * M + 12 * nz:
        raise ValueError("%s: %d bytes, expected %d" % (path, len(raw), 16 + 4 * M + 12 * nz))
    lens = np.frombuffer(raw, dtype=">i4", count=M, offset=16).astype(np.int64)
    colidx = np.frombuffer(raw, dtype=">i4", count=nz, offset=16 + 4 * M).astype(np.int32)
    val = np.frombuffer(raw, dtype=">f8", count=nz, offset=16 + 4 * M + 4 * nz).astype(np.float64)
    rowptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    rows = np.repeat(np.arange(M), lens)
    order = np.lexsort((colidx, rows))
    return M, N, rowptr, colidx[order], val[order]


def write(path, M, N, rowptr, colidx, val):
    rowptr = np.asarray(rowptr, dtype=np.int64)
    with open(path, "wb") as f:
        f.write(np.array([MAT_FILE_CLASSID, M, N, rowptr[-1]], dtype=">i4").tobytes())
        f.write(np.diff(rowptr).astype(">i4").tobytes())
        f.write(np.asarray(colidx).astype(">i4").tobytes())
        f.write(np.asarray(val).astype(">f8").tobytes())
