#!/usr/bin/env python3
"""Extracts the expected values that the reference's own test-suite holds for the
Krylov-Schur hot path into small JSON fixtures (data only: numbers printed in
``src/**/examples/**/output/*.out``).  Run in the build container, where the
reference tree is mounted:

    python tests/golden/make_golden.py [/root/reference]

The fixtures are committed; nothing under tests/ reads the reference tree at
test time.
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
NUM = r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?"


def read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def dump(name, obj, source):
    obj = dict(obj)
    obj["_source"] = source
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, indent=1)
    print("wrote", name)


def numbers_after(text, header, count):
    """`count` numbers (one per line) following the first line containing header."""
    tail = text.split(header, 1)[1]
    vals = []
    for line in tail.splitlines()[1:]:
        s = line.strip()
        if re.fullmatch(NUM, s):
            vals.append(float(s))
            if len(vals) == count:
                break
    assert len(vals) == count, (header, len(vals))
    return vals


def bv_test1():
    rel = "src/sys/classes/bv/examples/tests/output/test1_1_svec.out"
    t = read(rel)
    n, k, l = 10, 5, 3
    g = {}
    g["Y_after_mult"] = numbers_after(t, "After BVMult ", l * n)
    g["Y_after_multvec"] = numbers_after(t, "After BVMultVec", l * n)
    mtxt = t.split("After BVDot - ", 1)[1].split("After BVDotVec", 1)[0]
    rows = [[float(x) for x in re.findall(NUM, line)] for line in mtxt.splitlines()
            if re.match(r"^\s*" + NUM + r"\s", line) and "e+" in line or "e-" in line]
    rows = [r for r in rows if len(r) == k]
    assert len(rows) == l
    g["M"] = rows
    g["z"] = numbers_after(t, "After BVDotVec", k)
    g["X_after_multinplace"] = numbers_after(t, "After BVMultInPlace", k * n)
    g["norm_X0"] = re.search(r"2-Norm of X\[0\] = (\S+)", t).group(1)
    g["norm_fro"] = re.search(r"Frobenius Norm of X = (\S+)", t).group(1)
    g["first_row"] = [float(x) for x in re.findall(NUM, t.split("First row of X =", 1)[1])]
    dump("bv_test1.json", g, rel)


def bv_misc():
    g = {}
    t = read("src/sys/classes/bv/examples/tests/output/test2_1.out")
    g["test2_lines"] = t.strip().splitlines()[1:]
    g["test2_norm_ones"] = re.search(r"against X: (\S+)", t).group(1)
    t = read("src/sys/classes/bv/examples/tests/output/test4_1.out")
    g["test4_norm_X3"] = re.search(r"2-Norm of X\[3\] = (\S+)", t).group(1)
    g["test4_norm_fro"] = re.search(r"Frobenius Norm of X = (\S+)", t).group(1)
    t = read("src/sys/classes/bv/examples/tests/output/test13_1.out")
    g["test13_norm_fro"] = re.search(r"Frobenius Norm or X = (\S+)", t).group(1)
    t = read("src/sys/classes/bv/examples/tests/output/test6_1.out")
    g["test6_lines"] = t.strip().splitlines()[1:]
    t = read("src/sys/classes/bv/examples/tests/output/test7_1.out")
    g["test7_errors"] = re.findall(r"Norm of error: (\S+)", t)
    t = read("src/sys/classes/bv/examples/tests/output/test10_1.out")
    g["test10_lines"] = t.strip().splitlines()[1:]
    for t3 in ("test3", "test5", "test11"):
        g[t3 + "_lines"] = read("src/sys/classes/bv/examples/tests/output/%s_1.out" % t3).strip().splitlines()[1:]
    dump("bv_misc.json", g, "src/sys/classes/bv/examples/tests/output/test{2,3,4,5,6,7,10,11,13}_1.out")


def ds_tests():
    g = {}
    t = read("src/sys/classes/ds/examples/tests/output/test3_1.out")
    g["hep_compact_eigs"] = [x.strip() for x in t.split("Computed eigenvalues =", 1)[1].strip().splitlines()]
    t = read("src/sys/classes/ds/examples/tests/output/test1_1.out")
    body = t.split("Computed eigenvalues =", 1)[1]
    g["nhep_grcar_eigs"] = [x.strip() for x in body.strip().splitlines() if x.strip().endswith("i")]
    g["nhep_rnorm3"] = re.search(r"rnorm for 3rd vector = (\S+)", t).group(1)
    g["nhep_norm1"] = re.search(r"Norm of 1st vector = (\S+)", t).group(1)
    dump("ds_tests.json", g, "src/sys/classes/ds/examples/tests/output/test{1,3}_1.out")


def eps_tests():
    def eigs(rel, marker="required tolerance:"):
        t = read(rel)
        out = []
        for blk in t.split(marker)[1:]:
            line = blk.strip().splitlines()[0]
            out.append([x.strip() for x in line.split(",")])
        return out

    g = {}
    t = read("src/eps/examples/tutorials/output/ex1_1.out")
    g["ex1"] = {"n": 30, "nev": 1, "value": re.search(r"^\s+(\d\.\d+)\s+\S+e-\d+", t, re.M).group(1)}
    g["ex2"] = {"n": 72, "nev": 4, "ncv": 20, "values": eigs("src/eps/examples/tutorials/output/ex2_1.out")[0]}
    g["ex5"] = {"m": 15, "nev": 4, "which": "largest_real",
                "values": eigs("src/eps/examples/tutorials/output/ex5_1.out")[0]}
    g["ex19"] = {"grid": 10, "nev": 8, "ncv": 64, "which": "smallest_real",
                 "values": eigs("src/eps/examples/tutorials/output/ex19_1.out")[0]}
    g["test4"] = {"n": 30, "nev": 4, "values": eigs("src/eps/examples/tests/output/test4_1.out")[0]}
    g["test8"] = {"n": 20, "nev": 4, "ncv": 11, "values": eigs("src/eps/examples/tests/output/test8_1.out")[0]}
    g["test9"] = {"m": 15, "nev": 4, "which": "largest_magnitude",
                  "values": eigs("src/eps/examples/tests/output/test9_1.out")[0]}
    g["test9_3"] = {"m": 15, "nev": 3, "which": "largest_real", "tol": 1e-7, "solver": "arnoldi", "delayed": True,
                    "values": eigs("src/eps/examples/tests/output/test9_3.out")[0]}
    g["ex4"] = {"file": "rdb200.petsc", "nev": 4, "its": 5,
                "values": eigs("src/eps/examples/tutorials/output/ex4_1.out")[0]}
    g["test10_ks"] = {"n": 10, "m": 11, "nev": 4, "which": "largest_magnitude",
                      "values": eigs("src/eps/examples/tests/output/test10_1_ks.out")[0]}
    g["ex13"] = eigs("src/eps/examples/tutorials/output/ex13_1.out")[0]
    g["ex7"] = eigs("src/eps/examples/tutorials/output/ex7_1.out")[0]
    t2 = eigs("src/eps/examples/tests/output/test2_1.out")
    g["test2"] = {"n": 30, "nev": 4, "largest": t2[0], "smallest": t2[1]}
    dump("eps_tests.json", g, "src/eps/examples/{tutorials,tests}/output/*.out")


def svd_tests():
    def eigs(rel, marker="required tolerance:"):
        t = read(rel)
        return [[x.strip() for x in blk.strip().splitlines()[0].split(",")] for blk in t.split(marker)[1:]]

    g = {}
    t = read("src/svd/examples/tutorials/output/ex8_1.out")
    m = re.search(r"sigma_1=([\d.]+), sigma_n=([\d.]+)\n.*sigma_1/sigma_n=([\d.]+)", t)
    g["ex8"] = {"n": 30, "sigma_1": m.group(1), "sigma_n": m.group(2), "cond": m.group(3), "text": t}
    assert read("src/svd/examples/tests/output/test1_1.out") == t
    t = read("src/svd/examples/tutorials/output/ex14_1.out")
    g["ex14"] = {"file": "rdb200.petsc", "nsv": 4, "its": int(re.search(r"iterations of the method: (\d+)", t).group(1)),
                 "type": re.search(r"Solution method: (\w+)", t).group(1), "values": eigs("src/svd/examples/tutorials/output/ex14_1.out")[0],
                 "text": t}
    t = read("src/svd/examples/tutorials/output/ex15_1.out")
    g["ex15"] = {"n": 100, "mu": 1.49012e-08, "its": int(re.search(r"iterations of the method: (\d+)", t).group(1)),
                 "nconv": int(re.search(r"singular triplets: (\d+)", t).group(1)),
                 "sigma": re.findall(r"^\s+(\d+\.\d{6})\s+\S+$", t, re.M)}
    g["test3"] = {"rows": 35, "cols": 30, "nsv": 4, "values": eigs("src/svd/examples/tests/output/test3_1.out")[0]}
    g["test4"] = {"rows": 20, "cols": 22, "nsv": 1, "conv": "abs", "values": eigs("src/svd/examples/tests/output/test4_1.out")[0]}
    dump("svd_tests.json", g, "src/svd/examples/{tutorials,tests}/output/*.out")


if __name__ == "__main__":
    bv_test1()
    bv_misc()
    ds_tests()
    eps_tests()
    svd_tests()
