#!/usr/bin/env python3
"""Per-kernel launch durations from a rocprofv3 results database (rocpd, the default output of
`rocprofv3 --kernel-trace`): consecutive launches with the same name, grid and LDS size form one row.
Usage: rocpd_kernels.py results.db [substring of the kernel names to keep]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    keep = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = db.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count from kernels order by start").fetchall()
    groups = []
    for name, t0, t1, grid, wg, lds, vgpr in rows:
        if keep and keep not in name:
            continue
        key = (name.split("(")[0][-48:], grid // max(wg, 1), wg, lds, vgpr)
        for g in groups[-6:]:
            if g[0] == key:
                g[1].append((t1 - t0) / 1e3)
                break
        else:
            groups.append((key, [(t1 - t0) / 1e3]))
    for key, v in groups:
        v = sorted(v)
        print("%-50s workgroups %8d x %4d  lds %6d  vgpr %3d  launches %3d  median %9.1f us  min %9.1f us" % (
            key[0], key[1], key[2], key[3], key[4], len(v), v[len(v) // 2], v[0]))


if __name__ == "__main__":
    main()
