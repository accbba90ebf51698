#!/usr/bin/env python3
"""Summaries of rocprofv3 CSV output for a bench.py run, cut to the timed region.

bench.py launches a one-workgroup ``k_vec_set`` right before and right after the
timed restarts; this tool keeps the dispatches between those two markers.

    prof_summary.py trace  <kernel_trace.csv>              per-kernel calls / avg / total
    prof_summary.py pmc    <fetch counter_collection.csv> <write counter_collection.csv> [bench.json]
                           per-kernel HBM traffic: (2*FETCH_SIZE + WRITE_SIZE) KiB per launch
                           (FETCH_SIZE reads half of a wide coalesced stream on gfx950;
                           WRITE_SIZE is exact for 16-byte-per-lane stores)
    prof_summary.py counters <counter_collection.csv>       every counter of the pass, per kernel and launch
"""
import csv
import json
import re
import sys
from collections import OrderedDict


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+)", name)
    return m.group(1) if m else name


def rows_between_markers(path):
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))

    def is_marker(r):
        gx = r.get("Grid_Size_X") or r.get("Grid_Size")
        return short(r["Kernel_Name"]) == "k_vec_set" and int(gx) == 256
    ids = sorted({int(r["Dispatch_Id"]) for r in rows if is_marker(r)})
    if len(ids) < 2:
        raise SystemExit("%s: expected two marker dispatches, found %d" % (path, len(ids)))
    lo, hi = ids[0], ids[-1]
    return [r for r in rows if lo < int(r["Dispatch_Id"]) < hi]


def trace(path):
    agg = OrderedDict()
    for r in rows_between_markers(path):
        k = short(r["Kernel_Name"])
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(k, [0, 0, None, 0])
        a[0] += 1
        a[1] += d
        a[2] = d if a[2] is None else min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    out = []
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append({"kernel": k, "calls": a[0], "total_ms": a[1] / 1e6, "avg_ms": a[1] / a[0] / 1e6,
                    "min_ms": a[2] / 1e6, "max_ms": a[3] / 1e6, "pct": 100.0 * a[1] / total})
    return out


def counter_sums(path, counter):
    agg = OrderedDict()
    seen = set()
    for r in rows_between_markers(path):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        a = agg.setdefault(k, [0, 0.0])
        key = (r["Dispatch_Id"], r["Counter_Name"])
        if key not in seen:
            seen.add(key)
            a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main():
    mode = sys.argv[1]
    if mode == "trace":
        res = trace(sys.argv[2])
        print("%-22s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "%"))
        for r in res:
            print("%-22s %7d %12.3f %10.4f %10.4f %10.4f %6.2f" % (r["kernel"], r["calls"], r["total_ms"], r["avg_ms"],
                                                                    r["min_ms"], r["max_ms"], r["pct"]))
        print(json.dumps(res))
    elif mode == "pmc":
        fetch = counter_sums(sys.argv[2], "FETCH_SIZE")
        write = counter_sums(sys.argv[3], "WRITE_SIZE")
        model = {}
        if len(sys.argv) > 4:
            line = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
            names = {"dotvec": "k_dotw", "multvec": "k_multvec_dot", "matmult": "k_spmv_tiles",
                     "multinplace": "k_multinplace", "scale": "k_scale_cols"}
            if "k_spmv_rowcode" in fetch:               # the operator has the row-code form: that kernel multiplies
                names["matmult"] = "k_spmv_rowcode"
            for ev, kn in names.items():
                if ev in line["kernels"]:
                    model[kn] = line["kernels"][ev]["GB"] * 1e9
        res = OrderedDict()
        print("%-22s %7s %16s %16s %16s %16s %7s" % ("kernel", "calls", "fetch_B(x2)", "write_B", "traffic_B/launch",
                                                    "model_B/launch", "ratio"))
        for k in fetch:
            n = fetch[k][0]
            fb = 2.0 * fetch[k][1] * 1024.0
            wb = write.get(k, [0, 0.0])[1] * 1024.0
            per = (fb + wb) / n
            mod = model.get(k)
            res[k] = {"launches": n, "fetch_bytes_corrected": fb, "write_bytes": wb, "traffic_per_launch": per,
                      "algorithmic_per_launch": mod / n if mod else None,
                      "traffic_over_algorithmic": (fb + wb) / mod if mod else None}
            print("%-22s %7d %16.0f %16.0f %16.0f %16s %7s" % (k, n, fb, wb, per, "%.0f" % (mod / n) if mod else "-",
                                                              "%.3f" % ((fb + wb) / mod) if mod else "-"))
        print(json.dumps(res))
    elif mode == "counters":
        # every counter of one pass, summed over the instances and averaged over a kernel's launches
        agg = OrderedDict()
        for r in rows_between_markers(sys.argv[2]):
            k = short(r["Kernel_Name"])
            a = agg.setdefault(k, OrderedDict()).setdefault(r["Counter_Name"], [set(), 0.0])
            a[0].add(r["Dispatch_Id"])
            a[1] += float(r["Counter_Value"])
        print("%-22s %-28s %9s %20s" % ("kernel", "counter", "launches", "per launch"))
        res = OrderedDict()
        for k, cs in agg.items():
            for c, (ids, tot) in cs.items():
                print("%-22s %-28s %9d %20.0f" % (k, c, len(ids), tot / len(ids)))
                res.setdefault(k, {})[c] = {"launches": len(ids), "per_launch": tot / len(ids)}
        print(json.dumps(res))
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
