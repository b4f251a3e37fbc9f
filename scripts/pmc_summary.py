"""Summarise rocprofv3 PMC passes into HBM bytes per launch of each decode-step kernel.

usage: pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>

Each pass is `rocprofv3 --pmc <counter> --output-format csv -d <dir> -- python3 bench.py ...`.
Launches are keyed by (kernel, grid, workgroup, kernel launched before it): the
o-projection and the down-projection run the same GEMM instantiation on the same
grid and are told apart by their predecessor (attention vs the gate/up GEMM).
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 64 B
per 128-B request of a wide coalesced read; WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(?:mi355::)?([A-Za-z0-9_]+)", name)
    base = m.group(1) if m else name[:40]
    if base == "linear_tiled_kernel":
        t = re.search(r"<(.*)>", name)
        base += "<" + (t.group(1).replace("mi355::", "") if t else "") + ">"
    return base


def read(dirname, counter):
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"]),
                                 float(r["Counter_Value"])))
    rows.sort()
    out, prev = defaultdict(list), "-"
    for _, name, grid, wg, val in rows:
        s = short(name)
        out[(s, grid, wg, prev)].append(val)
        prev = s
    return out


def main():
    fetch, write = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
    res = {}
    for key, vals in sorted(fetch.items(), key=lambda kv: -sum(kv[1])):
        w = write.get(key, [])
        if len(vals) < 2:
            continue
        f_avg = sum(vals) / len(vals)
        w_avg = sum(w) / len(w) if w else 0.0
        res[f"{key[0]} grid={key[1]} wg={key[2]} after={key[3]}"] = {
            "launches": len(vals), "fetch_size_kb_avg": round(f_avg, 1), "write_size_kb_avg": round(w_avg, 1),
            "hbm_bytes_per_launch": int((2 * f_avg + w_avg) * 1024)}
    with open(sys.argv[3], "w") as fh:
        json.dump({"correction": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per "
                                 "128-B request of a wide coalesced read; WRITE_SIZE is exact)", "kernels": res}, fh, indent=1)
    for k, v in list(res.items())[:16]:
        print(f"{v['hbm_bytes_per_launch']:>12d} B x{v['launches']:<4d} {k}")


if __name__ == "__main__":
    main()
