#!/usr/bin/env python3
"""Timeline of ONE decode step from a rocprofv3 kernel trace: every kernel's duration and the
idle gap before it (previous kernel's end -> this kernel's start), summed per kernel kind.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 8 \
        --warmup 2 --no-cpu-baseline --no-kernel-timing
    python3 scripts/step_timeline.py gpurun_out/tl [out.json]

A step is the kernels between two consecutive argmax launches; the last complete step of the trace
is taken (steady state, graph replay)."""
import csv
import glob
import json
import os
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(?:mi355::)?(\w+)(<[^(]*>)?", name)
    if not m:
        return name[:60]
    base, targs = m.group(1), (m.group(2) or "")
    targs = re.sub(r"mi355::", "", targs)
    return (base + targs)[:90]


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {root}")
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # a step ends with the greedy pick: the key-to-id kernel of the chunked argmax, or the one-launch argmax
    last = "argmax_finish_kernel" if any("argmax_finish_kernel" in r[2] for r in rows) else "argmax_kernel"
    marks = [i for i, r in enumerate(rows) if last in r[2]]
    if len(marks) < 3:
        raise SystemExit("fewer than three argmax launches in the trace")
    steps = []
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a + 1:b + 1]
        if len(seg) > 100 and not any("prefill" in r[2] for r in seg):
            steps.append((rows[a][1], seg))
    prev_end, seg = steps[-1]
    t0 = seg[0][0]
    per = {}
    total_busy = total_gap = 0
    last_end = prev_end
    for s, e, n in seg:
        k = short(n)
        d = per.setdefault(k, {"launches": 0, "busy_us": 0.0, "gap_before_us": 0.0})
        d["launches"] += 1
        d["busy_us"] += (e - s) / 1e3
        gap = max(s - last_end, 0) / 1e3
        d["gap_before_us"] += gap
        total_busy += (e - s) / 1e3
        total_gap += gap
        last_end = max(last_end, e)
    wall = (seg[-1][1] - prev_end) / 1e3
    out = {"kernels_in_step": len(seg), "step_wall_us": round(wall, 1), "sum_kernel_us": round(total_busy, 1),
           "sum_gaps_us": round(total_gap, 1), "steps_in_trace": len(steps),
           "per_kernel": {k: {"launches": v["launches"], "avg_us": round(v["busy_us"] / v["launches"], 2),
                              "avg_gap_before_us": round(v["gap_before_us"] / v["launches"], 2),
                              "total_us": round(v["busy_us"], 1)}
                          for k, v in sorted(per.items(), key=lambda kv: -kv[1]["busy_us"])}}
    text = json.dumps(out, indent=1)
    print(text)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
