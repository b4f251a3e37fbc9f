#!/usr/bin/env python3
"""Timeline of ONE prefill step from a rocprofv3 kernel trace: the kernels from the step's first
launch to the sampling of its first tokens, their durations and the idle gaps between them
(previous kernel's end -> this kernel's start), summed per kernel kind.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 4 \
        --warmup 2 --no-cpu-baseline --no-kernel-timing --no-served --segments 1
    python3 scripts/prefill_timeline.py gpurun_out/tl [out.json]

The trace is cut at idle gaps above 0.4 ms (the host between two steps); the last piece that holds
prefill attention launches, up to the sampling of its first tokens, is the step reported."""
import csv
import glob
import json
import os
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    if name.startswith(("Cijk_", "Custom_Cijk_")):
        m = re.search(r"MT\d+x\d+x\d+", name)
        return "hipBLASLt GEMM " + (m.group(0) if m else "")
    m = re.match(r"(?:mi355::)?(?:at::native::)?([\w:]+)(<[^(]*>)?", name)
    return ((m.group(1) + (m.group(2) or "")) if m else name)[:80].replace("mi355::", "")


def main():
    root = sys.argv[1]
    rows = []
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # steps are separated by host work (scheduling, staging the next step's inputs): cut the trace
    # at idle gaps above 0.4 ms and take the last piece that holds prefill attention launches
    cuts = [0] + [i for i in range(1, len(rows)) if rows[i][0] - max(r[1] for r in rows[max(0, i - 4):i]) > 400_000] + [len(rows)]
    seg = None
    for a, b in zip(cuts[:-1], cuts[1:]):
        if any("paged_attn_prefill_kernel" in r[2] for r in rows[a:b]):
            seg, prev_end = rows[a:b], (rows[a - 1][1] if a else rows[a][0])
    if seg is None:
        raise SystemExit("no prefill attention launch in the trace")
    # the step ends with the sampling of its first tokens
    for i, r in enumerate(seg):
        if "argmax" in r[2] and any("paged_attn_prefill_kernel" in q[2] for q in seg[:i]):
            seg = seg[:i + 1]
            break
    per, busy, gaps = {}, 0.0, 0.0
    last_end = seg[0][0]
    big = []
    host_gap_before_step_us = round((seg[0][0] - prev_end) / 1e3, 1)
    for s, e, n in seg:
        k = short(n)
        d = per.setdefault(k, {"launches": 0, "busy_us": 0.0, "gap_before_us": 0.0})
        d["launches"] += 1
        d["busy_us"] += (e - s) / 1e3
        g = max(s - last_end, 0) / 1e3
        d["gap_before_us"] += g
        busy += (e - s) / 1e3
        gaps += g
        if g > 20:
            big.append({"before": k, "gap_us": round(g, 1)})
        last_end = max(last_end, e)
    out = {"kernels": len(seg), "wall_us": round((seg[-1][1] - seg[0][0]) / 1e3, 1), "busy_us": round(busy, 1),
           "gaps_us": round(gaps, 1), "idle_before_first_kernel_us": host_gap_before_step_us,
           "gaps_over_20us": big[:20],
           "per_kernel": {k: {"launches": v["launches"], "busy_us": round(v["busy_us"], 1),
                              "avg_us": round(v["busy_us"] / v["launches"], 2),
                              "gap_before_us": round(v["gap_before_us"], 1)}
                          for k, v in sorted(per.items(), key=lambda kv: -kv[1]["busy_us"])}}
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
