#!/usr/bin/env python3
"""Where a kernel's wave cycles go, from one rocprofv3 --pmc pass (csv): per kernel the averages of
the SQ counters of the pass and the fractions of SQ_WAVE_CYCLES (quad-cycle units) spent waiting
(parked on s_waitcnt / barrier), issue-stalled, and issuing; MFMA-busy as a fraction of the
kernel's SIMD-cycles when GRBM_GUI_ACTIVE is in the pass.
usage: pmc_kernel_mix.py <dir> [kernel-name substring]"""
import csv, glob, json, os, re, sys
from collections import defaultdict

root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = re.sub(r"^void ", "", r["Kernel_Name"])
            if pat and pat not in name:
                continue
            key = re.sub(r"mi355::", "", name.split("(")[0])[:80] + f" grid={r['Grid_Size']}"
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    d = {"launches": len(next(iter(cs.values()))), **{c: round(v, 1) for c, v in avg.items()}}
    wc = avg.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
            if c in avg:
                d["frac_" + c] = round(avg[c] / wc, 3)
    if "GRBM_GUI_ACTIVE" in avg and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        cycles = avg["GRBM_GUI_ACTIVE"] / 8.0           # summed over the 8 XCDs
        d["kernel_cycles"] = round(cycles)
        d["mfma_busy_frac"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 256 * 4), 4)
    out[k] = d
print(json.dumps(out, indent=1))
