# LDS bank conflicts per kernel of the headline bench (one PMC pass): conflict cycles against LDS-active cycles
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MI355_TILED_LINEARS=1
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_lds -- python3 bench.py --steps 4 --warmup 1 --no-kernel-timing --no-cpu-baseline --no-served --segments 1 ${BENCH_ARGS:-} > gpurun_out/pmc_lds.log 2>&1 || exit 1
python3 - <<'P'
import csv, glob, collections, json, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void mi355::", "")[:90]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_LDS": n[k] += 1
out = {}
for k, v in acc.items():
    if v["SQ_INSTS_LDS"] == 0: continue
    out[k] = {"launches": n[k], "lds_insts_per_launch": round(v["SQ_INSTS_LDS"] / n[k]), "bank_conflict_cycles_per_launch": round(v["SQ_LDS_BANK_CONFLICT"] / n[k]),
              "lds_active_cycles_per_launch": round(v["SQ_ACTIVE_INST_LDS"] / n[k]),
              "conflict_share_of_lds_active": round(v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_ACTIVE_INST_LDS"], 1), 3)}
json.dump(out, open("gpurun_out/pmc_lds.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["bank_conflict_cycles_per_launch"] * kv[1]["launches"])[:24]:
    print(k, v)
P
rm -rf gpurun_out/pmc_lds
