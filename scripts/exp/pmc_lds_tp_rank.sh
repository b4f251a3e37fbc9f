# LDS bank conflicts per kernel of a TP4 x 128-row and a TP8 x 256-row rank step (one PMC pass)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_lds_tp -- python3 scripts/bench_tp_rank.py --tps 4,8 --only sharded-timed > gpurun_out/pmc_lds_tp.log 2>&1 || exit 1
python3 - <<'P'
import csv, glob, collections, json, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_lds_tp/**/*counter_collection.csv", recursive=True):
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
json.dump(out, open("gpurun_out/pmc_lds_tp.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["bank_conflict_cycles_per_launch"] * kv[1]["launches"])[:30]:
    if v["bank_conflict_cycles_per_launch"]: print(k, v)
P
rm -rf gpurun_out/pmc_lds_tp
