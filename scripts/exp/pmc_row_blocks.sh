set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/rb_fetch -- python3 scripts/exp/run_row_block_forms.py > gpurun_out/rb_fetch.log 2>&1 || exit 1
python3 - <<'P'
import csv, glob, collections, json
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/rb_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "linear_tiled" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"])].append(float(r["Counter_Value"]))
out = {f"{k[0]} grid={k[1]}": {"launches": len(v), "hbm_read_MB_per_launch": round(2 * 1024 * sum(v) / len(v) / 1e6, 1)} for k, v in acc.items()}
json.dump({"algorithmic_weight_MB": 234.9, "note": "FETCH_SIZE x 2 x 1024 B (64 B counted per 128-B request on gfx950)", "kernels": out}, open("gpurun_out/rb_fetch.json", "w"), indent=1)
print(json.dumps(out, indent=1))
P
rm -rf gpurun_out/rb_fetch
