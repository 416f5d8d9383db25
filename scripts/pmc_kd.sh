# SQ counters of the kd-tree build's kernels (one counter-only pass): where the waves' cycles go.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_kd
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d gpurun_out/pmc_kd -- python3 scripts/kd_probe.py > gpurun_out/pmc_kd.out 2> gpurun_out/pmc_kd.err &&
python3 - <<'PY'
import csv, glob, re, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_kd/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        name = (m.group(1) if m else r["Kernel_Name"][:30]) + "/" + r["Grid_Size"]
        rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel/grid | launches | wave_cycles per launch | wait_any % | wait_inst % | active % | valu insts/wave | lds insts/wave | bank-conflict cycles / wave_cycles %")
for k, c in sorted(rows.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    wc = sum(c.get("SQ_WAVE_CYCLES", [0])) or 1.0
    waves = sum(c.get("SQ_WAVES", [0])) or 1.0
    nl = len(c.get('SQ_WAVE_CYCLES', [])) or 1
    print(f"{k} | {nl} | {wc/nl:.3g} | {100*sum(c.get('SQ_WAIT_ANY',[0]))/wc:.0f} | {100*sum(c.get('SQ_WAIT_INST_ANY',[0]))/wc:.0f} | {100*sum(c.get('SQ_ACTIVE_INST_ANY',[0]))/wc:.0f} | {sum(c.get('SQ_INSTS_VALU',[0]))/waves:.0f} | {sum(c.get('SQ_INSTS_LDS',[0]))/waves:.0f} | {100*sum(c.get('SQ_LDS_BANK_CONFLICT',[0]))/wc:.1f}")
PY
