# SQ counters of the alignment kernel in the headline batch, level by level (the grid size tells the level): share of
# wave cycles parked on s_waitcnt / barrier, stalled at issue, issuing; VALU instructions per wave.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_icp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_icp -- python3 bench.py --steps 2 --warmup 1 --no-extras --cpu-pairs 0 > gpurun_out/pmc_icp.out 2> gpurun_out/pmc_icp.err &&
python3 - <<'PY'
import csv, glob, re, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_icp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "image_icp" not in r["Kernel_Name"]:
            continue
        gx = int(r["Grid_Size"])
        name = f"image_icp_head_kernel grid={gx}"
        rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel | launches | wait_any % | wait_inst % | active % | valu insts/wave | vmem-read insts/wave")
for k, c in sorted(rows.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    wc = sum(c.get("SQ_WAVE_CYCLES", [0])) or 1.0
    waves = sum(c.get("SQ_WAVES", [0])) or 1.0
    print(f"{k} | {len(c.get('SQ_WAVE_CYCLES', []))} | {100*sum(c.get('SQ_WAIT_ANY',[0]))/wc:.0f} | {100*sum(c.get('SQ_WAIT_INST_ANY',[0]))/wc:.0f} | {100*sum(c.get('SQ_ACTIVE_INST_ANY',[0]))/wc:.0f} | {sum(c.get('SQ_INSTS_VALU',[0]))/waves:.0f} | {sum(c.get('SQ_INSTS_VMEM_RD',[0]))/waves:.0f}")
PY
