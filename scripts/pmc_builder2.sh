# Second SQ pass over the frame builder's kernels: which PIPE the cycles go to.  SQ_BUSY_CYCLES = cycles the SQs had
# waves; SQ_ACTIVE_INST_VALU / _LDS / _VMEM / _SCA = cycles (x4) an instruction of that kind was executing;
# SQ_INST_CYCLES_VMEM etc.  VALU utilisation = ACTIVE_INST_VALU x 4 / (BUSY_CYCLES x SIMDs per SQ ...): printed as a ratio
# to SQ_WAVE_CYCLES per wave and as VALU-active cycles per SIMD against the kernel's duration.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_builder2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/pmc_builder2 -- python3 scripts/build_trace_probe.py ${FRAMES:-32} ${PRIO:--1} > gpurun_out/pmc_builder2.out 2> gpurun_out/pmc_builder2.err &&
python3 - <<'PY'
import csv, glob, re, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_builder2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        name = (m.group(1) if m else r["Kernel_Name"][:30]) + "/" + r["Grid_Size"]
        rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel/grid | launches | wave_cycles | busy_cycles | active VALU / wave_cycles % | LDS % | VMEM % | scalar % | salu insts/wave | waves")
for k, c in sorted(rows.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    wc = sum(c.get("SQ_WAVE_CYCLES", [0])) or 1.0
    waves = sum(c.get("SQ_WAVES", [0])) or 1.0
    g = lambda n: sum(c.get(n, [0]))
    print(f"{k} | {len(c.get('SQ_WAVE_CYCLES', []))} | {wc:.3g} | {g('SQ_BUSY_CYCLES'):.3g} | {100*g('SQ_ACTIVE_INST_VALU')/wc:.0f} | {100*g('SQ_ACTIVE_INST_LDS')/wc:.0f} | {100*g('SQ_ACTIVE_INST_VMEM')/wc:.0f} | {100*g('SQ_ACTIVE_INST_SCA')/wc:.0f} | {g('SQ_INSTS_SALU')/waves:.0f} | {waves:.0f}")
PY
