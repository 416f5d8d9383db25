# kernel timeline of the one-rank RCCL rehearsal: what happens between one step's last kernel and the next step's first
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29521 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
rm -rf gpurun_out/nccl_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/nccl_trace -- python3 bench.py --gpus 1 --steps 6 --warmup 2 --rehearse-collective --no-extras --cpu-pairs 0 > gpurun_out/nccl_trace.json 2> gpurun_out/nccl_trace.err
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/nccl_trace/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [(r["Kernel_Name"].split("(")[0][-40:], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# print the sequence around job_finish kernels
idx = [i for i, n in enumerate(names) if "job_finish" in n[0]]
for i in idx[3:6]:
    for j in range(i - 2, min(i + 8, len(names))):
        n, s, e = names[j]
        gap = (s - names[j - 1][2]) / 1000 if j else 0
        print(f"{n:42s} dur {(e - s) / 1000:8.1f} us   gap before {gap:8.1f} us")
    print("---")
PY
