# one-rank RCCL rehearsal of the N > 1 path (process group, all-gather on the context stream) on a one-GPU box
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29523 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
for q in default 4; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  echo -n "GPU_MAX_HW_QUEUES=$q (unset -> the package asks for 16): "
  python3 bench.py --gpus 1 --steps 10 --warmup 3 --rehearse-collective --no-extras --cpu-pairs 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.0f pairs/s  ms/step %.3f  gather ok %s' % (d['value'], d['ms_per_step'], d['extra'].get('gather_matches_local_poses')))"
done
