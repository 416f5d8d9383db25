# Pair groups on separate streams: bench value for 1..4 streams at 64 and 128 pairs per GPU.
for P in 64 128; do
for S in 1 2 3 4; do
  echo "== pairs=$P streams=$S"
  A3D_ICP_STREAMS=$S python bench.py --steps 10 --warmup 3 --no-extras --cpu-pairs 0 --pairs-per-gpu $P | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.0f pairs/s  ms/step %.3f  failed %s' % (d['value'], d['ms_per_step'], d['extra'].get('failed_pairs')))"
done; done
