P=64
for cfg in "1 1" "2 1.5" "3 1.5" "3 1" "3 2" "4 1.5" "4 2"; do
  set -- $cfg
  echo "== pairs=$P streams=$1 waves=$2"
  A3D_ICP_STREAMS=$1 A3D_ICP_WAVES=$2 python bench.py --steps 20 --warmup 3 --no-extras --cpu-pairs 0 --pairs-per-gpu $P 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.0f pairs/s  ms/step %.3f  failed %s' % (d['value'], d['ms_per_step'], d['extra'].get('failed_pairs')))"
done
