run() {
  echo "== $*"
  env "$@" python bench.py --steps 20 --warmup 5 --no-extras --cpu-pairs 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.0f pairs/s  ms/step %.3f  failed %s' % (d['value'], d['ms_per_step'], d['extra'].get('failed_pairs')))"
}
run A3D_ICP_WAVES_LEVELS=1.5,1.5,1.5
run A3D_ICP_WAVES_LEVELS=3,1.5,1.5
run A3D_ICP_WAVES_LEVELS=2,1.5,1.5
run A3D_ICP_WAVES_LEVELS=1,1.5,1.5
run A3D_ICP_WAVES_LEVELS=1.5,1,1
run A3D_ICP_WAVES_LEVELS=1.5,2,1
run A3D_ICP_WAVES_LEVELS=2,2,1
