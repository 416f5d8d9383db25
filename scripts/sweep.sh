# One parameterised tuning sweep over bench.py (replaces the round-1 variants*/streams_sweep*/level_knobs*/
# hybrid/waves/pairs scripts).  Each argument is one configuration: environment assignments, optionally followed
# by "--" and extra bench.py flags.  Run on an MI355X from the repository root, e.g.
#   bash scripts/sweep.sh "A3D_ICP_STREAMS=1" "A3D_ICP_STREAMS=3 A3D_ICP_WAVES=1.5" "A3D_ICP_STREAMS=3 -- --pairs-per-gpu 128"
#   bash scripts/sweep.sh "A3D_ICP_VARIANT=8,1" "A3D_ICP_ACCUM=mfma A3D_ICP_VARIANT=16,2" "A3D_ICP_PERSISTENT_LEVELS=4"
export A3D_LIBRARY=${A3D_LIBRARY:-$(cd "$(dirname "$0")/.." && pwd)/align3d_amd/csrc/libalign3d_hip_diag.so}  # the knobs exist in the diagnostics build only
STEPS=${STEPS:-20}
WARMUP=${WARMUP:-5}
for cfg in "$@"; do
  envs="${cfg%%--*}"
  extra=""
  case "$cfg" in *--*) extra="${cfg#*--}";; esac
  echo "== $cfg"
  env $envs timeout -k 10 300 python3 bench.py --steps $STEPS --warmup $WARMUP --no-extras --cpu-pairs 0 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('value %.0f pairs/s  ms/step %.3f  avg_launch_us %.1f  frac %.3f  failed %s  level us %s  level frac %s' % (d['value'], d['ms_per_step'], r.get('avg_launch_us', 0), r['frac'], d['extra'].get('failed_pairs'), ['%.1f' % r.get('level%d_avg_launch_us' % l, 0) for l in range(3)], ['%.3f' % (r.get('level%d_frac' % l) or 0) for l in range(3)]))"
done
