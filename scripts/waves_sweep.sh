for W in 1 2 3 4 5; do
  echo "waves $W: $(A3D_ICP_WAVES=$W timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-extras --cpu-pairs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.3f pairs/s %.0f kernel_us %.1f frac %.3f'%(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac']))")"
done
