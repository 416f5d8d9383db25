for P in 8 16 24 32 64 128; do
  echo "pairs $P: $(timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-extras --cpu-pairs 0 --pairs-per-gpu $P 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.3f pairs/s %.0f us/pair %.1f kernel_us %.1f frac %.3f'%(d['ms_per_step'], d['value'], 1e6/d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac']))")"
done
