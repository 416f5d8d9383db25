for ab in 0 1 2; do for lay in raw packed; do
  echo "ablate $ab layout $lay: $(A3D_ICP_ABLATE=$ab A3D_ICP_LAYOUT=$lay timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-extras --cpu-pairs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.3f kernel_us %.1f frac %.3f'%(d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac']))")"
done; done
