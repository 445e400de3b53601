python -m pytest tests -q -m gpu --tb=short -x 2>&1 | tail -3
for k in 16 24 32 48 64; do
  echo -n "K1MAX=$k: "; MLD_K1MAX=$k python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1),'M/s', {k:round(v['avg_ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})"
done
