for B in 8 16 32 64 128 256; do
  echo -n "B=$B: "; python bench.py --steps 40 --warmup 5 --cpu-seconds 0 --frames-per-step $B --unique-frames 16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1),'M/s', 'ms/frame', round(d['ms_per_frame']*1e3,3),'us', {k:round(v['avg_ms']*1e3/d['config']['frames_per_step'],3) for k,v in d['roofline']['kernels'].items()})"
done
