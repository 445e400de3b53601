for NC in 1 2 4; do
  echo -n "contexts=$NC: "; python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --frames-per-step 256 --contexts $NC 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); S=d['config']['frame_slots_per_launch']; print(round(d['value']/1e6,1),'M/s', 'us/frame', round(d['ms_per_frame']*1e3,3), {k:round(v['avg_ms']*1e3/S,3) for k,v in d['roofline']['kernels'].items()})"
done
python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --frames-per-step 512 --contexts 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=512 nc=2', round(d['value']/1e6,1),'M/s')"
