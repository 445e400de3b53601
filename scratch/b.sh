for cfg in "1024 1" "2048 1"; do set -- $cfg
  echo -n "B=$1 contexts=$2: "; python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --frames-per-step $1 --contexts $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); S=d['config']['frame_slots_per_launch']; print(round(d['value']/1e6,1),'M/s', 'us/frame', round(d['ms_per_frame']*1e3,3), {k:round(v['avg_ms']*1e3/S,3) for k,v in d['roofline']['kernels'].items()})"
done
