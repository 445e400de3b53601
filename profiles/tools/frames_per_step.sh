for f in 64 128 256 512 1024 2048; do
python bench.py --steps 40 --warmup 5 --repeats 3 --no-estimated --config-frames 0 --streaming-batches 0 --cpu-seconds 0 --latency-frames 0 --frames-per-step $f --verify-slots 2 --no-exclusive > gpurun_out/fps_$f.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/fps_$f.json').read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print($f, 'G/s', round(d['value']/1e9,3), 'ms/step', round(d['ms_per_step'],4), 'us/frame', round(d['ms_per_step']*1e3/$f,4), {n:round(v['avg_ms']*1e3,1) for n,v in k.items() if 'avg_ms' in v})
PY
done
