import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(round(d["value"]/1e9,3), round(d["ms_per_step"],4), {k:round(v["avg_ms"]*1e3,1) for k,v in r["kernels"].items()}, d["verified"], d["config"].get("contexts"), d["config"].get("handover"))
