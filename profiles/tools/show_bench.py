import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("value",d["value"]/1e9,"ms_per_step",d["ms_per_step"],"verified",d["verified"])
print({k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if not isinstance(v,(dict,list)) and k!="formula_note"})
print("estimated",{k:v for k,v in d["plane_estimated"].items() if k not in ("kernels_ms_per_launch",)})
c=d["configs"]
print("2k",{k:v for k,v in c["2"]["near_returns"].items() if k not in ("workload","result_types","roofline")})
print("2k roofline",c["2"]["near_returns"]["roofline"])
print("3n", c["3"]["near_returns"]["modes"]["c0_dispose"])
print("5", {k:v for k,v in c["5"].items() if k!="batched"})
for k,v in c["5"]["batched"].items(): print("5b",k,v)
l=d["latency"]
print("lat", l["ms_per_frame_median"], l["stride32"]["ms_per_frame_median"], l["pinned_source"].get("explanation"))
print("lat est", l["estimated"]["ransac"]["ms_per_frame_median"], l["estimated"]["ransac"]["stride32"]["ms_per_frame_median"], l["estimated"]["semantic"]["ms_per_frame_median"], l["estimated"]["semantic"]["stride32"]["ms_per_frame_median"])
print("process", l["process"])
s=d["streaming"]; print("streaming", {k:v for k,v in s.items() if k!="stride32"}, s["stride32"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
