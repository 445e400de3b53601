import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
def find(o):
    if isinstance(o,dict):
        if "two_contexts" in o: return o
        for v in o.values():
            r=find(v)
            if r: return r
r=find(d); t=r["two_contexts"]["classify"]
print(r["sequences"], round(r["ms_per_step"],4), [round(x,3) for x in t["ms_per_step_runs"]], [round(x,3) for x in t["submit_ms_per_step_runs"]])
