"""Which runtime call blocks the submitting thread?  Reads a rocprofv3 --hip-runtime-trace --hsa-trace CSV pair and prints
every HIP call longer than `ms` milliseconds inside the benchmark's timed phase together with the HSA calls nested in it.
   rocprofv3 --hip-runtime-trace --hsa-trace --output-format csv -d /tmp/t -- python3 bench.py --only-config 5 --leg 64t
   python profiles/tools/find_host_stall.py /tmp/t 2.0"""
import csv
import glob
import sys

root, ms = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
hip = list(csv.DictReader(open(glob.glob(root + "/*/*hip_api_trace.csv")[0])))
hsa_files = glob.glob(root + "/*/*hsa_api_trace.csv")
hsa = list(csv.DictReader(open(hsa_files[0]))) if hsa_files else []
t0 = int(hip[0]["Start_Timestamp"])
for r in hip:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if b - a < ms * 1e6:
        continue
    print(f"{(a - t0) / 1e9:8.3f} s  {r['Function']:32s} {(b - a) / 1e6:8.2f} ms  thread {r['Thread_Id']}")
    inner = [(int(h["End_Timestamp"]) - int(h["Start_Timestamp"]), h["Function"]) for h in hsa
             if h["Thread_Id"] == r["Thread_Id"] and int(h["Start_Timestamp"]) >= a and int(h["End_Timestamp"]) <= b]
    for d, f in sorted(inner, reverse=True)[:6]:
        if d > 50_000:
            print(f"{'':14s}{f:40s} {d / 1e6:8.2f} ms")
