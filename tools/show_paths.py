import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if "error" in d: print(d); continue
    print(d["name"], d["frames_per_s"], d["roofline_frac"], [(k["name"],k["ms_per_frame"]) for k in d["kernels"]], d["latency_mode"]["roofline_frac"], d["latency_mode"]["kernels_ms"])
