import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"])
r = j["roofline"]
print({k: r[k] for k in ("bound", "achieved", "peak", "frac", "traffic") if k in r})
print(r.get("request_floor"))
print(r.get("traffic_commit"), r.get("traffic_stale"))
for leg in ("repeats", "mixed"):
    print(leg, j[leg].get("ms_per_step"), j[leg].get("value"), j[leg].get("failed"))
print(j["cpu_baseline"])
print(j.get("end_to_end", {}).get("value"), j.get("host_to_host", {}).get("value"))
