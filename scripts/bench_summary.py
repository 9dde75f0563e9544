import json, sys
d = json.load(open(sys.argv[1]))
print("maps/s %.2f  ms/map %.2f  stages %s" % (d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in d["ms_per_stage"].items()}))
if "cpu_baseline" in d: print("cpu:", d["cpu_baseline"])
for k, v in d["roofline_all"].items():
    print("  %-14s %6.2f ms  %4d launches  achieved %8.2f %s  frac %.3f" % (k, v["ms_per_map"], v["launches_per_map"], v["achieved"], v["unit"], v["frac"]))
