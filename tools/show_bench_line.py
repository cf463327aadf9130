import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline ms", l["ms_per_step"], "frac", l["roofline"]["frac"], "box frac", l["roofline"]["frac_of_box_store_stream"], l["roofline"]["zone_walks_in_process"])
s=l["secondary"]
for k in ("held_pair","pipelined","rollout","graph","host_gather"):
    if k in s: print(k, {a:b for a,b in s[k].items() if a not in ("what",)})
for k,v in s.get("workloads",{}).items(): print(k, json.dumps(v))
print("errors", s.get("errors"))
print(l["cpu_baseline"].get("configs0_python_literal"))
