import json
l=json.loads([x for x in open("gpurun_out/r6_rehearsal2.jsonl").read().splitlines() if x.startswith("{")][-1])
print(l["metric"][:80]); print(l["n_gpus"], l["value"], l["ms_per_step"], l["config"]["process_group"])
print({k:v for k,v in l["secondary"]["host_gather"].items() if k not in ("what","rehearsal")})
print([ (r["rank"], r["kernel_ms"], r["spread"]) for r in l["roofline"]["per_rank"]])
