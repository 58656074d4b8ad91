import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], "clips/s", d["ms_per_step"], "ms")
print(json.dumps(d["rccl"], indent=1))
