"""stdin: the JSON line of `python bench.py` -> value, ms per step and the CostRegNet stage times (A/B runs of the switches)."""
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d["stages_ms"]
print(d["value"], d["ms_per_step"], {k: s[k] for k in sorted(s) if k.startswith("costreg.") or k == "absmax"})
