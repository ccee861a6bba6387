"""Print the interesting fields of bench.py's JSON line (reads stdin)."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d["value"]), d["ms_per_step"], {k: round(v, 3) for k, v in d.get("kernel_ms_raw", {}).items()}, d.get("single_cloud_latency_ms"))
