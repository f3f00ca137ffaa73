"""stdin: bench.py output; argv[1:]: label words -> one short line (images/s, ms/step, fused, raster fwd / bwd ms)."""
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = j["kernels_avg_ms"]
print(" ".join(sys.argv[1:]), round(j["value"], 1), "ms", round(j["ms_per_step"], 3), "fused",
      round((j.get("fused_path") or {}).get("value", 0), 1), "fwd", k["raster_fwd"], "bwd", k["raster_bwd"],
      "sort", k["sort"], "map", k["map_isect"])
