# r03g: depth as a fourth channel of the colour pass + the launch diet's bin-edge tail with a capped grid
mkdir -p gpurun_out/r03g
O=gpurun_out/r03g
timeout 900 python -m pytest tests/test_gpu_depth_channel.py -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_fused.py tests/test_gpu_calltrace.py -x -q > $O/tests.log 2>&1; grep -E "passed|failed|^E " $O/tests.log | tail -6
B="python bench.py --no-cpu-baseline"
timeout 300 $B > $O/bench_default.json 2>$O/err; python profiles/scripts/benchline.py "default" < $O/bench_default.json
timeout 300 $B --with-depth > $O/bench_depth.json 2>$O/err; python profiles/scripts/benchline.py "with-depth" < $O/bench_depth.json
SGN_DEPTH_CHANNEL=off timeout 300 $B --with-depth > $O/bench_depth_off.json 2>$O/err; python profiles/scripts/benchline.py "with-depth two-pass" < $O/bench_depth_off.json
timeout 300 $B --scene-graph > $O/bench_sg.json 2>$O/err; python profiles/scripts/benchline.py "scene-graph" < $O/bench_sg.json
SGN_DEPTH_CHANNEL=off timeout 300 $B --scene-graph > $O/bench_sg_off.json 2>$O/err; python profiles/scripts/benchline.py "scene-graph two-pass" < $O/bench_sg_off.json
python - <<'P'
import json
for f in ("bench_default","bench_depth","bench_depth_off"):
    j=json.loads(open("gpurun_out/r03g/%s.json"%f).read().strip().splitlines()[-1])
    e=j.get("eval_images_per_s") or {}
    print(f, "eval", round(e.get("value",0),1), "rgb-only", round((e.get("rgb_alpha_only") or {}).get("value",0),1), "fused", round((e.get("fused_path") or {}).get("value",0),1), "| fused train", round((j.get("fused_path") or {}).get("value",0),1), "| kernels", {k:v for k,v in j["kernels_avg_ms"].items() if k in ("tile_bins","raster_fwd","pack_records")})
P
