# r05a: first GPU call of round 5 — the new tests (scene graph under the data-parallel harness, checked row exchange,
# known answers, default ballot ranking), the whole GPU suite, smoke, the driver's command, the scene-graph step with two
# ranks sharing the GPU over gloo (drop-in and fused), and a first counter campaign with the round-5 tooling.
mkdir -p gpurun_out/r05a
O=$PWD/gpurun_out/r05a
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_dp_scene_graph.py tests/test_gpu_dp.py tests/test_known_answers.py tests/test_gpu_sort_stability.py -m gpu -x -q > $O/tests_new.log 2>&1; grep -E "passed|failed|^E |^FAILED" $O/tests_new.log | tail -8
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^E |^FAILED" $O/tests.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; tail -2 $O/bench_$name.err | cut -c1-300; }
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python profiles/scripts/benchline.py driver20 < $O/bench_driver.json; tail -2 $O/bench_driver.err | cut -c1-300
run default
run sg --scene-graph
run sg_forcedp --scene-graph --force-dp
dp2() { name=$1; shift; SGN_DP_BACKEND=gloo SGN_BENCH_SHARE_GPU=1 timeout 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python profiles/scripts/benchline.py $name < $O/bench_$name.json; tail -3 $O/bench_$name.err | cut -c1-300; }
dp2 sg_dp2_gloo --scene-graph
dp2 sg_dp2_gloo_fused --scene-graph --path fused
python - <<'PY'
import json
for n in ("sg_dp2_gloo", "sg_dp2_gloo_fused", "sg_forcedp", "driver"):
    try:
        d = json.loads(open(f"gpurun_out/r05a/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, "dp:", json.dumps(d["config"].get("dp", {}))[:900])
        if n == "driver":
            r = d["roofline"]
            print("roofline:", {k: r[k] for k in ("bound", "limiter", "kernel", "achieved", "frac", "traffic")}, r["pmc"], r["budget_rate"]["over_peak"])
            print("per_kernel:", json.dumps(r["per_kernel"]))
            print("repeat:", d["repeat"], "sort:", d["config"]["sort_ranking"])
    except Exception as e:
        print(n, "ERR", repr(e))
PY
sha256sum street-gaussians-ns_amd/csrc/raster.hip > $O/raster_hip.sha256
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $REPO/profiles/microbench/valu_rates.hip -o /tmp/valu_rates 2> $O/microbench_build.err
PA="SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
PB="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_WAIT_INST_ANY"
rocprofv3 --pmc $PA -d /tmp/cal_a -o p -- /tmp/valu_rates --calib > /tmp/cal_a.log 2>&1
python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/cal_a -name "p_results.db" | head -1) > $O/calib_pmc_a.md
pmc() { suf=$1; shift; BENCH="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra $@"
  for pair in "a:$PA" "b:$PB" "fetch_size:FETCH_SIZE" "write_size:WRITE_SIZE"; do
    nm=${pair%%:*}; ctr=${pair#*:}; rm -rf /tmp/pm
    rocprofv3 --pmc $ctr -d /tmp/pm -o p -- $BENCH > /tmp/pm.log 2>&1
    python $REPO/profiles/summarize_rocpd.py pmc $(find /tmp/pm -name "p_results.db" | head -1) > $O/pmc_${nm}${suf}.md
  done; echo "pmc$suf done"; grep -c raster $O/pmc_a${suf}.md; }
pmc ""
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > /tmp/kt.log 2>&1
python $REPO/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $O/kernel_stats_dropin.md
python $REPO/profiles/summarize_rocpd.py gaps $(find /tmp/kt -name "p_results.db" | head -1) > $O/gaps_dropin.md 2>&1
tail -1 $O/kernel_stats_dropin.md; head -1 $O/gaps_dropin.md
