# r04u: A/B of an EXPERIMENT that was measured and removed: the objects group's alpha-only gradient riding on the main
# pass's reverse walk (one compare + one multiply-add per entry into the main v_alpha; the group's stretch below the
# main pass's start walked as a prologue on the group's own list; SGN_GROUP_RIDE switched it).  Result
# (profiles/r04u_group_ride_ab.log, r04u_kernel_stats_group_ride.md): 559 images/s riding vs 607 with the group's
# own reverse walk over its own list (the committed form): the objects' walks reach far below the main pass's (they do
# not saturate behind an opaque background), so nearly the whole group walk is prologue, and as a scalar-chase prologue
# inside the main kernels it lost the LDS-batched path and the long-walk kernel's balance (long-walk kernel 590 us).
mkdir -p gpurun_out/r04u
O=$PWD/gpurun_out/r04u
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_groups.py -q -x 2>&1 | tail -3 > $O/tests_groups.log; tail -1 $O/tests_groups.log
for g in "1 1" "1 0" "0 1" "1 1" "1 0"; do
  set -- $g
  SGN_GROUP_ACC=$1 SGN_GROUP_RIDE=$2 python bench.py --scene-graph --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python profiles/scripts/benchline.py group_acc=$1 ride=$2 | tee -a $O/ab.log
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra --scene-graph --path fused > /tmp/kt.log 2>&1; python $REPO/profiles/summarize_rocpd.py kernels $(find /tmp/kt -name "p_results.db" | head -1) > $O/kernel_stats_sg_fused.md; head -8 $O/kernel_stats_sg_fused.md | cut -c1-140
