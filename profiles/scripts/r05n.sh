#!/bin/bash
# round 5, call n: kernel-by-kernel timeline of ONE steady-state step of the default drop-in path (eager quats check) and of
# the deferred-check form: where does the device wait for the host?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
REPO=$PWD
O=$REPO/gpurun_out/r05n; mkdir -p $O
cd /tmp
for mode in eager deferred; do
  rm -rf /tmp/kt
  SGN_QUAT_CHECK=$mode rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-extra > /tmp/kt_$mode.log 2>&1
  db=$(find /tmp/kt -name "p_results.db" | head -1)
  python $REPO/profiles/summarize_rocpd.py timeline $db project_fwd 100 > $O/timeline_$mode.md 2>&1
  python $REPO/profiles/summarize_rocpd.py timeline $db project_fwd 101 > $O/timeline_${mode}_next.md 2>&1
  head -1 $O/timeline_$mode.md
  tail -2 /tmp/kt_$mode.log | cut -c1-300
done
