# r06w: after the depth-request fix and the `library_defaults` fixture: the suite with defaults and under the non-default
# configurations that failed in r06v (every failure listed)
mkdir -p gpurun_out/r06w
O=$PWD/gpurun_out/r06w
timeout 1500 python -m pytest tests -m gpu -q > $O/tests_defaults.log 2>&1; echo "defaults: $(grep -E 'passed|failed' $O/tests_defaults.log | tail -1)" | tee $O/tests_other_configurations.log
grep -E "^FAILED|^ERROR" $O/tests_defaults.log | cut -c1-200 | head
for cfg in "quat_check=deferred" "graph_proofs=off" "speculative_binning=off,early_rank=off,tile_order=off" "binning_cache=off,window_matching=off,list_window=off,depth_channel=off" "tile_culling=off,quadrant_masks=on,concurrent_backward=off,group_accumulations=off" "one_call_nodes=off,sort_rank=atomic"; do
  tag=$(echo $cfg | tr '=,' '__')
  SGN_OPTIONS="$cfg" timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_options.py > $O/tests_$tag.log 2>&1
  echo "SGN_OPTIONS=$cfg: $(grep -E 'passed|failed' $O/tests_$tag.log | tail -1)" | tee -a $O/tests_other_configurations.log
  grep -E "^FAILED|^ERROR" $O/tests_$tag.log | cut -c1-200 | head -20
done
