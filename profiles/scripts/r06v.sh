# r06v: the whole GPU suite under non-default configurations, every failure listed (no -x)
mkdir -p gpurun_out/r06v
O=$PWD/gpurun_out/r06v
for cfg in "quat_check=deferred" "sort_rank=atomic" "graph_proofs=off" "speculative_binning=off,early_rank=off,tile_order=off" "binning_cache=off,window_matching=off,list_window=off,depth_channel=off" "tile_culling=off,quadrant_masks=on,concurrent_backward=off,group_accumulations=off"; do
  tag=$(echo $cfg | tr '=,' '__')
  SGN_OPTIONS="$cfg" timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_options.py > $O/tests_$tag.log 2>&1
  echo "SGN_OPTIONS=$cfg: $(grep -E 'passed|failed' $O/tests_$tag.log | tail -1)" | tee -a $O/tests_other_configurations.log
  grep -E "^FAILED|^ERROR" $O/tests_$tag.log | cut -c1-200 | head -40
done
