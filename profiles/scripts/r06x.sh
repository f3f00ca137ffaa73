mkdir -p gpurun_out/r06x
O=$PWD/gpurun_out/r06x
timeout 1500 python -m pytest tests -m gpu -q > $O/tests_defaults.log 2>&1; echo "defaults: $(grep -E 'passed|failed' $O/tests_defaults.log | tail -1)" | tee $O/tests_other_configurations.log
grep -E "^FAILED|^ERROR" $O/tests_defaults.log | cut -c1-200 | head
for cfg in "binning_cache=off,window_matching=off,list_window=off,depth_channel=off" "tile_culling=off,quadrant_masks=off,early_rank=on,depth_channel=on"; do
  tag=$(echo $cfg | tr '=,' '__')
  SGN_OPTIONS="$cfg" timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_options.py > $O/tests_$tag.log 2>&1
  echo "SGN_OPTIONS=$cfg: $(grep -E 'passed|failed' $O/tests_$tag.log | tail -1)" | tee -a $O/tests_other_configurations.log
  grep -E "^FAILED|^ERROR" $O/tests_$tag.log | cut -c1-200 | head -20
done
