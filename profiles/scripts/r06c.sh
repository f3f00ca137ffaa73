# r06c: GPU suite on the tree with the semantics switches, the parity-pin kit, the image bound
mkdir -p gpurun_out/r06c
O=$PWD/gpurun_out/r06c
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|^FAILED|^E   " $O/tests.log | tail -20
python -m pytest tests/test_gpu_grad_at_size.py -m gpu -q -s -k "whole_image" 2>&1 | grep "image bound" > $O/image_bound.log; cat $O/image_bound.log | cut -c1-260 | head -40
