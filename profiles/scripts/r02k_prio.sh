# priority of the auxiliary stream (the backward's long-walk kernel) x library variant, street and metric scenes
V=$GRAFT_REPO_ROOT/street-gaussians-ns_amd/sgn_rast/variants/$1
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
for rep in 1 2; do
for v in base var; do
  if [ $v = var ]; then export SGN_RAST_LIB=$V; else unset SGN_RAST_LIB; fi
  for pr in 0 -1; do
    export SGN_AUX_PRIORITY=$pr
    timeout 300 python bench.py --street --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python profiles/scripts/benchline.py street $v prio $pr
    timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python profiles/scripts/benchline.py metric $v prio $pr
  done
done
done
