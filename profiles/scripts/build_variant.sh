#!/bin/bash
# Build an A/B variant of libsgnrast.so with extra -D flags for ONE translation unit (the other objects are reused from
# the normal build), next to the product library; select it at run time with SGN_RAST_LIB=<path> (sgn_rast/_lib.py).
#   profiles/scripts/build_variant.sh radix_sort ipt8       -DSGN_RS_IPT_LARGE=8      (2048-key sort tiles,  r02m_c.sh)
#   profiles/scripts/build_variant.sh radix_sort ipt32      -DSGN_RS_IPT_LARGE=32     (8192-key sort tiles,  r02m_e.sh)
#   profiles/scripts/build_variant.sh radix_sort rankballot -DSGN_RS_RANK_ATOMIC=0    (ballot-match ranking, r02m_h.sh)
#   profiles/scripts/build_variant.sh radix_sort rankatomic -DSGN_RS_RANK_ATOMIC=1    (the default since r02m, r02m_g.sh)
# The variants are git-ignored (*.so) but travel to the GPU box with gpurun.
set -e
tu=$1; name=$2; shift 2
src=$tu.hip; [ "$tu" = api ] && src=api.cpp
cd "$(dirname "$0")/../../street-gaussians-ns_amd/csrc"
make -j8 > /dev/null
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function -I../../include -I."
[ "$tu" = raster ] && FL="$FL -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FL "$@" -c $src -o /tmp/${tu}_$name.o
objs=""
for o in project sh binning radix_sort raster cubemap loss optim quat exchange api; do
  if [ $o = $tu ]; then objs="$objs /tmp/${tu}_$name.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../sgn_rast/libsgnrast_$name.so $objs
echo "built street-gaussians-ns_amd/sgn_rast/libsgnrast_$name.so"
