#!/bin/bash
# Timing experiments on the resident kernels: variants of libscnerf_hip.so whose mlp_fwd_h3 / mlp_bwd_h3 objects are
# rebuilt with tools/lab/scn_lab.h in front of the product's scn_lab.h and -DSCN_H3_<flag> (the pd 3 translation units),
# the other objects taken from the product
# build.  The variants travel to the GPU box (*.so is git-ignored, not gpurun-ignored); there:
#   SCNERF_HIP_LIB=tools/ubench/lib_h3_<tag>.so python tools/bench_h3.py --only-resident-train
#   tools/ablate_h3.sh TAG "-DSCN_H3_NO_STORE" [TAG2 "..."] ...
set -e
cd "$(dirname "$0")/.."
python -m scnerf_amd.csrc.build > /dev/null
others=$(ls scnerf_amd/csrc/_obj/*.o | grep -v "mlp_fwd_h3_pd3.o\|mlp_bwd_h3_pd3.o")
while [ $# -gt 1 ]; do
  tag=$1; defs=$2; shift 2
  (
  for f in mlp_fwd_h3_pd3 mlp_bwd_h3_pd3; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $defs \
      -Itools/lab -Iscnerf_amd/csrc/device -Iscnerf_amd/csrc -Iinclude -c scnerf_amd/csrc/$f.hip -o /tmp/ablate_${tag}_$f.o &
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/lib_h3_${tag}.so $others /tmp/ablate_${tag}_mlp_fwd_h3_pd3.o /tmp/ablate_${tag}_mlp_bwd_h3_pd3.o
  echo "built tools/ubench/lib_h3_${tag}.so"
  ) &
done
wait
