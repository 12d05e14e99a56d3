#!/bin/bash
# Timing experiments: builds variants of libscnerf_hip.so with -DSCN_<flag> defines into tools/ubench/
# (*.so is git-ignored but travels to the GPU box); run each there with
#   SCNERF_HIP_LIB=tools/ubench/lib_<tag>.so python tools/microbench.py
#   tools/ablate.sh TAG "-DSCN_X=1 -DSCN_Y=2" [TAG2 "..."] ...
set -e
cd "$(dirname "$0")/.."
while [ $# -gt 1 ]; do
  tag=$1; defs=$2; shift 2
  objs=""
  for f in scnerf_amd/csrc/*.hip; do
    o=/tmp/ablate_${tag}_$(basename $f .hip).o
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $defs \
      -Iscnerf_amd/csrc/device -Iscnerf_amd/csrc -Iinclude -c $f -o $o &
    objs="$objs $o"
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/lib_${tag}.so $objs
  echo "built tools/ubench/lib_${tag}.so"
done
