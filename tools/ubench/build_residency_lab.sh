#!/bin/bash
# Builds tools/ubench/libresidency_lab[_<tag>].so (gfx950) from residency_lab.hip + residency_lds_lab.hip; the library travels
# to the GPU box with the snapshot (*.so is git-ignored, not gpurun-ignored).
#   tools/ubench/build_residency_lab.sh                      the plain build
#   tools/ubench/build_residency_lab.sh TAG "-DSCN_H3_NO_STORE" [TAG2 "..."]    ablation builds (tools/lab/scn_lab.h)
set -e
cd "$(dirname "$0")/../.."
build() {
  tag=$1; defs=$2
  out=tools/ubench/libresidency_lab${tag:+_$tag}.so
  objs=""
  for f in residency_lab residency_lds_lab; do
    [ -f tools/ubench/$f.hip ] || continue
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $defs ${SAVE_TEMPS:+-save-temps=obj} \
      -Itools/lab -Itools/ubench -Iscnerf_amd/csrc/device -Iscnerf_amd/csrc -Iinclude -Wall -Wno-unused-function \
      -c tools/ubench/$f.hip -o /tmp/${f}_${tag:-plain}.o &
    objs="$objs /tmp/${f}_${tag:-plain}.o"
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs
  echo "built $out"
}
if [ $# -eq 0 ]; then build "" ""; fi
while [ $# -gt 1 ]; do build "$1" "$2"; shift 2; done
