#!/bin/bash
# does the SYNTHETIC guest (tools/ubench/guest_write_lab.hip) lose writes beside the PRODUCT's data-gradient kernel without the claim?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in scnerf_amd/libscnerf_hip.so tools/ubench/lib_h3_noclaim_bwd.so; do
  SCNERF_HIP_LIB=$lib timeout 120 python tools/host_loop.py dgrad 22 > /dev/null 2>&1 &
  hp=$!
  sleep 10
  timeout 60 tools/ubench/guest_write_lab guest 8 "synthetic guest beside dgrad of $(basename $lib)"
  wait $hp
done
