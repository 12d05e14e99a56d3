#!/bin/bash
# (GPU box) the guest-write reproducer in both settings: one process / two streams, then host and guest as two PROCESSES.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=tools/ubench/guest_write_lab
S=${1:-6}
echo "== one process, two streams"
timeout 120 $L $S
echo "== two processes (host process loops one mode; 'host regs' in the guest's line is not meaningful there)"
for m in "0:host 424 regs, 1 wave/SIMD, epilogue + stream" "3:host 424 regs, 1 wave/SIMD, MFMAs only" "2:host 248 regs, narrow tail" "1:host 512 regs (claimed)"; do
  mode=${m%%:*}; label=${m#*:}
  timeout 60 $L host $mode $((S + 6)) > /dev/null 2>&1 &
  hp=$!
  sleep 2
  timeout 60 $L guest $S "2 proc: $label"
  wait $hp
done
