#!/bin/bash
# (GPU box) which HOST kernels make a guest wave on their SIMD lose register writes: the lab's three residencies (tools/ubench/residency_lab.hip,
# residency_lds_lab.hip; all on the fp16 matrix pipe, all leave room for the 72-register victim of tools/guest_probe.py) as the neighbour process.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in "h3:chain32 352 regs (256 + 96 acc), 1 wave/SIMD, 32x32x16" "h3p:chain16 196 regs, 2 waves/SIMD, 16x16x32" "lds:chainlds 110 regs, 2 waves/SIMD, 32x32x16"; do
  kind=${k%%:*}; label=${k#*:}
  python tools/residency_lab.py --kinds $kind --no-check --loop-seconds 24 > /dev/null 2>&1 &
  nb=$!
  python tools/guest_probe.py --external 8 "$label" 2>&1 | grep external
  wait $nb
done
