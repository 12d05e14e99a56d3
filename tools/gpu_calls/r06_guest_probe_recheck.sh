#!/bin/bash
# the round-5 reproducer at round-6 sources: the product (all 512 registers claimed) and a variant whose data-gradient kernel does NOT
# claim them (scn_wave.h copied with the clobber removed, mlp_bwd_h3_pd3 rebuilt against it) as the neighbour process
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 python tools/guest_probe.py 6 1 "idle,resident dgrad,whole step" product tools/ubench/lib_h3_noclaim_bwd.so 2>&1 | grep -v "amdgpu.ids\|Warning\|warn"
