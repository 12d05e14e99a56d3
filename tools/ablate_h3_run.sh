#!/bin/bash
# (GPU box) times the ablation builds of tools/ablate_h3.sh against the product build, two rounds:
#   bash tools/ablate_h3_run.sh > gpurun_out/r03/ablation_h3.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "# resident kernels at P = 786 432 (4096 rays x 192 samples), ms per launch; variants remove one ingredient (results are wrong, times are not)"
echo "# variant             forward(train)   data gradients   | sustained over 1.5 s: forward ms / GHz / W / Mcycles,  data gradients ms / GHz / W / Mcycles"
for round in 1 2; do
  for v in product noepi nostore nostream nope bare; do
    if [ $v = product ]; then lib=""; else lib="tools/ubench/lib_h3_$v.so"; fi
    out=$(SCNERF_HIP_LIB=$lib python tools/bench_h3.py --only-resident-train 2> /dev/null | tail -1)
    python - "$v" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
def sus(k):
    t = d.get(k + " sustained")
    return "%6.3f %5.2f %5d %6.2f" % (t["ms"], t["ghz"], t["watt"], t["ms"] * t["ghz"]) if t else "-"
print("%-20s %10.3f %16.3f   | %s   %s" % (sys.argv[1], d["resident train"], d["resident dgrad"], sus("resident train"), sus("resident dgrad")))
PY
  done
done
