#!/bin/bash
# (GPU box) follow-up probes of the residency lab: what do the stream and the stores really cost, and is the socket's power
# limit part of it?   bash tools/ubench/residency_lab_probe.sh r05probe
set -u
TAG=${1:-r05probe}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
L=tools/ubench/libresidency_lab
{
echo "# plain"; python tools/residency_lab.py --reps 8 --kinds h3,h3p --no-check
echo "# plain, all-zero weights and biases"; python tools/residency_lab.py --reps 8 --kinds h3,h3p --zero
for v in nostore nostream nostore_nostream noepi bare nobarrier; do
  [ -f ${L}_$v.so ] || continue
  echo "# $v"; python tools/residency_lab.py --reps 8 --kinds h3,h3p --no-check --lib ${L}_$v.so
done
echo "# bare, all-zero weights"; python tools/residency_lab.py --reps 8 --kinds h3,h3p --zero --lib ${L}_bare.so
} > $O/probe.txt 2>&1
# power / clock while each variant loops for 6 s
for spec in "plain:" "zero:--zero" "bare:--lib ${L}_bare.so" "nostream:--lib ${L}_nostream.so" "nostore:--lib ${L}_nostore.so"; do
  name=${spec%%:*}; flags=${spec#*:}
  for k in h3 h3p; do
    python tools/residency_lab.py --reps 3 --kinds $k --no-check --loop-seconds 6 $flags > $O/loop_${name}_$k.json 2>&1 &
    PID=$!
    sleep 2.5
    for i in 1 2 3 4; do
      out=$(rocm-smi --showpower --showclocks 2> /dev/null)
      sclk=$(echo "$out" | grep -i "sclk" | head -1 | grep -o "([0-9]*Mhz)" | tr -dc '0-9')
      pw=$(echo "$out" | grep -i "power (W)" | head -1 | sed 's/.*: *//' | cut -d. -f1)
      echo "$name $k sclk_MHz ${sclk:-?} socket_W ${pw:-?}" >> $O/power.txt
      sleep 0.6
    done
    wait $PID
    python - $O/loop_${name}_$k.json "$name $k" >> $O/power.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%s looped: ms best %.3f median %.3f" % (sys.argv[2], d["ms_best"], d["ms_median"]))
PY
  done
done
cat $O/probe.txt $O/power.txt
