#!/bin/bash
# On the GPU box: socket power and shader clock (rocm-smi, ~1 s period) while the default training step loops.
#   bash tools/power_trace.sh [bench.py flags] > gpurun_out/r03/power_trace.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 1500 --warmup 5 --no-cpu --no-extras "$@" > /tmp/power_bench.json 2> /dev/null &
BPID=$!
echo "# rocm-smi --showpower --showclocks sampled while \`python bench.py --steps 1500 --warmup 5 --no-cpu --no-extras $*\` runs"
echo "sample  sclk_MHz  socket_W"
i=0
while kill -0 $BPID 2> /dev/null; do
  i=$((i + 1))
  out=$(rocm-smi --showpower --showclocks 2> /dev/null)
  sclk=$(echo "$out" | grep -i "sclk" | head -1 | grep -o "([0-9]*Mhz)" | tr -dc '0-9')
  pw=$(echo "$out" | grep -i "power (W)" | head -1 | sed 's/.*: *//' | cut -d. -f1)
  printf "%5d  %8s  %8s\n" $i "${sclk:-?}" "${pw:-?}"
  sleep 0.7
done
wait $BPID
python - <<'PY'
import json
d = json.loads(open('/tmp/power_bench.json').read().strip().splitlines()[-1])
print("# bench line of that loop: %.0f rays/s, %.3f ms per step" % (d["value"], d["ms_per_step"]))
PY
