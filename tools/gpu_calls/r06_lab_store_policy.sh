#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06l; mkdir -p $O
LAB_CHECK_ROWS=97 timeout 600 python tools/residency_lab.py --samples 786432 --reps 10 --kinds h3,h3-st-default,h3-st-nt,h3-st-sc0sc1nt,h3-st-sc1nt,h3-st-sc0nt,h3-st-sc1,h3-st-sc0sc1,h3,h3-st-nt,h3-st-sc0sc1nt,h3-st-sc1nt,h3-st-sc0nt > $O/lab_store_policy.txt 2>&1
python - <<'PY'
import json
for ln in open("gpurun_out/r06l/lab_store_policy.txt"):
    if ln.startswith("{"):
        r = json.loads(ln); print("%-16s best %.3f median %.3f ok %s" % (r["kind"], r["ms_best"], r["ms_median"], r.get("ok")))
    else:
        print(ln.rstrip()[:200])
PY
