#!/bin/bash
# (GPU box) the residency lab: check + times of every kind, ablation builds, counters (one per pass, --kernel-trace only).
#   bash tools/ubench/residency_lab_run.sh r05lab [kinds]      -> gpurun_out/<tag>/lab_residency.txt
set -u
TAG=${1:-r05lab}; KINDS=${2:-h3,h3p,h3-infer,h3p-infer}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O/pmc
{
echo "# chain of eight 256 -> 256 ReLU layers, 786 432 samples, three fp16 products per product (tools/ubench/residency_lab.hip)"
echo "# -- check against fp64 + times, plain build"
python tools/residency_lab.py --samples 786432 --reps 8 --kinds $KINDS
for v in nostore nostream noepi bare; do
  lib=tools/ubench/libresidency_lab_$v.so
  [ -f $lib ] || continue
  echo "# -- ablation build: $v (results are wrong, times are not)"
  python tools/residency_lab.py --samples 786432 --reps 6 --kinds ${ABL_KINDS:-h3,h3p} --no-check --lib $lib
done
} > $O/lab_residency.txt 2>&1
for k in ${PMC_KINDS:-h3 h3p}; do
  for c in ${COUNTERS:-GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS}; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/${k}_$c -o x -- python tools/residency_lab.py --samples 786432 --reps 3 --kinds $k --no-check > /dev/null 2> $O/pmc_${k}_$c.err
    f=$(find $O/pmc/${k}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" "$k" "$c" >> $O/lab_pmc.txt <<'PY'
import csv, sys
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == sys.argv[3] and "chain" in r["Kernel_Name"]]
print("%-10s %-28s %s" % (sys.argv[2], sys.argv[3], " ".join("%.4g" % v for v in vals)))
PY
    rm -rf $O/pmc/${k}_$c
  done
done
rm -rf $O/pmc
cat $O/lab_residency.txt; cat $O/lab_pmc.txt
