#!/bin/bash
# (GPU box) candidate (b), bare, beside the bare builds of the other two residencies; counters for (b).
set -u
TAG=${1:-r05lds}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O/pmc
L=tools/ubench/libresidency_lab
{
echo "# (b) bare: activations' cut planes resident in LDS, 8 waves on 64 x 64 sub-blocks, ring of 2 x 16 KB, one barrier per K slab"
python tools/residency_lab.py --reps 8 --kinds lds
echo "# (b) bare, all-zero weights"; python tools/residency_lab.py --reps 8 --kinds lds --zero
echo "# (b) without the ring's stream (slots keep their first slabs)"; python tools/residency_lab.py --reps 8 --kinds lds --no-check --lib ${L}_nostream.so
echo "# (b) without the barrier (racy)"; python tools/residency_lab.py --reps 8 --kinds lds --no-check --lib ${L}_nobarrier.so
echo "# (b) without stream and barrier: MFMAs + fragment reads only"; python tools/residency_lab.py --reps 8 --kinds lds --no-check --lib ${L}_barest.so
echo "# today's residency and (a), bare builds (MFMAs + fragment reads + chunk barriers; real fragments in every buffer)"
python tools/residency_lab.py --reps 8 --kinds h3,h3p --no-check --lib ${L}_bare.so
echo "# ... and without their chunk barriers"; python tools/residency_lab.py --reps 8 --kinds h3,h3p --no-check --lib ${L}_barest.so
} > $O/lds.txt 2>&1
for c in GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/$c -o x -- python tools/residency_lab.py --reps 3 --kinds lds --no-check > /dev/null 2> /dev/null
  f=$(find $O/pmc/$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$c" >> $O/lds_pmc.txt <<'PY'
import csv, sys
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == sys.argv[2] and "chain" in r["Kernel_Name"]]
print("lds        %-28s %s" % (sys.argv[2], " ".join("%.4g" % v for v in vals)))
PY
done
rm -rf $O/pmc
cat $O/lds.txt $O/lds_pmc.txt
