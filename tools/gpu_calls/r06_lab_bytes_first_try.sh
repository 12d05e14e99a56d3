#!/bin/bash
# round 6, third GPU call: guest-write reproducer (LDS-heavy host), the gate-compaction lab (producer side), the fixed GPU tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O/pmc
timeout 300 bash tools/ubench/guest_write_lab_run.sh 6 > $O/guest_write_lab.txt 2>&1
{
echo "# chain of eight 256 -> 256 ReLU layers, 786 432 samples (tools/ubench/residency_lab.hip): today's epilogue (h3) vs gate-compacted stores (h3-compact)"
python tools/residency_lab.py --samples 786432 --reps 8 --kinds h3,h3-compact,h3-infer
} > $O/lab_bytes.txt 2>&1
for k in h3 h3-compact; do
  for c in WRITE_SIZE FETCH_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/${k}_$c -o x -- python tools/residency_lab.py --samples 786432 --reps 3 --kinds $k --no-check > /dev/null 2> $O/pmc_${k}_$c.err
    f=$(find $O/pmc/${k}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" "$k" "$c" >> $O/lab_bytes_pmc.txt <<'PY'
import csv, sys
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == sys.argv[3] and "chain" in r["Kernel_Name"]]
print("%-12s %-28s %s" % (sys.argv[2], sys.argv[3], " ".join("%.5g" % v for v in vals)))
PY
    rm -rf $O/pmc/${k}_$c
  done
done
rm -rf $O/pmc $O/pmc_*.err
(timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_camera.py tests/test_gpu_render.py tests/test_pack_cache.py tests/test_randoms.py -m gpu -q > $O/gpu_tests.txt 2>&1; echo rc=$? >> $O/gpu_tests.txt)
cp gpurun_out/parity_r06.json $O/parity_r06.json 2>/dev/null
cat $O/guest_write_lab.txt; cat $O/lab_bytes.txt; cat $O/lab_bytes_pmc.txt; tail -15 $O/gpu_tests.txt
