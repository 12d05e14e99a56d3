#!/bin/bash
# round 6, fifth GPU call: the re-judged adversarial cases, the bench's own counter passes, the coarse / fine forward launches in isolation
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
(timeout 1500 python -m pytest "tests/test_gpu_kernels.py::test_resident_layers_are_fp32_grade" "tests/test_gpu_kernels.py::test_resident_data_gradients_follow_the_fused_chain_row_by_row" "tests/test_gpu_camera.py::test_combined_config3_step_gradients_with_decisions_aligned" tests/test_bench_line.py tests/test_hot_kernels_no_scratch.py -m gpu -q --timeout 900 -s > $O/gpu_tests.txt 2>&1; echo rc=$? >> $O/gpu_tests.txt)
cp gpurun_out/parity_r06.json $O/parity_r06.json 2>/dev/null
grep -E "passed|failed|^FAILED|^E  |two RCCL ranks|rc=" $O/gpu_tests.txt | head -40
timeout 200 python tools/bench_coarse_stage.py --iters 40 > $O/coarse_stage_alone.json 2>&1; tail -1 $O/coarse_stage_alone.json
timeout 200 python tools/bench_fine_stage.py > $O/fine_stage_alone.json 2>&1; tail -2 $O/fine_stage_alone.json
