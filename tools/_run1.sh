cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/n1
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_optimizer.py -m gpu -q -x > gpurun_out/n1/tests.txt 2>&1; echo rc=$? >> gpurun_out/n1/tests.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-extras > gpurun_out/n1/bench.json 2> gpurun_out/n1/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/n1/trace -o t -- python bench.py --steps 10 --warmup 3 --no-cpu --no-extras > gpurun_out/n1/bench_traced.json 2> gpurun_out/n1/trace.err
python tools/pmc_summary.py --help > /dev/null 2>&1
ls gpurun_out/n1/trace | head
tail -3 gpurun_out/n1/tests.txt
