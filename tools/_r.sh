python -m pytest tests/test_kernels_gpu.py -x -q -k "plan or segreduce or gcl_forward" 2>&1 | tail -2
python bench.py --dense --d 512 --batch 64 --steps 5 --warmup 3 --no-cpu-baseline 2>/dev/null | python tools/benchline.py dense
python bench.py --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | python tools/benchline.py d256
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03/prof_dense -o dn -- python3 $GRAFT_REPO_ROOT/bench.py --dense --d 512 --batch 64 --steps 5 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03/prof_dense.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/trace_summary.py gpurun_out/r03/prof_dense/dn_results.db --steps 8 --md | head -16
