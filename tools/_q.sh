mkdir -p gpurun_out/q
python -m pytest tests/test_kernels_gpu.py tests/test_graphs_gpu.py -q -x 2>&1 | tail -2
python -m pytest tests/test_native_step_gpu.py -q -x 2>&1 | tail -2
for v in 15 15; do PM_SIDE_STREAM=$v python tools/phase_times.py; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/q/phases7.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/q/trace -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/q/trace_bench.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/step_sequence.py gpurun_out/q/trace > gpurun_out/q/step_sequence4.txt
rm -rf gpurun_out/q/trace
sed -n 28,42p gpurun_out/q/step_sequence4.txt | cut -c1-100
