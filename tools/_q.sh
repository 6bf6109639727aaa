mkdir -p gpurun_out/q
python -m pytest tests/test_kernels_gpu.py -q -x -k "bn_small or bn_train" 2>&1 | tail -3
python -m pytest tests/test_native_step_gpu.py tests/test_model_gpu.py -q -x 2>&1 | tail -2
python tools/phase_times.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/q/ph_bn.txt
