mkdir -p gpurun_out/q
python -m pytest tests/test_native_step_gpu.py tests/test_model_gpu.py tests/test_kernels_gpu.py -q -x 2>&1 | tail -2
for lib in polyphemus_amd/variants/libpm_oldgcl.so ""; do PM_LIB_PATH=$lib python tools/phase_times.py; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/q/ph_heads.txt
