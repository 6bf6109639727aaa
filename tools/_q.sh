mkdir -p gpurun_out/q
for m in 15 13 15 13; do PM_SIDE_STREAM=$m python tools/phase_times.py | head -2; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/q/mask13.txt
for m in 15 13; do PM_SIDE_STREAM=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python tools/benchline.py "mask=$m"; done | tee -a gpurun_out/q/mask13.txt
