for sp in 0 2 4 8 16; do echo split=$sp; PM_GCL_DW_SPLIT=$sp WHICH=w python tools/gcl_bench.py; done
for sp in 0 4; do echo split=$sp; PM_GCL_DW_SPLIT=$sp python bench.py --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | python tools/benchline.py s$sp; done
