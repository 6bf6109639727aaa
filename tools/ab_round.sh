#!/bin/bash
# Same-box A/B of step variants: tools/ab_round.sh <out-dir under gpurun_out> [VAR=VALUE[,VAR=VALUE] ...]
# Every argument after the directory is one variant: comma-separated environment assignments (PM_* switches, PM_LIB_PATH=...
# of a tools/build_variants.py library); "base" = the default step.  Two bench lines per variant, interleaved (A B A B) so
# that clock drift of the box hits all arms alike; prints value, ms/step and the HIP-event averages of the four GCL kernels.
set -u
O=gpurun_out/$1; shift
mkdir -p "$O"
for rep in 1 2; do
  for v in "$@"; do
    tag=$(echo "$v" | tr '/=,' '___')
    if [ "$v" = base ]; then envs=(); else IFS=',' read -ra envs <<< "$v"; fi
    env "${envs[@]}" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2> "$O/ab_${tag}_$rep.err" > "$O/ab_${tag}_$rep.json"
    python - "$O/ab_${tag}_$rep.json" "$v" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[2], "FAILED", e); raise SystemExit
k = r["roofline"]; s = r["roofline_segreduce"]
oth = k["other_gemm_classes"]
print(f"{sys.argv[2]:40s} {r['value']:9.1f} bar-graphs/s {r['ms_per_step']:7.3f} ms | gcl_fwd {s['avg_launch_us']:6.2f} "
      f"dagg {oth.get('gcl_dagg', {}).get('avg_us', 0):6.2f} dw {oth.get('gcl_dw', {}).get('avg_us', 0):6.2f} "
      f"seg_bwd {s.get('backward', {}).get('avg_launch_us', 0):6.2f} us")
PY
  done
done
