for sd in 1234 1236; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --seed $sd 2>/dev/null > gpurun_out/seed_$sd.json; python - gpurun_out/seed_$sd.json $sd <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = r["roofline"]; s = r["roofline_segreduce"]; oth = k["other_gemm_classes"]
print(f"seed {sys.argv[2]} {r['value']:9.1f} {r['ms_per_step']:7.3f} ms | gcl_fwd {s['avg_launch_us']:6.2f} dagg {oth.get('gcl_dagg', {}).get('avg_us', 0):6.2f} dw {oth.get('gcl_dw', {}).get('avg_us', 0):6.2f} seg_bwd {s.get('backward', {}).get('avg_launch_us', 0):6.2f} rows_w {oth.get('gemm_NN_rows_w', {}).get('avg_us', 0):6.2f} tiles {r['config']['row_tiles_per_rank']} N {r['config']['nodes_per_gpu']}")
PY
done
