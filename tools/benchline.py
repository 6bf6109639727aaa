import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); o = dict(r["roofline"]["other_gemm_classes"]); 
dom = r["roofline"]["kernel"]; 
print(sys.argv[1] if len(sys.argv) > 1 else "", r["value"], r["ms_per_step"], "DOM", dom[7:30], r["roofline"]["avg_launch_us"], {k[5:]: v["avg_us"] for k, v in o.items() if "planes" in k}, "seg", r["roofline_segreduce"]["avg_launch_us"], r["roofline_segreduce"]["backward"]["avg_launch_us"])
