#!/bin/bash
# One gpurun call that produces everything profiles/ keeps for a state of the code (run from the repo root on the GPU box):
#   tools/profile_round.sh <out-dir under gpurun_out>
# full GPU suite, smoke, bench line, rocprofv3 kernel trace summary, PMC traffic passes (FETCH_SIZE / WRITE_SIZE separately).
set -u
O=gpurun_out/$1
mkdir -p "$O"
python -m pytest tests -m gpu -q --durations=15 > "$O/pytest_gpu.log" 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1
python bench.py > "$O/bench.json" 2> "$O/bench.err"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$O/trace.log" 2>&1
KT=$(find "$O/trace" -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py "$KT" --steps 13 --md > "$O/kernel_stats.md" 2>> "$O/trace.log"
cp "$(find "$O/trace" -name "*kernel_stats.csv" | head -1)" "$O/kernel_stats.csv"
rm -rf "$O/trace"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_$c" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$O/pmc_$c.log" 2>&1
done
python tools/pmc_traffic.py "$O/pmc_FETCH_SIZE" "$O/pmc_WRITE_SIZE" --workload B256_d256_nb2_L8 > "$O/pmc_traffic.json" 2>> "$O/trace.log"
find "$O" -name "*kernel_trace.csv" -delete
tail -3 "$O/pytest_gpu.log"; tail -2 "$O/smoke.log"; python tools/benchline.py final < "$O/bench.json"
