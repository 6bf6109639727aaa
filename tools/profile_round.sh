#!/bin/bash
# One gpurun call that produces everything profiles/ keeps for a state of the code (run from the repo root on the GPU box):
#   tools/profile_round.sh <out-dir under gpurun_out>
# full GPU suite, smoke, bench lines (configs[1]; d = 512 = training.json; dense shard of configs[4]; LMD16 = configs[2]),
# rocprofv3 kernel-trace summaries (d = 256, d = 512, dense), PMC traffic passes (FETCH_SIZE / WRITE_SIZE separately; d = 256
# and d = 512).
set -u
O=gpurun_out/$1
mkdir -p "$O"
python -m pytest tests -m gpu -q --durations=15 > "$O/pytest_gpu.log" 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1
python bench.py --d 512 --steps 10 --warmup 4 --no-cpu-baseline --no-other-workloads > "$O/bench_d512.json" 2> "$O/bench_d512.err"
python bench.py --dense --d 512 --batch 64 --steps 5 --warmup 3 --no-cpu-baseline > "$O/bench_dense.json" 2> "$O/bench_dense.err"
python bench.py --batch 64 --n-bars 16 --steps 10 --warmup 4 --no-cpu-baseline > "$O/bench_lmd16.json" 2> "$O/bench_lmd16.err"
export TMPDIR=/tmp
trace() {   # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_$name" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads "$@" > "$O/trace_$name.log" 2>&1
  KT=$(find "$O/trace_$name" -name "*kernel_trace.csv" | head -1)
  python tools/trace_summary.py "$KT" --steps 0 --md > "$O/kernel_stats_$name.md" 2>> "$O/trace_$name.log"
  cp "$(find "$O/trace_$name" -name "*kernel_stats.csv" | head -1)" "$O/kernel_stats_$name.csv"
  python tools/step_sequence.py "$KT" > "$O/step_sequence_$name.txt" 2>> "$O/trace_$name.log"
  rm -rf "$O/trace_$name"
}
trace d256
trace d512 --d 512
trace dense --dense --d 512 --batch 64
pmc() {     # name, workload key, bench args...
  local name=$1 key=$2; shift 2
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_${name}_$c" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads "$@" > "$O/pmc_${name}_$c.log" 2>&1
  done
  python tools/pmc_traffic.py "$O/pmc_${name}_FETCH_SIZE" "$O/pmc_${name}_WRITE_SIZE" --workload "$key" > "$O/pmc_traffic_$name.json" 2>> "$O/trace_d256.log"
  find "$O" -name "*kernel_trace.csv" -delete
  rm -rf "$O/pmc_${name}_FETCH_SIZE" "$O/pmc_${name}_WRITE_SIZE"
}
sq() {      # name, bench args...: matrix-core / issue counters of the tile kernels (SQ block; two passes of four counters)
  local name=$1; shift
  local i=0
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/sq_${name}_$i" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads "$@" > "$O/sq_${name}_$i.log" 2>&1
  done
  { echo "# rocprofv3 --pmc (two passes), bench.py --steps 2 --warmup 1 $*: per-launch averages; SQ_VALU_MFMA_BUSY_CYCLES counts cycles,"
    echo "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* quad-cycles summed over waves, SQ_BUSY_CU_CYCLES per CU (MI355X_MICROARCH.md)"
    for i in 1 2; do python tools/pmc_kernels.py "$O/sq_${name}_$i" k_gcl_fwd k_gcl_dagg k_gcl_dw k_wide k_rows_w k_rows_tn k_segreduce_bwd k_unembed k_bar_fwd k_bar_bwd; done
  } > "$O/sq_counters_$name.txt" 2>> "$O/trace_d256.log"
  rm -rf "$O/sq_${name}_1" "$O/sq_${name}_2"
}
sq d256
sq d512 --d 512
sq dense --dense --d 512 --batch 64
# the bench batch has 256 row tiles = one per CU; seven of the eight seeds 1234..1241 have 257-261
for sd in 1234 1235 1236 1237; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --seed $sd 2>/dev/null | python tools/benchline.py "seed $sd"; done > "$O/bench_seeds.txt" 2>&1
for m in 0 65535; do PM_SIDE_STREAM=$m python tools/phase_times.py; done > "$O/phase_times.txt" 2>&1
pmc d256 B256_d256_nb2_L8
pmc d512 B256_d512_nb2_L8 --d 512
# the bench line once more with this run's PMC files in place (bench.py reports roofline.traffic only when their source digest matches)
cp "$O/pmc_traffic_d256.json" profiles/pmc_traffic.json; cp "$O/pmc_traffic_d512.json" profiles/pmc_traffic_d512.json
python bench.py > "$O/bench.json" 2> "$O/bench.err"
tail -3 "$O/pytest_gpu.log"; tail -2 "$O/smoke.log"; python tools/benchline.py final < "$O/bench.json"; python tools/benchline.py d512 < "$O/bench_d512.json"; python tools/benchline.py dense < "$O/bench_dense.json"; python tools/benchline.py lmd16 < "$O/bench_lmd16.json"
