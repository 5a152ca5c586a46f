#!/bin/bash
# front-end half of tools/run_profiles.sh (after a front-end change: the BA-side files of the full run stay valid)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=r06; O=gpurun_out/$R; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
rm -rf $O/benchprof
rocprofv3 --kernel-trace --stats --output-format rocpd -d $O/benchprof -o run -- timeout 600 python bench.py --no-cpu --no-ba --no-pcie --streams 1 --no-pipelined --steps 3 --warmup 1 > $O/bench_traced.json 2> $O/benchprof.log || tail -5 $O/benchprof.log
db=$(find $O/benchprof -name "*.db" 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py $db $O/bench_kernel_stats.csv && python tools/kstats_print.py $O/bench_kernel_stats.csv | head -10
timeout 600 python tools/fast_phase_prof.py 2>/dev/null | python -c "import sys,json; t=sys.stdin.read(); i=t.index('{'); print(json.dumps(json.loads(t[i:])))" > $O/fast_phase_prof.json; cat $O/fast_phase_prof.json
timeout 1200 bash tools/run_pmc.sh > $O/pmc.log 2>&1 || tail -5 $O/pmc.log
for i in 1 2 3; do timeout 600 python bench.py --no-cpu 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); lb=j['localba']; print('run', j['value'], 'one stream', j.get('one_stream',{}).get('frames_per_s'), 'fast', j['kernels']['fast_cells']['ms_per_launch_batch'], 'localba', lb['localba_solves_per_s'], lb['localba_covis_solves_per_s'], lb['localba_dense_solves_per_s'], 'single ms', lb['localba_ms_per_solve_latency'], 'c5 ms/it', lb['globalba_500kf_ms_per_iteration'], 'loop', lb['globalba_loop_500kf_ms_per_iteration'])"; done > $O/bench_repeats.txt 2>&1; cat $O/bench_repeats.txt
PROFILES_OUT=$PWD/$O/collected2 python tools/collect_profiles.py $R | tail -12
rm -rf gpurun_out/pmc_x_insts gpurun_out/pmc_x_active gpurun_out/pmc_x_fetch gpurun_out/pmc_x_write $O/benchprof
