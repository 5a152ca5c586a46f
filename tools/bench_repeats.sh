#!/bin/bash
# N runs of the default bench.py on one box: value, one-stream rate, host-fed fraction, LocalBA figures of every run (the run-to-run spread).
# usage: bash tools/bench_repeats.sh [N=5] > profiles/<round>_bench_repeats.txt
cd "$(dirname "$0")/.."
N=${1:-5}
for i in $(seq $N); do
  timeout 900 python bench.py 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
l = j['localba']; c = l['roofline']['cases']
print('run %s: value %.0f frames/s, one stream %.0f, host-fed %.3f of its bound, k_fast_cells %.3f ms; LocalBA %.0f solves/s batched (c4_batched %.3f), %.2f ms single; c5 %.3f, eight sub-maps %.3f' % (
      '$i', j['value'], j['one_stream']['value'], j['pcie_inclusive']['frac_of_bound'], j['roofline']['ms_per_launch'], l['localba_solves_per_s'], c['c4_batched']['frac'],
      l['localba_ms_per_solve_latency'], c['c5']['frac'], c['c5_batched8']['frac']))"
done
