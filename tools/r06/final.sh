#!/bin/bash
O=gpurun_out/r06; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -5 $O/smoke.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_final.txt 2>&1; grep -n 'passed\|failed' $O/tests_final.txt | tail -2
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err; tail -c 600 $O/bench_final.json
