#!/bin/bash
O=gpurun_out/r06; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_all.txt 2>&1; tail -3 $O/tests_all.txt
python tools/lba_latency.py 2>&1 | tail -1
python tools/chol_wg_prof.py 2>&1 | tail -8
python tools/chol_persist_prof.py 2>&1 | grep -A8 ns_per_phase
