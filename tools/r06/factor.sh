#!/bin/bash
O=gpurun_out/r06; mkdir -p $O
REBUILD=1 python tools/factor_ab.py > $O/factor_ab.txt 2>&1; tail -5 $O/factor_ab.txt
timeout 1500 python -m pytest tests/test_gpu_ba.py tests/test_gpu_ba_structures.py tests/test_gpu_ba_full_size.py -x -q -m gpu > $O/tests_ba.txt 2>&1; tail -4 $O/tests_ba.txt
python tools/lba_latency.py > $O/lba_latency.txt 2>&1; tail -1 $O/lba_latency.txt
