#!/bin/bash
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_extractor.py tests/test_gpu_track.py -x -q -m gpu > $O/tests_ext.txt 2>&1; grep -n 'passed\|failed' $O/tests_ext.txt | tail -3
GPU_MAX_HW_QUEUES=8 python tools/frontend_ab.py 20 256 2 2>&1 | tail -2
timeout 900 python tools/fuzz_extractor.py 300 2>&1 | tail -3
