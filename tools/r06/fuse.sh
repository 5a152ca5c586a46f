#!/bin/bash
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_localmapping.py tests/test_gpu_compat_cpp.py -x -q -m gpu -s > $O/tests_fuse.txt 2>&1; grep -n 'passed\|failed\|search_and_fuse\|fuse_many\|FAIL\|Error\|assert' $O/tests_fuse.txt | tail -12
