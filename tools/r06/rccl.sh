#!/bin/bash
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rccl.py -x -q -m gpu -k "c_abi or torchrun or world_size_1" > $O/tests_rccl.txt 2>&1; tail -15 $O/tests_rccl.txt
