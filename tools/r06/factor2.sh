#!/bin/bash
O=gpurun_out/r06; mkdir -p $O
REBUILD=1 python tools/factor_ab.py > $O/factor_ab.txt 2>&1; tail -5 $O/factor_ab.txt
