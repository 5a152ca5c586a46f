#!/bin/bash
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ba.py tests/test_gpu_ba_stop.py tests/test_gpu_ba_structures.py tests/test_gpu_ba_full_size.py tests/test_gpu_essential_graph.py tests/test_gpu_concurrency.py -x -q -m gpu 2>&1 | tail -3
bash tools/r06/evalscratch.sh 2>&1 | grep -v "passed\|^\.\|amdgpu.ids"
ORBHIP_BA_BSOLVE_WAVES=0 python tools/ba_batch_thr.py 64:12 2>&1 | tail -1
python tools/ba_batch_thr.py 64:12 2>&1 | tail -1
ORBHIP_BA_BSOLVE_WAVES=0 python tools/ba_batch_thr.py 64:12 2>&1 | tail -1
python tools/ba_batch_thr.py 64:12 2>&1 | tail -1
ORBHIP_LIB=$PWD/tools/scratch/lib_bw_prof/liborbslam_hip.so python tools/bw_prof.py 100 10000 50000 10 > $O/bsolve_sky4_phases_c4.txt 2>&1
ORBHIP_LIB=$PWD/tools/scratch/lib_bw_prof/liborbslam_hip.so python tools/bw_prof.py 500 50000 250000 10 > $O/bsolve_sky4_phases_c5.txt 2>&1
python tools/bsolve_ab.py > $O/bsolve_ab.txt 2>&1; tail -1 $O/bsolve_ab.txt
