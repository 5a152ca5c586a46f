#!/bin/bash
# k_ba_eval<0> without the scratch copy of Jc / r: per-kernel times of a 64-problem band batch, the GPU BA tests, throughput
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
rm -rf $O/lbaprof_band
rocprofv3 --kernel-trace --stats --output-format rocpd -d $O/lbaprof_band -o run -- timeout 600 python tools/ba_batch_thr.py 64:1 > $O/lbaprof_band.log 2>&1 || tail -5 $O/lbaprof_band.log
db=$(find $O/lbaprof_band -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py $db $O/localba_batch64_kernel_stats_new.csv && python tools/kstats_print.py $O/localba_batch64_kernel_stats_new.csv | head -12
rm -rf $O/lbaprof_band
timeout 1200 python -m pytest tests/test_gpu_ba.py tests/test_gpu_ba_structures.py -x -q -m gpu 2>&1 | tail -2
python tools/ba_batch_thr.py 64:12 2>&1 | tail -3
