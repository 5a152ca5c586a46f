#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ba.py tests/test_gpu_ba_structures.py -x -q -m gpu 2>&1 | tail -2
for st in band covis dense; do
  rm -rf $O/lbaprof_$st
  ORBHIP_BENCH_STRUCTURE=$st rocprofv3 --kernel-trace --stats --output-format rocpd -d $O/lbaprof_$st -o run -- timeout 600 python tools/ba_batch_thr.py 64:1 > $O/lbaprof_$st.log 2>&1 || tail -5 $O/lbaprof_$st.log
  db=$(find $O/lbaprof_$st -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py $db $O/wg_${st}.csv > /dev/null && echo "== $st" && python tools/kstats_print.py $O/wg_${st}.csv | head -3
  rm -rf $O/lbaprof_$st
done
python tools/bsolve_ab.py | head -4
