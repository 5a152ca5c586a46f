#!/bin/bash
python tools/bsolve_ab.py 2>&1 | tail -2
bash tools/r06/single_kstats.sh 2>&1 | head -3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
for v in 1 0; do
  rm -rf $O/p_c5
  ORBHIP_BA_BSOLVE_WAVES=$v ORBHIP_BA_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format rocpd -d $O/p_c5 -o run -- timeout 300 python tools/gba_c5.py 500 50000 250000 10 > $O/p_c5.log 2>&1
  db=$(find $O/p_c5 -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py $db $O/p_c5.csv > /dev/null && python tools/kstats_print.py $O/p_c5.csv | head -3
  rm -rf $O/p_c5
done
