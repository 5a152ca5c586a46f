#!/bin/bash
# per-kernel times of single C4-size LocalBA solves (every dispatch a kernel node of its own: ORBHIP_BA_GRAPH=0)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
rm -rf $O/p_single
ORBHIP_BA_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format rocpd -d $O/p_single -o run -- timeout 300 python tools/lba_one.py > $O/p_single.log 2>&1
db=$(find $O/p_single -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py $db $O/localba_single_kernel_stats.csv > /dev/null && python tools/kstats_print.py $O/localba_single_kernel_stats.csv | head -24
rm -rf $O/p_single
tail -3 $O/p_single.log
python tools/lba_one.py 2>&1 | tail -2
