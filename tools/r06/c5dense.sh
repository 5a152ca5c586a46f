#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
ORBHIP_BENCH_STRUCTURE=dense timeout 600 python tools/gba_c5.py 500 50000 250000 8 2>&1 | tail -4
rm -rf $O/gbaprof_dense
ORBHIP_BENCH_STRUCTURE=dense rocprofv3 --kernel-trace --stats --output-format rocpd -d $O/gbaprof_dense -o run -- timeout 600 python tools/gba_c5.py 500 50000 250000 8 > $O/gbaprof_dense.log 2>&1 || tail -5 $O/gbaprof_dense.log
db=$(find $O/gbaprof_dense -name "*.db" 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py $db $O/gba_c5_dense_kernel_stats.csv && python tools/kstats_print.py $O/gba_c5_dense_kernel_stats.csv | head -12
rm -rf $O/gbaprof_dense
