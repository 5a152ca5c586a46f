#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for v in base ev_NO_E ev_NO_HC ev_NO_PT; do
  rm -rf $O/p_$v
  L=$PWD/tools/scratch/lib_$v/liborbslam_hip.so; [ $v = base ] && L=$PWD/ceres_mono_orb_slam2_amd/lib/liborbslam_hip.so
  ORBHIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format rocpd -d $O/p_$v -o run -- timeout 300 python tools/ba_batch_thr.py 64:1 > $O/p_$v.log 2>&1
  db=$(find $O/p_$v -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py $db $O/p_$v.csv > /dev/null && echo "== $v" && python tools/kstats_print.py $O/p_$v.csv | grep "k_ba_eval<0>"
  rm -rf $O/p_$v
done
