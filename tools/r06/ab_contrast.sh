#!/bin/bash
export GPU_MAX_HW_QUEUES=8
for c in 0.25 0.5; do
  for i in 1 2; do
    echo "contrast $c new:"; ORBHIP_AB_CONTRAST=$c python tools/frontend_ab.py 20 256 2 2>&1 | tail -2 | cut -c1-200
    echo "contrast $c old:"; ORBHIP_AB_CONTRAST=$c ORBHIP_LIB=$PWD/tools/scratch/lib_oldfast/liborbslam_hip.so python tools/frontend_ab.py 20 256 2 2>&1 | tail -2 | cut -c1-200
  done
done
