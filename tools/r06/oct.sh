#!/bin/bash
# the octree with its node arrays addressed as LDS (offsets from one base): new = product, mid = the extractor before that change
# (tools/scratch/lib_mid), old = the r05 extractor (tools/scratch/lib_oldfast); alternating on one box
export GPU_MAX_HW_QUEUES=8
one() { python bench.py --no-cpu --no-ba --no-pcie 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']; print('$1', round(j['value']), round(j['one_stream']['value']), {n: round(k[n]['ms_per_launch_batch'],4) for n in ('fast_cells','describe','octree','blur')})"; }
for i in 1 2 3; do
  one new
  ORBHIP_LIB=$PWD/tools/scratch/lib_mid/liborbslam_hip.so one mid
  ORBHIP_LIB=$PWD/tools/scratch/lib_oldfast/liborbslam_hip.so one old
done
