#!/bin/bash
# alternating A/B on one box: the library in tools/scratch/lib_oldfast/ ("old") against the product ("new"); bench.py's front-end legs only
export GPU_MAX_HW_QUEUES=8
for i in 1 2 3; do
  python bench.py --no-cpu --no-ba --no-pcie 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']; print('new', round(j['value']), round(j['one_stream']['value']), {n: round(k[n]['ms_per_launch_batch'],3) for n in ('fast_cells','describe','octree','blur')})"
  ORBHIP_LIB=$PWD/tools/scratch/lib_oldfast/liborbslam_hip.so python bench.py --no-cpu --no-ba --no-pcie 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']; print('old', round(j['value']), round(j['one_stream']['value']), {n: round(k[n]['ms_per_launch_batch'],3) for n in ('fast_cells','describe','octree','blur')})"
done
