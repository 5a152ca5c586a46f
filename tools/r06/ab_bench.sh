#!/bin/bash
export GPU_MAX_HW_QUEUES=8
for i in 1 2 3 4; do
  python bench.py --no-cpu --no-ba --no-pcie 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', round(j['value']), j['kernels']['fast_cells']['ms_per_launch_batch'])"
  ORBHIP_LIB=$PWD/tools/scratch/lib_oldfast/liborbslam_hip.so python bench.py --no-cpu --no-ba --no-pcie 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', round(j['value']), j['kernels']['fast_cells']['ms_per_launch_batch'])"
done
