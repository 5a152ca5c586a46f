#!/bin/bash
# A/B on one box: r05's orb_extractor.hip against the current one, alternating, the pipelined two-stream front-end rate
export GPU_MAX_HW_QUEUES=8
for i in 1 2 3; do
  echo "old:"; ORBHIP_LIB=$PWD/tools/scratch/lib_oldfast/liborbslam_hip.so python tools/frontend_ab.py 20 256 2 2>&1 | tail -2 | cut -c1-220
  echo "new:"; python tools/frontend_ab.py 20 256 2 2>&1 | tail -2 | cut -c1-220
done
python bench.py --no-cpu --no-ba --no-pcie 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench value', j['value'], 'fast', j['kernels']['fast_cells'])"
ORBHIP_LIB=$PWD/tools/scratch/lib_oldfast/liborbslam_hip.so python bench.py --no-cpu --no-ba --no-pcie 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench value (r05 extractor)', j['value'], 'fast', j['kernels']['fast_cells'])"
