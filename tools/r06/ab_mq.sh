#!/bin/bash
# k_fast_cells with two cells per wave sharing one survivor queue (tools/scratch/lib_fastmq: orb_extractor.hip + tools/experiments/fast_shared_queue.patch, -DFAST_CPW=2) against
# the product (one cell per wave): exactness on the extractor tests, then alternating bench.py front-end legs on one box
export GPU_MAX_HW_QUEUES=8
ORBHIP_LIB=$PWD/tools/scratch/lib_fastmq/liborbslam_hip.so timeout 900 python -m pytest tests/test_gpu_extractor.py -q -m gpu -x 2>&1 | tail -3
one() { python bench.py --no-cpu --no-ba --no-pcie 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']; print('$1', round(j['value']), round(j['one_stream']['value']), {n: round(k[n]['ms_per_launch_batch'],4) for n in ('fast_cells','describe','octree','blur')})"; }
for i in 1 2 3 4; do
  one cpw1
  ORBHIP_LIB=$PWD/tools/scratch/lib_fastmq/liborbslam_hip.so one cpw2
done
