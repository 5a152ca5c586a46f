#!/bin/bash
# HIP API + kernel + copy trace of tests/test_gpu_concurrency.py (Tracking beside LocalBA + GlobalBA threads), analysed on the box
# (the database is ~150 MB): bash tools/trace_concurrency.sh [0|1]   - the argument is ORBHIP_WS_COPY_KERNEL (0 = hipMemcpyAsync staging).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export ORBHIP_WS_COPY_KERNEL=${1:-1}
# (the switch is read by experiments builds only: bash tools/build_experiments.sh first)
[ "$ORBHIP_WS_COPY_KERNEL" = 0 ] && export ORBHIP_LIB=$PWD/tools/exp_lib/liborbslam_hip.so
O=gpurun_out/conc; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format rocpd -d $O/prof -o run -- python -m pytest tests/test_gpu_concurrency.py -q -s > $O/log.txt 2>&1
echo "ORBHIP_WS_COPY_KERNEL=$ORBHIP_WS_COPY_KERNEL"; grep -E "Tracking|passed|failed" $O/log.txt
db=$(find $O/prof -name "*.db" | head -1); python tools/trace_delay_analysis.py $db 2>&1 | tail -24
rm -rf $O/prof
