#!/bin/bash
# End-of-round evidence: rocprofv3 kernel-trace stats of the bench command, the LocalBA batch, the C++ API latencies and the
# 2-rank shared-GPU dry run.  Everything lands under gpurun_out/r02/ ; copy what is judged into profiles/.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
rocprofv3 --kernel-trace --stats --output-format rocpd -d $O/benchprof -o run -- python bench.py --no-cpu --steps 3 --warmup 1 > $O/benchprof.log 2>&1 || tail -5 $O/benchprof.log
python tools/rocpd_stats.py $(find $O/benchprof -name "*.db" | head -1) $O/bench_kernel_stats.csv | head -30
rocprofv3 --kernel-trace --stats --output-format rocpd -d $O/lbaprof -o run -- python tools/ba_batch_thr.py 16:8 > $O/lbaprof.log 2>&1 || tail -5 $O/lbaprof.log
python tools/rocpd_stats.py $(find $O/lbaprof -name "*.db" | head -1) $O/localba_batch16_kernel_stats.csv | head -12
g++ -O2 -std=c++17 -I include tools/cpp/api_latency.cpp -o /tmp/api_latency -L ceres_mono_orb_slam2_amd/lib -lorbslam_hip
LD_LIBRARY_PATH=ceres_mono_orb_slam2_amd/lib:/opt/rocm/lib /tmp/api_latency > $O/api_latency_cpp.json; cat $O/api_latency_cpp.json
python tools/api_latency.py 2>/dev/null | tail -1 > $O/api_latency_py.json; cat $O/api_latency_py.json
ORBHIP_BENCH_SHARED_GPU=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu > $O/bench_2rank_shared.json 2> $O/bench_2rank.err || tail -5 $O/bench_2rank.err
python tools/fast_phase_prof.py 2>/dev/null | tail -9 > $O/fast_phase_prof.json
bash tools/run_pmc.sh > $O/pmc.log 2>&1 || tail -5 $O/pmc.log
python tools/pmc_extract.py r02 | tail -6
