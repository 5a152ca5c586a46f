#!/bin/bash
# rocprofv3 PMC passes over ONE lockstep batch of 64 C4-size LocalBA problems (tools/ba_batch_thr.py 64:1:1; experiments build with
# ORBHIP_BA_GRAPH=0 so that every launch is a dispatch of its own): instruction counts, activity, FETCH_SIZE, WRITE_SIZE in separate
# runs (--pmc only); tools/ba_pmc_summary.py prints / writes the per-kernel table.  usage: bash tools/run_ba_pmc.sh [round]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=${1:-r05}; O=gpurun_out/$R; mkdir -p $O
if [ ! -f tools/exp_lib/liborbslam_hip.so ] || [ -n "$(find ceres_mono_orb_slam2_amd/csrc include -newer tools/exp_lib/liborbslam_hip.so -type f | head -1)" ]; then bash tools/build_experiments.sh > /dev/null || exit 1; fi
export ORBHIP_LIB=$PWD/tools/exp_lib/liborbslam_hip.so ORBHIP_BA_GRAPH=0
run() { tag=$1; shift
  rm -rf gpurun_out/bapmc_$tag
  rocprofv3 --pmc "$@" --output-format rocpd -d gpurun_out/bapmc_$tag -o run -- timeout 300 python tools/ba_batch_thr.py 64:1:1 > gpurun_out/bapmc_$tag.log 2>&1 || tail -5 gpurun_out/bapmc_$tag.log
  f=$(find gpurun_out/bapmc_$tag -name "*.db" | head -1); [ -n "$f" ] && [ "$f" != "gpurun_out/bapmc_$tag/run_results.db" ] && mv "$f" gpurun_out/bapmc_$tag/run_results.db
}
run insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run active GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
python tools/ba_pmc_summary.py $O/ba_batch64_pmc.json
rm -rf gpurun_out/bapmc_insts gpurun_out/bapmc_active gpurun_out/bapmc_fetch gpurun_out/bapmc_write
