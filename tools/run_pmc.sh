#!/bin/bash
# The four rocprofv3 PMC passes over tools/frontend_only.py (separate runs, --pmc only: no trace domains), ALL at the bench's
# batch of 256 frames; rocpd databases under gpurun_out/pmc_x_{insts,active,fetch,write}; tools/pmc_extract.py <round> turns
# them into profiles/<round>_pmc_{valu,traffic}.json.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
# (ORBHIP_OVERLAP_BLUR is an experiments-build knob: the product library always overlaps the blur)
# (the experiments build must be as new as the sources: a stale one lacks symbols the Python mirror binds)
if [ ! -f tools/exp_lib/liborbslam_hip.so ] || [ -n "$(find ceres_mono_orb_slam2_amd/csrc include -newer tools/exp_lib/liborbslam_hip.so -type f | head -1)" ]; then bash tools/build_experiments.sh > /dev/null || exit 1; fi
export ORBHIP_LIB=$PWD/tools/exp_lib/liborbslam_hip.so
export ORBHIP_OVERLAP_BLUR=0      # per-kernel counters and cycles: every kernel alone on its stream
B=${PMC_BATCH:-256}
run() {  # tag counters...
  tag=$1; shift 1
  rm -rf gpurun_out/pmc_x_$tag
  rocprofv3 --pmc "$@" --output-format rocpd -d gpurun_out/pmc_x_$tag -o run -- python tools/frontend_only.py $B 3 > gpurun_out/pmc_x_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_x_$tag.log; exit 1; }
  f=$(find gpurun_out/pmc_x_$tag -name "*.db" | head -1); [ -n "$f" ] && [ "$f" != "gpurun_out/pmc_x_$tag/run_results.db" ] && mv "$f" gpurun_out/pmc_x_$tag/run_results.db
  ls -la gpurun_out/pmc_x_$tag | tail -2
}
run insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA
run active GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
echo $B > gpurun_out/pmc_x_batch.txt
