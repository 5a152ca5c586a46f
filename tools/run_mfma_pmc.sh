#!/bin/bash
# FP64-MFMA counter passes over the three forms of the reduced solve ("r04 MFMA PMC pass over the new kernels"):
#   batch   k_chol_wg            64 C4-size problems per lockstep batch (tools/ba_batch_thr.py 64:1)
#   single  k_chol_persist       single C4-size LocalBA solves (tools/lba_one.py)
#   c5      k_chol_persist       one GlobalBA at C5 size, 10 iterations (tools/gba_c5.py)
#   batch_covis / batch_dense     the 64-problem batch on covisibility-structured / dense reduced systems (round 6)
# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (no other trace domain), on the
# experiments build with ORBHIP_BA_GRAPH=0 so that every dispatch is a kernel node of its own; tools/mfma_c5.py turns each database
# into profiles/<round>_mfma_<tag>.json.   usage: bash tools/run_mfma_pmc.sh r04
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=${1:-r06}; O=gpurun_out/$R; mkdir -p $O
# (the experiments build must be as new as the sources: a stale one lacks symbols the Python mirror binds)
if [ ! -f tools/exp_lib/liborbslam_hip.so ] || [ -n "$(find ceres_mono_orb_slam2_amd/csrc include -newer tools/exp_lib/liborbslam_hip.so -type f | head -1)" ]; then bash tools/build_experiments.sh > /dev/null || exit 1; fi
export ORBHIP_LIB=$PWD/tools/exp_lib/liborbslam_hip.so ORBHIP_BA_GRAPH=0
run() { tag=$1; shift
  rm -rf $O/mfma_$tag
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format rocpd -d $O/mfma_$tag -o run -- timeout 300 "$@" > $O/mfma_$tag.log 2>&1 || tail -5 $O/mfma_$tag.log
  db=$(find $O/mfma_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python tools/mfma_c5.py $db $O/mfma_$tag.json | grep -A7 "k_chol_wg\|k_chol_persist\"\|k_chol_persist_blk\|factorisation_and" | head -40
  rm -rf $O/mfma_$tag
}
run batch python tools/ba_batch_thr.py 64:1
run single python tools/lba_one.py
run c5 python tools/gba_c5.py 500 50000 250000 10
# round 6: the same batch with the structures a map of the reference has (synth.make_ba_graph_covis)
ORBHIP_BENCH_STRUCTURE=covis run batch_covis python tools/ba_batch_thr.py 64:1
ORBHIP_BENCH_STRUCTURE=dense run batch_dense python tools/ba_batch_thr.py 64:1
