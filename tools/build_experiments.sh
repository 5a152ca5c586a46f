#!/bin/bash
# Builds the library with -DORBHIP_EXPERIMENTS (the knobs and kernels of measured-and-rejected alternatives: ORBHIP_BA_PERSIST=2 /
# k_chol_persist_2l, ORBHIP_BA_OB, ORBHIP_BA_2L_CLASSIC, ORBHIP_BA_GRAPH, ORBHIP_OVERLAP_BLUR, ORBHIP_EXTRACT_CONE, ORBHIP_DESC_XCD,
# ORBHIP_FAST_XCD, ORBHIP_OCT_LEVELS, ORBHIP_MATCH_SPLIT, ORBHIP_TRACK_ROUNDS, -DBLUR_COL_SLIDE=1) into tools/exp_lib/; the tools that
# need them run with ORBHIP_LIB=tools/exp_lib/liborbslam_hip.so.  Extra flags (e.g. -DORBHIP_CHOL_PROF) are passed through.
# The product library (ceres_mono_orb_slam2_amd/lib, __graft_entry__.build) does not contain any of it.
cd "$(dirname "$0")/.."
O=tools/exp_lib; mkdir -p $O
C=ceres_mono_orb_slam2_amd/csrc
objs=""
for f in capi_common orb_extractor orb_matcher orb_frame orb_vocab ba_solver orb_track orb_localmap orb_comm; do
  X=""; [ $f = orb_extractor ] && X="-mllvm -amdgpu-mfma-vgpr-form"      # (as __graft_entry__.EXTRA_FLAGS)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DORBHIP_EXPERIMENTS -Itools $X "$@" -c $C/$f.hip -o $O/$f.o || exit 1
  objs="$objs $O/$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/liborbslam_hip.so $objs && echo "built $O/liborbslam_hip.so"
# the product sources with k_ba_schur's in-kernel phase stamps (tools/schur_prof.py): only ba_solver differs, no experiment switches
L=ceres_mono_orb_slam2_amd/lib
if [ -f $L/capi_common.o ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DORBHIP_SCHUR_PROF -c $C/ba_solver.hip -o $O/ba_solver_sprof.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/liborbslam_hip_sprof.so $L/capi_common.o $L/orb_extractor.o $L/orb_matcher.o $L/orb_frame.o $L/orb_vocab.o $O/ba_solver_sprof.o $L/orb_track.o $L/orb_localmap.o $L/orb_comm.o && echo "built $O/liborbslam_hip_sprof.so"
fi
