// ============================================================================
// oracle/ref_dbow2_shim.cpp -- C-ABI shim over the REFERENCE's own DBoW2 classes (test infrastructure only).
//
// This file is compiled TOGETHER WITH the reference's sources, taken where they lie:
//   /root/reference/lib/DBoW2/DBoW2/BowVector.cpp      (BowVector::addWeight / addIfNotExist / normalize)
//   /root/reference/lib/DBoW2/DBoW2/FeatureVector.cpp  (FeatureVector::addFeature)
// by `make -C oracle ref` into oracle/_ref/libdbow2_ref.so (git-ignored, travels to the GPU box as a built file).  These are
// the only translation units of the reference's hot path that compile with the standard library alone: everything else
// needs OpenCV / Eigen / Ceres / Sophus (ScoringObject.cpp and TemplatedVocabulary.h pull in <opencv2/core/core.hpp>;
// no stand-in headers are written).  What it pins: the merge half of Frame::ComputeBoW (src/Frame.cc:322-327 ->
// TemplatedVocabulary::transform lib/DBoW2/DBoW2/TemplatedVocabulary.h:1124-1200): per-feature (word, idf weight, node)
// triples -> BowVector (weights of equal words accumulated in feature order, L1 norm summed in ascending word order,
// division) and FeatureVector (node -> feature indices).  The loop below is that function's body with the single-feature
// descent replaced by its precomputed result.
// ============================================================================
#include <cstdint>
#include "BowVector.h"
#include "FeatureVector.h"

extern "C" {

// returns the number of BowVector entries; *n_fv_nodes = number of FeatureVector entries.  fv_off has n_fv_nodes + 1 entries.
int ref_bow_merge(int n, const int32_t* wid, const double* w, const uint32_t* nid, int norm /*0 = L1, 1 = L2, -1 = none*/,
                  uint32_t* bow_word, double* bow_value, uint32_t* fv_node, uint32_t* fv_off, uint32_t* fv_idx, int* n_fv_nodes) {
  DBoW2::BowVector v;
  DBoW2::FeatureVector fv;
  for (int i = 0; i < n; i++) {
    if (w[i] > 0) {                                    // "not stopped" (TemplatedVocabulary.h:1173-1178)
      v.addWeight((DBoW2::WordId)wid[i], w[i]);
      fv.addFeature((DBoW2::NodeId)nid[i], (unsigned int)i);
    }
  }
  if (norm == 0) v.normalize(DBoW2::L1);
  else if (norm == 1) v.normalize(DBoW2::L2);
  int k = 0;
  for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) { bow_word[k] = it->first; bow_value[k] = it->second; }
  int m = 0; uint32_t pos = 0;
  for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++m) {
    fv_node[m] = it->first; fv_off[m] = pos;
    for (size_t j = 0; j < it->second.size(); j++) fv_idx[pos++] = it->second[j];
  }
  fv_off[m] = pos;
  *n_fv_nodes = m;
  return k;
}

// BowVector::addIfNotExist (IDF / BINARY weighting branch, TemplatedVocabulary.h:1180-1195): first weight of a word wins
int ref_bow_merge_if_not_exist(int n, const int32_t* wid, const double* w, uint32_t* bow_word, double* bow_value) {
  DBoW2::BowVector v;
  for (int i = 0; i < n; i++) if (w[i] > 0) v.addIfNotExist((DBoW2::WordId)wid[i], w[i]);
  int k = 0;
  for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) { bow_word[k] = it->first; bow_value[k] = it->second; }
  return k;
}

}  // extern "C"
