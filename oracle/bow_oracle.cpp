// ============================================================================
// oracle/bow_oracle.cpp -- CPU restatement of DBoW2's TemplatedVocabulary::transform as Frame::ComputeBoW uses it
// (reference lib/DBoW2/DBoW2/TemplatedVocabulary.h:1124-1200 batch transform, :1218-1260 single-feature descent,
// lib/DBoW2/DBoW2/BowVector.cpp:34-84 addWeight / normalize, FeatureVector::addFeature, FORB::distance
// lib/DBoW2/DBoW2/FORB.cpp:81-101; src/Frame.cc:322-327 calls it with levelsup = 4; ORBVocabulary = TF_IDF + L1_NORM).
//
// TEST INFRASTRUCTURE ONLY (see oracle/orb_oracle.cpp header for the rule).
//
// PARITY STATUS: the MERGE half (orc_bow_merge: BowVector::addWeight / normalize, FeatureVector::addFeature) is PINNED against
// the reference's own BowVector.cpp / FeatureVector.cpp, compiled from /root/reference into oracle/_ref/libdbow2_ref.so
// (`make -C oracle ref`; tests/test_oracle_bow_ref.py + tests/golden/bow_merge_ref.npz).  The DESCENT half stays "parity
// unpinned": TemplatedVocabulary.h and FORB.cpp need OpenCV (FORB's descriptor type is a cv::Mat; absent, no stand-ins), and
// the vocabulary blob (Vocabulary/ORBvoc.txt) is not shipped.  The
// tree is taken in flattened form (what loadFromTextFile builds in m_nodes): per node a 256-bit descriptor, its children
// in order, a word id (leaves) and an idf weight.  Pinned in tests/test_oracle_bow.py by a numpy brute-force descent.
// ============================================================================
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

namespace {
inline int hamming(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 8; i++) {                       // the bit trick of FORB::distance, 32 bits at a time
    uint32_t x, y;
    std::memcpy(&x, a + 4 * i, 4); std::memcpy(&y, b + 4 * i, 4);
    uint32_t v = x ^ y;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    d += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return d;
}
}  // namespace

extern "C" {

// single-feature descent (:1218-1260).  nid defaults to 0 (root) when the leaf is reached above nid_level (the reference
// leaves it uninitialised there; ORB vocabularies have all leaves at depth L).
void orc_bow_descend(const uint8_t* node_desc, const uint32_t* child_off, const uint32_t* children, const int32_t* word_id,
                     const double* weight, int L, int levelsup, const uint8_t* feat, int32_t* wid, double* w, uint32_t* nid) {
  const int nid_level = L - levelsup;
  *nid = 0;
  uint32_t final_id = 0;
  int current_level = 0;
  do {
    ++current_level;
    const uint32_t lo = child_off[final_id], hi = child_off[final_id + 1];
    final_id = children[lo];
    double best_d = hamming(feat, node_desc + 32 * (size_t)final_id);
    for (uint32_t e = lo + 1; e < hi; e++) {
      const uint32_t id = children[e];
      const double d = hamming(feat, node_desc + 32 * (size_t)id);
      if (d < best_d) { best_d = d; final_id = id; }
    }
    if (current_level == nid_level) *nid = final_id;
  } while (child_off[final_id + 1] > child_off[final_id]);
  *wid = word_id[final_id];
  *w = weight[final_id];
}

// the merge half of the batch transform (:1166-1200 with BowVector::addWeight / normalize(L1) and FeatureVector::addFeature written
// out on std::map): per-feature (word, weight, node) triples -> BowVector as ascending (word, value) pairs, FeatureVector as
// CSR over ascending node ids with the feature indices of each node in ascending order.  PINNED against the reference's own
// BowVector.cpp / FeatureVector.cpp compiled into oracle/_ref/libdbow2_ref.so (tests/test_oracle_bow_ref.py).
void orc_bow_merge(int n, const int32_t* wid, const double* w, const uint32_t* nid, uint32_t* bow_word, double* bow_value,
                   int* n_words, uint32_t* fv_node, uint32_t* fv_off, uint32_t* fv_idx, int* n_fv_nodes) {
  std::map<uint32_t, double> v;
  std::map<uint32_t, std::vector<uint32_t>> fv;
  for (int i = 0; i < n; i++) {
    if (w[i] > 0) {                                   // not stopped
      auto it = v.lower_bound((uint32_t)wid[i]);
      if (it != v.end() && it->first == (uint32_t)wid[i]) it->second += w[i]; else v.insert(it, {(uint32_t)wid[i], w[i]});
      fv[nid[i]].push_back((uint32_t)i);
    }
  }
  double norm = 0.0;                                  // BowVector::normalize(L1)
  for (auto& kv : v) norm += std::fabs(kv.second);
  if (norm > 0.0) for (auto& kv : v) kv.second /= norm;
  int k = 0;
  for (auto& kv : v) { bow_word[k] = kv.first; bow_value[k] = kv.second; k++; }
  *n_words = k;
  int m = 0; uint32_t pos = 0;
  for (auto& kv : fv) {
    fv_node[m] = kv.first; fv_off[m] = pos;
    for (uint32_t f : kv.second) fv_idx[pos++] = f;
    m++;
  }
  fv_off[m] = pos;
  *n_fv_nodes = m;
}

// batch transform, TF_IDF weighting + L1 normalisation (:1124-1200): descent per feature, then the merge above.
void orc_bow_transform(const uint8_t* node_desc, const uint32_t* child_off, const uint32_t* children, const int32_t* word_id,
                       const double* weight, int n_nodes, int L, int levelsup, const uint8_t* desc, int n, uint32_t* bow_word,
                       double* bow_value, int* n_words, uint32_t* fv_node, uint32_t* fv_off, uint32_t* fv_idx, int* n_fv_nodes) {
  (void)n_nodes;
  std::vector<int32_t> wid(n > 0 ? n : 1); std::vector<double> w(n > 0 ? n : 1); std::vector<uint32_t> nid(n > 0 ? n : 1);
  for (int i = 0; i < n; i++)
    orc_bow_descend(node_desc, child_off, children, word_id, weight, L, levelsup, desc + 32 * (size_t)i, &wid[i], &w[i], &nid[i]);
  orc_bow_merge(n, wid.data(), w.data(), nid.data(), bow_word, bow_value, n_words, fv_node, fv_off, fv_idx, n_fv_nodes);
}

// L1 score between two BowVectors (L1Scoring::score, lib/DBoW2/DBoW2/ScoringObject.cpp): 1 - 0.5 * sum |v1 - v2| over
// common words written as  -sum(|a-b| - |a| - |b|) / 2
double orc_bow_score_l1(const uint32_t* w1, const double* v1, int n1, const uint32_t* w2, const double* v2, int n2) {
  double score = 0;
  int a = 0, b = 0;
  while (a < n1 && b < n2) {
    if (w1[a] == w2[b]) { score += std::fabs(v1[a] - v2[b]) - std::fabs(v1[a]) - std::fabs(v2[b]); a++; b++; }
    else if (w1[a] < w2[b]) a++;
    else b++;
  }
  return -score / 2.0;
}

}  // extern "C"
