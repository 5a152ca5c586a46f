// ============================================================================
// orb_vocab.hip -- Frame::ComputeBoW on the device: DBoW2's TemplatedVocabulary::transform(features,
// BowVector&, FeatureVector&, levelsup) (reference lib/DBoW2/DBoW2/TemplatedVocabulary.h:1124-1260; called with
// levelsup = 4 from src/Frame.cc:322-327 and src/KeyFrame.cc; ORBVocabulary = TF_IDF weighting, L1 norm).
//
// The vocabulary tree is held in HBM in flattened form (what loadFromTextFile builds in m_nodes: 256-bit node
// descriptors, children in order, word id + idf weight of the leaves; k = 10, L = 6 is 1.1 M nodes = 36 MB).
//   k_bow_descend   16 lanes per feature: the lanes take the children of the current node (two 16-byte loads each),
//                   xor + v_bcnt, then a 4-step shuffle arg-min on (distance << 8 | child rank) - first minimum wins,
//                   as the reference's strict '<' scan; four features descend per wave in lock step.
// The per-feature (word, weight, node) triples are merged on the host in the reference's order (std::map by word id,
// weights added in feature order, L1 normalisation summed in ascending word order) - a few microseconds of work next
// to the 60 Hamming distances per feature.  No CPU fallback.
// ============================================================================
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "common.h"

struct orbv_ctx {
  int device = 0, n_nodes = 0, L = 0;
  orbhip::DevBuf desc, child_off, children, word_id, weight;
};

namespace orbhip {

__global__ __launch_bounds__(256) void k_bow_descend(const uint4* __restrict__ node_desc, const uint32_t* __restrict__ child_off,
                                                     const uint32_t* __restrict__ children, const int* __restrict__ word_id,
                                                     const double* __restrict__ weight, int nid_level, const uint4* __restrict__ feat,
                                                     int n, int* __restrict__ out_word, double* __restrict__ out_weight,
                                                     uint32_t* __restrict__ out_node) {
  const int sub = threadIdx.x & 15;
  const int f = (blockIdx.x * 256 + threadIdx.x) >> 4;
  const bool live = f < n;
  uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
  if (live) { a0 = feat[2 * (size_t)f]; a1 = feat[2 * (size_t)f + 1]; }
  uint32_t cur = 0, nid = 0;
  int level = 0;
  bool done = !live;
  // every 16-lane group walks its own feature; groups of a wave that finish early idle until the deepest one is done
  while (__any(!done)) {
    uint32_t lo = 0, hi = 0;
    if (!done) { lo = child_off[cur]; hi = child_off[cur + 1]; }
    uint32_t best = 0xFFFFFFFFu;                       // (distance << 8 | rank) in the low 24 bits' order, child id carried along
    uint32_t best_id = 0;
    for (uint32_t base = lo; base < hi; base += 16) {  // k <= 16: one pass
      const uint32_t e = base + sub;
      uint32_t key = 0xFFFFFFFFu, id = 0;
      if (e < hi) {
        id = children[e];
        const uint4 b0 = node_desc[2 * (size_t)id], b1 = node_desc[2 * (size_t)id + 1];
        const int d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                      __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
        key = ((uint32_t)d << 20) | (e - lo);          // strict '<' scan in child order == minimum of (d, rank)
      }
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) {
        const uint32_t ok = __shfl_xor(key, o, 16), oid = __shfl_xor(id, o, 16);
        if (ok < key) { key = ok; id = oid; }
      }
      if (key < best) { best = key; best_id = id; }
    }
    if (!done) {
      cur = best_id;
      ++level;
      if (level == nid_level) nid = cur;
      if (child_off[cur + 1] == child_off[cur]) done = true;     // leaf
    }
  }
  if (live && sub == 0) { out_word[f] = word_id[cur]; out_weight[f] = weight[cur]; out_node[f] = nid; }
}

}  // namespace orbhip

using namespace orbhip;
#define VCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(_e)); return ORBHIP_ENODEV; } } while (0)

namespace orbhip {
// The host half of TemplatedVocabulary::transform (:1166-1200): the per-feature (word, weight, node) triples of the descent merged
// into the BowVector (std::map order, weights of equal words added in feature order, L1 norm) and the FeatureVector (CSR).
// Shared by orbv_transform and the device-resident TrackReferenceKeyFrame step (orb_track.hip).
void orbv_merge_host(const int32_t* word, const double* wt, const uint32_t* node, int n, uint32_t* bow_word, double* bow_value, int* n_words,
                     uint32_t* fv_node, uint32_t* fv_off, uint32_t* fv_idx, int* n_fv_nodes) {
  // ---- BowVector: std::map order (ascending word id); equal ids add their weights in feature order (:1158, BowVector.cpp:34-46)
  std::vector<int> live;
  for (int i = 0; i < n; i++) if (wt[i] > 0) live.push_back(i);            // "not stopped" (:1156)
  // (sorted as packed (id << 32 | feature index) keys: the order of a stable sort by id, without its indirect comparisons)
  std::vector<uint64_t> keys(live.size());
  for (size_t k = 0; k < live.size(); k++) keys[k] = ((uint64_t)(uint32_t)word[live[k]] << 32) | (uint32_t)live[k];
  std::sort(keys.begin(), keys.end());
  int nw = 0;
  for (size_t k = 0; k < keys.size(); k++) {
    const int i = (int)(uint32_t)keys[k];
    if (nw && bow_word[nw - 1] == (uint32_t)word[i]) bow_value[nw - 1] += wt[i];
    else { bow_word[nw] = (uint32_t)word[i]; bow_value[nw] = wt[i]; nw++; }
  }
  double norm = 0.0;                                                          // L1 (BowVector.cpp:62-84)
  for (int k = 0; k < nw; k++) norm += std::fabs(bow_value[k]);
  if (norm > 0.0) for (int k = 0; k < nw; k++) bow_value[k] /= norm;
  *n_words = nw;
  // ---- FeatureVector: ascending node id, feature indices ascending inside a node
  for (size_t k = 0; k < live.size(); k++) keys[k] = ((uint64_t)node[live[k]] << 32) | (uint32_t)live[k];
  std::sort(keys.begin(), keys.end());
  int m = 0; uint32_t pos = 0;
  for (size_t k = 0; k < keys.size(); k++) {
    const int i = (int)(uint32_t)keys[k];
    if (!m || fv_node[m - 1] != node[i]) { fv_node[m] = node[i]; fv_off[m] = pos; m++; }
    fv_idx[pos++] = (uint32_t)i;
  }
  fv_off[m] = pos;
  *n_fv_nodes = m;
}
}  // namespace orbhip

extern "C" {

int orbv_destroy(orbv_ctx* c);

// ---- ORBvoc.txt: TemplatedVocabulary::loadFromTextFile (lib/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1423) ------------------
// "k L scoring weighting" then one line per node (ids 1, 2, ... in file order; node 0 = root): "parent is_leaf d0 .. d31 weight".
// Children keep the file order (m_nodes[pid].children.push_back), words are numbered in file order.  Host only (no device).
// The reference's `while (!f.eof())` loop turns the empty line after the last '\n' into one more node (parent 0, descriptor
// uninitialised memory): that accident is not reproduced - blank lines are skipped.
int orbv_parse_text(const char* path, int32_t* k_out, int32_t* L_out, int32_t* scoring, int32_t* weighting, int32_t* n_nodes, int32_t* n_children,
                    uint8_t** node_desc, uint32_t** child_off, uint32_t** children, int32_t** word_id, double** weight) {
  ORBHIP_REQUIRE(path && n_nodes && n_children && node_desc && child_off && children && word_id && weight, ORBHIP_EINVAL, "NULL argument");
  FILE* f = std::fopen(path, "rb");
  if (!f) { set_error("cannot open vocabulary file %s", path); return ORBHIP_EINVAL; }
  long sz = -1;
  if (std::fseek(f, 0, SEEK_END) == 0) sz = std::ftell(f);
  if (sz < 0 || std::fseek(f, 0, SEEK_SET) != 0) { std::fclose(f); set_error("cannot size vocabulary file %s", path); return ORBHIP_EINVAL; }
  std::vector<char> buf((size_t)sz + 1);
  const size_t got = std::fread(buf.data(), 1, (size_t)sz, f);
  const bool read_err = std::ferror(f) != 0;
  std::fclose(f);
  if (read_err || got != (size_t)sz) { set_error("short read of vocabulary file %s (%zu of %ld bytes)", path, got, sz); return ORBHIP_EINVAL; }
  buf[got] = 0;
  char* p = buf.data(); char* end = p + got;
  auto next_line = [&](char*& a, char*& b) -> bool {              // [a, b) = next non-blank line
    while (p < end) {
      a = p; while (p < end && *p != '\n') p++;
      b = p; if (p < end) p++;
      for (char* q = a; q < b; q++) if (*q != ' ' && *q != '\r' && *q != '\t') return true;
    }
    return false;
  };
  char *a, *b;
  if (!next_line(a, b)) { set_error("empty vocabulary file"); return ORBHIP_EINVAL; }
  char* q = a;
  // a token = strtol / strtod must ADVANCE (a missing token would otherwise parse as 0) and must stay inside its line
  bool tok_ok = true;
  // and must end at white space ("0.5" is not the integer 0 followed by the number .5)
  auto ends = [](const char* e, const char* lim) { return e == lim || *e == ' ' || *e == '\t' || *e == '\r' || *e == '\n' || *e == 0; };
  auto tol = [&](char*& c, char* lim) -> long { char* e = c; const long v = std::strtol(c, &e, 10); if (e == c || e > lim || !ends(e, lim)) tok_ok = false; c = e; return v; };
  auto tod = [&](char*& c, char* lim) -> double { char* e = c; const double v = std::strtod(c, &e); if (e == c || e > lim || !ends(e, lim)) tok_ok = false; c = e; return v; };
  const long k = tol(q, b), L = tol(q, b), n1 = tol(q, b), n2 = tol(q, b);
  if (!tok_ok) { set_error("not a DBoW2 text vocabulary (header needs 4 integers)"); return ORBHIP_EINVAL; }
  if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) { set_error("not a DBoW2 text vocabulary (header %ld %ld %ld %ld)", k, L, n1, n2); return ORBHIP_EINVAL; }
  std::vector<int32_t> parent(1, -1), wid(1, -1); std::vector<uint8_t> desc(32, 0); std::vector<double> wt(1, 0.0);
  int nwords = 0;
  while (next_line(a, b)) {
    q = a;
    const long pid = tol(q, b), leaf = tol(q, b);
    const size_t nid = parent.size();
    if (pid < 0 || (size_t)pid >= nid) { set_error("vocabulary node %zu names parent %ld", nid, pid); return ORBHIP_EINVAL; }
    parent.push_back((int32_t)pid);
    for (int i = 0; i < 32; i++) {
      const long v = tol(q, b);
      if (v < 0 || v > 255) tok_ok = false;
      desc.push_back((uint8_t)v);
    }
    wt.push_back(tod(q, b));
    if (!tok_ok) { set_error("vocabulary node %zu: expected 'parent leaf 32 bytes weight' (35 tokens)", nid); return ORBHIP_EINVAL; }
    wid.push_back(leaf > 0 ? nwords++ : -1);
  }
  const size_t n = parent.size();
  {                                                              // a k-ary tree of depth L has at most (k^(L+1) - 1) / (k - 1) nodes
    double lim = 1.0, pw = 1.0;
    for (long l = 0; l < L; l++) { pw *= (double)std::max<long>(k, 1); lim += pw; }
    if ((double)n > lim) { set_error("vocabulary has %zu nodes, more than a %ld-ary tree of depth %ld can hold", n, k, L); return ORBHIP_EINVAL; }
  }
  std::vector<uint32_t> off(n + 1, 0);
  for (size_t i = 1; i < n; i++) off[parent[i] + 1]++;
  for (size_t i = 0; i < n; i++) off[i + 1] += off[i];
  std::vector<uint32_t> ch(n > 1 ? n - 1 : 1), fill(off.begin(), off.end() - 1);
  for (size_t i = 1; i < n; i++) ch[fill[parent[i]]++] = (uint32_t)i;         // ascending node id = file order = push_back order
  auto dup = [](const void* src, size_t bytes) { void* d = std::malloc(bytes ? bytes : 1); if (d && bytes) std::memcpy(d, src, bytes); return d; };
  *node_desc = (uint8_t*)dup(desc.data(), desc.size()); *child_off = (uint32_t*)dup(off.data(), off.size() * 4); *children = (uint32_t*)dup(ch.data(), (n - 1) * 4);
  *word_id = (int32_t*)dup(wid.data(), n * 4); *weight = (double*)dup(wt.data(), n * 8);
  if (k_out) *k_out = (int32_t)k; if (L_out) *L_out = (int32_t)L; if (scoring) *scoring = (int32_t)n1; if (weighting) *weighting = (int32_t)n2;
  *n_nodes = (int32_t)n; *n_children = (int32_t)(n - 1);
  return 0;
}
void orbv_free_parsed(void* p) { std::free(p); }

int orbv_create(const uint8_t* node_desc, const uint32_t* child_off, const uint32_t* children, const int32_t* word_id, const double* weight, int n_nodes, int L,
                int device, orbv_ctx** out);
int orbv_load_text(const char* path, int device, orbv_ctx** out) {
  int32_t k = 0, L = 0, n = 0, nc = 0; uint8_t* nd = nullptr; uint32_t *co = nullptr, *ch = nullptr; int32_t* wi = nullptr; double* wt = nullptr;
  int32_t scoring = 0, weighting = 0;
  if (int rc = orbv_parse_text(path, &k, &L, &scoring, &weighting, &n, &nc, &nd, &co, &ch, &wi, &wt)) return rc;
  if (scoring != 0 || weighting != 0) {                          // DBoW2 enums: L1_NORM = 0, TF_IDF = 0 - what orbv_transform / orbv_score_l1 implement
    std::free(nd); std::free(co); std::free(ch); std::free(wi); std::free(wt);
    set_error("vocabulary %s uses scoring %d / weighting %d; only L1_NORM (0) / TF_IDF (0) - ORBvoc.txt - are implemented", path, scoring, weighting);
    return ORBHIP_EINVAL;
  }
  const int rc = orbv_create(nd, co, ch, wi, wt, n, L, device, out);
  std::free(nd); std::free(co); std::free(ch); std::free(wi); std::free(wt);
  return rc;
}

int orbv_create(const uint8_t* node_desc, const uint32_t* child_off, const uint32_t* children, const int32_t* word_id,
                const double* weight, int n_nodes, int L, int device, orbv_ctx** out) {
  ORBHIP_REQUIRE(node_desc && child_off && children && word_id && weight && out && n_nodes > 1 && L > 0, ORBHIP_EINVAL, "NULL argument");
  ORBHIP_REQUIRE(child_off[0] == 0 && child_off[1] > 0, ORBHIP_EINVAL, "the root (node 0) must have children");
  const uint32_t nchild = child_off[n_nodes];
  for (int i = 0; i < n_nodes; i++) ORBHIP_REQUIRE(child_off[i + 1] >= child_off[i] && child_off[i + 1] - child_off[i] <= 255, ORBHIP_EINVAL, "bad children table");
  for (uint32_t e = 0; e < nchild; e++) ORBHIP_REQUIRE(children[e] > 0 && children[e] < (uint32_t)n_nodes, ORBHIP_EINVAL, "child index out of range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (the HIP path has no CPU fallback)"); return ORBHIP_ENODEV; }
  ORBHIP_REQUIRE(device >= 0 && device < ndev, ORBHIP_EINVAL, "bad device index");
  VCHK(hipSetDevice(device));
  orbv_ctx* c = new orbv_ctx();
  c->device = device; c->n_nodes = n_nodes; c->L = L;
  int rc = 0;
  if ((rc = c->desc.ensure((size_t)n_nodes * 32)) || (rc = c->child_off.ensure((size_t)(n_nodes + 1) * 4)) || (rc = c->children.ensure((size_t)std::max<uint32_t>(nchild, 1) * 4)) ||
      (rc = c->word_id.ensure((size_t)n_nodes * 4)) || (rc = c->weight.ensure((size_t)n_nodes * 8))) { orbv_destroy(c); return rc; }
  hipError_t e = hipMemcpy(c->desc.p, node_desc, (size_t)n_nodes * 32, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(c->child_off.p, child_off, (size_t)(n_nodes + 1) * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess && nchild) e = hipMemcpy(c->children.p, children, (size_t)nchild * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(c->word_id.p, word_id, (size_t)n_nodes * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(c->weight.p, weight, (size_t)n_nodes * 8, hipMemcpyHostToDevice);
  if (e != hipSuccess) { set_error("orbv_create: upload failed: %s", hipGetErrorString(e)); orbv_destroy(c); return ORBHIP_ENODEV; }
  *out = c;
  return 0;
}

int orbv_destroy(orbv_ctx* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  c->desc.release(); c->child_off.release(); c->children.release(); c->word_id.release(); c->weight.release();
  delete c;
  return 0;
}

int orbv_descend_device(orbv_ctx* c, const uint8_t* d_desc, int n, int levelsup, int32_t* d_word, double* d_weight, uint32_t* d_node, void* stream) {
  ORBHIP_REQUIRE(c && n >= 0, ORBHIP_EINVAL, "NULL argument");
  if (n == 0) return 0;
  ORBHIP_REQUIRE(d_desc && d_word && d_weight && d_node, ORBHIP_EINVAL, "NULL argument");
  hipLaunchKernelGGL(k_bow_descend, dim3((n * 16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, c->desc.as<uint4>(), c->child_off.as<uint32_t>(),
                     c->children.as<uint32_t>(), c->word_id.as<int>(), c->weight.as<double>(), c->L - levelsup, (const uint4*)d_desc, n, d_word,
                     d_weight, d_node);
  VCHK(hipGetLastError());
  return 0;
}

int orbv_transform(orbv_ctx* c, const uint8_t* desc, int n, int levelsup, uint32_t* bow_word, double* bow_value, int* n_words,
                   uint32_t* fv_node, uint32_t* fv_off, uint32_t* fv_idx, int* n_fv_nodes) {
  ORBHIP_REQUIRE(c && n >= 0 && n_words && n_fv_nodes && fv_off, ORBHIP_EINVAL, "NULL argument");
  *n_words = 0; *n_fv_nodes = 0; fv_off[0] = 0;
  if (n == 0) return 0;
  ORBHIP_REQUIRE(desc && bow_word && bow_value && fv_node && fv_idx, ORBHIP_EINVAL, "NULL argument");
  VCHK(hipSetDevice(c->device));
  DevBuf dd, dw, dv, dn;
  auto cleanup = [&]() { dd.release(); dw.release(); dv.release(); dn.release(); };
  int rc = 0;
  if ((rc = dd.ensure((size_t)n * 32)) || (rc = dw.ensure((size_t)n * 4)) || (rc = dv.ensure((size_t)n * 8)) || (rc = dn.ensure((size_t)n * 4))) { cleanup(); return rc; }
  std::vector<int32_t> word(n); std::vector<double> wt(n); std::vector<uint32_t> node(n);
  hipError_t e = hipMemcpy(dd.p, desc, (size_t)n * 32, hipMemcpyHostToDevice);
  if (e == hipSuccess) { rc = orbv_descend_device(c, dd.as<uint8_t>(), n, levelsup, dw.as<int32_t>(), dv.as<double>(), dn.as<uint32_t>(), nullptr); if (rc) { cleanup(); return rc; } }
  if (e == hipSuccess) e = hipMemcpy(word.data(), dw.p, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(wt.data(), dv.p, (size_t)n * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(node.data(), dn.p, (size_t)n * 4, hipMemcpyDeviceToHost);
  cleanup();
  if (e != hipSuccess) { set_error("orbv_transform: %s", hipGetErrorString(e)); return ORBHIP_ENODEV; }
  orbhip::orbv_merge_host(word.data(), wt.data(), node.data(), n, bow_word, bow_value, n_words, fv_node, fv_off, fv_idx, n_fv_nodes);
  return 0;
}

double orbv_score_l1(const uint32_t* w1, const double* v1, int n1, const uint32_t* w2, const double* v2, int n2) {
  double score = 0;                                                           // L1Scoring::score (ScoringObject.cpp:23-68)
  int a = 0, b = 0;
  while (a < n1 && b < n2) {
    if (w1[a] == w2[b]) { score += std::fabs(v1[a] - v2[b]) - std::fabs(v1[a]) - std::fabs(v2[b]); a++; b++; }
    else if (w1[a] < w2[b]) a++;
    else b++;
  }
  return -score / 2.0;
}

}  // extern "C"
