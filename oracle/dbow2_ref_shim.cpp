// dbow2_ref_shim.cpp -- C entry points over the REFERENCE's own DBoW2 containers, compiled together with
// /root/reference/Thirdparty/DBoW2/DBoW2/{FeatureVector,BowVector}.cpp (unmodified, where they lie) into
// oracle/_ref/libdbow2_ref.so by oracle/Makefile (target `ref`).  Test infrastructure only: it pins the
// container semantics the CSR FeatureVector of the C-ABI must reproduce (ascending node ids, feature
// indices in insertion order, BowVector::addWeight accumulation order and L1 normalisation).
// These two files are the only part of the reference that builds here (no OpenCV / Eigen needed).
#include <cstdint>

#include "BowVector.h"      // -I /root/reference/Thirdparty/DBoW2/DBoW2
#include "FeatureVector.h"

extern "C" {

// FeatureVector::addFeature(node_of_feature[i], i) for i = 0..n-1 (the call pattern of
// TemplatedVocabulary::transform, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1180), then the map flattened:
// out_nodes[k] ascending, out_offsets[k..k+1] the range of node k in out_indices.  Returns the node count.
int ref_featvec_build(const uint32_t* node_of_feature, int n, uint32_t* out_nodes, int32_t* out_offsets,
                      uint32_t* out_indices) {
  DBoW2::FeatureVector fv;
  for (int i = 0; i < n; i++) fv.addFeature(node_of_feature[i], (unsigned int)i);
  int k = 0, pos = 0;
  out_offsets[0] = 0;
  for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++k) {
    out_nodes[k] = it->first;
    for (size_t j = 0; j < it->second.size(); j++) out_indices[pos++] = it->second[j];
    out_offsets[k + 1] = pos;
  }
  return k;
}

// BowVector::addWeight(word[i], weight[i]) for every i with weight[i] > 0 in feature order
// (TemplatedVocabulary.h:1176-1179), then normalize(L1) when l1_normalize != 0 (:1187-1190, the ORB vocabulary's
// L1_NORM scoring).  Returns the number of words; ids ascending.
int ref_bowvec_build(const uint32_t* word, const double* weight, int n, int l1_normalize, uint32_t* out_ids,
                     double* out_values) {
  DBoW2::BowVector bv;
  for (int i = 0; i < n; i++)
    if (weight[i] > 0) bv.addWeight(word[i], weight[i]);
  if (l1_normalize) bv.normalize(DBoW2::L1);
  int k = 0;
  for (DBoW2::BowVector::const_iterator it = bv.begin(); it != bv.end(); ++it, ++k) {
    out_ids[k] = it->first;
    out_values[k] = it->second;
  }
  return k;
}

}  // extern "C"
