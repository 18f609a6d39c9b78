// match_kernels.h -- argument blocks of the matching kernels (k_match.hip).
#pragma once
#include <stdint.h>

#include "kernels.h"

namespace orbfe {

struct NodePair {  // one vocabulary node present in both FeatureVectors
  int32_t off1, cnt1, off2, cnt2;
};

struct BowArgs {
  const NodePair* pairs;
  const uint8_t* desc1; const uint8_t* hasMp1 /* NULL: all set */; const float* angle1; const uint32_t* indices1;
  const uint8_t* desc2; const uint8_t* hasMp2 /* NULL for the KF-Frame form */; const float* angle2;
  const uint32_t* indices2;
  int angleStride;  // floats between consecutive angles (1 for plain arrays, 7 inside cv::KeyPoint records)
  float nnratio;
  int strictLow;   // 0: SearchByBoW(KF,F) accepts best <= TH_LOW; 1: (KF,KF) best < TH_LOW
  int32_t* match;  // KF-Frame: indexed by frame feature; KF-KF: indexed by KF1 feature
  int8_t* bin;     // rotation-histogram bin of the accepted match, same indexing
};

struct TriQuery { uint32_t idx1; int32_t off2, cnt2; };

struct TriArgs {
  const TriQuery* queries; int nQueries;
  const uint8_t* desc1; const float* x1; const float* y1; const float* angle1; const uint8_t* stereo1;
  const uint8_t* desc2; const uint8_t* hasMp2; const float* x2; const float* y2; const float* angle2;
  const int32_t* octave2; const uint8_t* stereo2; const uint32_t* indices2;
  const float* F12; float ex, ey;
  const float* scaleFactors2; const float* levelSigma2_2;
  int onlyStereo;
  int32_t* match; int8_t* bin;
};

struct StereoArgs {
  const float* kpL; const uint8_t* descL; int N;   // 7 floats per keypoint (cv::KeyPoint layout)
  const float* kpR; const uint8_t* descR; int Nr;
  PyramidViews pyrL, pyrR; int frameL, frameR;
  const float* scaleTab;  // device: mvScaleFactor[0..15] then mvInvScaleFactor[0..15]
  float mbf, maxD;
  float* uRight; float* depth; int32_t* sad;
  // optional row index of the right keypoints (NULL -> scan all): rows+1 starts and Nr indices per pair
  const int32_t* rowStart; const int32_t* sortedIdx; int rows; int bandR;
  // sortedRec[p] = (uR, yR, octave, iR) of the right keypoint at sorted position p (written by k_stereo_bucket next
  // to sortedIdx): the candidate scan reads ONE 16-byte record instead of the index and then three fields of the
  // 28-byte keypoint, i.e. one dependent memory round trip less per left keypoint
  const float4* sortedRec;
};

struct StereoBatch {  // frames 2p / 2p+1 of an extractor batch are the left / right image of pair p
  const float* kp; const uint8_t* desc; const int32_t* n; int capacity;
  float* uRight; float* depth; int32_t* sad;   // [nPairs * capacity]
};

// DBoW2 vocabulary tree on the device (vocabulary.hip).  Nodes are renumbered breadth-first so that the children of a
// node are ADJACENT records, in file order (the order TemplatedVocabulary::transform scans them in, :1236-1250): one
// level of the descent is one contiguous read of nChild x 48 bytes, and the record a lane loads to take its child's
// distance already carries what the next level needs if that child wins.
struct VocabNode {       // 48 bytes
  uint32_t d[8];         // FORB descriptor
  uint32_t firstChild;   // BFS position of the first child
  uint32_t nChild;       // bits 0-7: number of children (0 = leaf); bit 8: weight > 0
  uint32_t id;           // NodeId of the reference = line order of the text file (root = 0)
  int32_t word;          // WordId (leaves, in file order) or -1
};
struct VocabDevice {
  const VocabNode* nodes;  // [nNodes], BFS order, root at 0
  const double* weight;    // [nNodes], by BFS position
  uint32_t rootChildren;   // children of the root (positions 1 .. rootChildren)
};

// Batched FeatureVector construction: frame f uses desc[f*capacity ..], n[f] descriptors
struct FeatVecBatch {
  const uint8_t* desc; const int32_t* n; int capacity;
  int frameStep;             // frame f reads slot f*frameStep of desc / n (2: the left frames of an L,R-interleaved batch); 0 = 1
  int sortN;                 // power of two >= capacity
  uint32_t* word; double* weight;          // optional per-feature outputs [nFrames*capacity]
  unsigned long long* keys;  // [nFrames*capacity] scratch: (node << 32 | feature) of the used features, ~0 otherwise
  uint32_t* fvNodes;         // [nFrames*capacity] node ids ascending
  int32_t* fvOffsets;        // [nFrames*(capacity+1)]
  uint32_t* fvIndices;       // [nFrames*capacity]
  int32_t* fvCount;          // [nFrames] nodes per frame
};
void launch_vocab_featvec(hipStream_t s, const VocabDevice& v, const FeatVecBatch& b, int nFrames, int nidLevel);

struct BowBatch {  // consecutive-frame SearchByBoW over a device-resident batch
  const float* kp; const uint8_t* desc; int capacity;
  int frameStep;             // frame p reads slot p*frameStep of kp / desc (as FeatVecBatch); 0 = 1
  const uint32_t* fvNodes; const int32_t* fvOffsets; const uint32_t* fvIndices; const int32_t* fvCount;
  float nnratio;
  int32_t* match; int8_t* bin;  // [nPairs*capacity], match pre-set to -1
};

// Frame grid + window search (k_window.hip)
constexpr int GRID_MAX_FEATURES = 16384;  // keys carry a 16-bit index and are sorted in 64 KB of LDS
struct GridFrame {
  const float* x; const float* y; const int32_t* octave; const float* uRight /* NULL: monocular */;
  const uint8_t* desc; int n;
  float minX, minY, wInv, hInv;  // mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv
};
struct WindowQueries {
  const float* x; const float* y; const float* r; const int32_t* minLevel; const int32_t* maxLevel;
  const uint8_t* active /* NULL: all */; const float* ur /* NULL: no stereo check */;
  const uint8_t* desc /* NULL: indices only */; int n; int K;
  // BEST mode (best != NULL; Fuse x2 / SearchBySim3, src/ORBmatcher.cc:1029-1075, 1217-1260): no list leaves the device --
  // per query the keypoint with the smallest distance in scan order (first minimum), optionally behind Fuse's chi-square
  // gate (gateUr = the projected ur of every query, invSigma2 = mvInvLevelSigma2), kept when its distance <= maxDist
  int32_t* best; const float* gateUr; const float* invSigma2; int gate; int maxDist;
};
// The claim loops of the projection searches on the device (k_window_claim): the reference walks its map points / key points
// IN ORDER and a point that takes a feature hides it from the later ones (src/ORBmatcher.cc:77-78, 1572-1574, 1726-1727,
// 431-432, 492-493).  One workgroup per job iterates "every query chooses among the features no EARLIER query holds" to its
// fixed point -- which is the sequential result (query 0's choice is final after round 1, query 1's after round 2, ...; in
// practice 2-3 rounds) -- so the candidate lists never leave the device: the match array comes back instead.
enum { CLAIM_BEST = 0, CLAIM_RATIO = 1, CLAIM_INIT = 2 };
struct ClaimJob {
  const int32_t* count; const uint32_t* cand;  // k_window_search's lists: (dist << 16 | feature) in scan order
  int K, nq, n;                 // list capacity, queries, features of the frame
  const uint8_t* active;        // [nq] NULL: every query takes part
  const uint8_t* blocked;       // [n] features that are taken at entry, NULL: none
  const uint8_t* blockVal;      // [nq] whether query i's match hides its feature (MapPoint::Observations() > 0), NULL: always
  const int32_t* octave;        // [n] (CLAIM_RATIO: the best / second-best level test, src/ORBmatcher.cc:100-127)
  const float* qAngle; const float* fAngle;  // rotation histogram (checkOri)
  const float* fx; const float* fy; const float* qx; const float* qy;  // CLAIM_INIT: "update prev matched" (:595-600):
  float* prevX; float* prevY;   // out [nq]: the matched feature's position, else the query's own
  int mode, maxDist, checkOri;
  float nnratio;
  int32_t* choice;  // [nq] scratch: the query's current choice (feature, or dist << 16 | feature in CLAIM_INIT), -1 none
  int32_t* link;    // [nq] scratch (CLAIM_INIT: the other queries that chose the same feature)
  int32_t* owner;   // [2 n] ([n] for CLAIM_INIT) scratch in HBM, NULL: the kernel's dynamic LDS holds it
  int32_t* match;   // out.  BEST / RATIO: [n], query that holds feature i or -1.  INIT: [nq], feature of query i or -1
  int32_t* header;  // out [4]: largest list length (> K: the lists were truncated, search again), matches, rounds, 0
};
void launch_window_claim(hipStream_t s, const ClaimJob* d_jobs, const ClaimJob* h_first /* host copy of job 0 */, int nJobs, size_t ldsBytes,
                         bool initForm /* all jobs CLAIM_INIT, or none */);
void launch_grid_build(hipStream_t s, const GridFrame& f, uint32_t* sortedKey, int32_t* cellOff);
void launch_frame_from_records(hipStream_t s, const float* d_kp, const uint8_t* d_desc, int n, float* x, float* y, float* angle,
                               int32_t* octave, uint8_t* descOut, uint8_t* stereoZero);
struct WindowSearchJob {  // one job of k_window_search_multi; blockStart = first workgroup of the job (4 queries per workgroup)
  GridFrame f; const uint32_t* sortedKey; const int32_t* cellOff; WindowQueries q; int32_t* count; uint32_t* cand; int blockStart;
};
void launch_window_search_multi(hipStream_t s, const WindowSearchJob* d_jobs, int nJobs, int totalBlocks);
void launch_window_search(hipStream_t s, const GridFrame& f, const uint32_t* sortedKey, const int32_t* cellOff,
                          const WindowQueries& q, int32_t* count, uint32_t* cand);

void launch_search_by_bow(hipStream_t s, const BowArgs& a, int nPairs, int maxCnt2);
void launch_search_by_bow_multi(hipStream_t s, const BowArgs* d_args, const int32_t* d_pairStart, int K, int nPairsTotal, int maxCnt2);
void launch_search_by_bow_batch(hipStream_t s, const BowBatch& b, int nPairs, int checkOri, int32_t* d_nMatches,
                                int maxNodes /* bound on FeatureVector entries per frame, 0: capacity */);
void launch_search_triangulation(hipStream_t s, const TriArgs& a);
void launch_search_triangulation_multi(hipStream_t s, const TriArgs* d_args, const int32_t* d_blockStart, int K, int totalBlocks);
void launch_rot_prune(hipStream_t s, int32_t* match, const int8_t* bin, int n, int checkOri, int32_t* nMatches);
void launch_rot_prune_batch(hipStream_t s, int32_t* match, const int8_t* bin, int n, int nArrays, int checkOri, int32_t* nMatches);
void launch_stereo(hipStream_t s, const StereoArgs& a, int32_t* d_nStereo);
void launch_stereo_batch(hipStream_t s, const StereoArgs& a, const StereoBatch& b, int nPairs, int32_t* d_nStereo);

}  // namespace orbfe
