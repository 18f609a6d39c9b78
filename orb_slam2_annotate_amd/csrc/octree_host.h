// octree_host.h -- host implementation of ORBextractor::DistributeOctTree
// (src/ORBextractor.cc:566-808) on flat arrays, written in the "generation" form that the
// device kernel uses: every split pass builds the next node list as
//   reverse(children created in this pass) ++ survivors of the old list, in order,
// which is what the reference's std::list push_front/erase sequence produces.
#pragma once
#include <stdint.h>
#include <vector>

#include "kernels.h"

namespace orbfe {

// cand: candidates of one level in emission order (coordinates relative to minBorder).
// Writes selected keypoints (level coordinates, + minBorder) in list order; returns count.
int distribute_octree_host(const Candidate* cand, int n, int minX, int maxX, int minY, int maxY,
                           int N, LevelKp* out, int outCap);

}  // namespace orbfe
