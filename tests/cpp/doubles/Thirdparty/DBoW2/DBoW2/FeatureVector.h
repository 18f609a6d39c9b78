// DBoW2::FeatureVector IS a std::map (Thirdparty/DBoW2/DBoW2/FeatureVector.h:23-24); test double, see ../../../README.md  (Thirdparty/DBoW2/DBoW2/FeatureVector.h:23-24)
#pragma once
#include <map>
#include <vector>
namespace DBoW2 {
typedef unsigned int NodeId;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {};
}  // namespace DBoW2
