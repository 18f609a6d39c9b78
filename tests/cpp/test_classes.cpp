// Builds against include/orbfe_classes.hpp + liborbfe.so and exercises the C++ host layer the way
// Frame does: two extractor instances on two threads (src/Frame.cc:78-81), getters, pyramid access,
// stereo matching, DescriptorDistance.  Prints a checksum the Python test compares with the oracle.
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "orbfe_classes.hpp"

using namespace orbfe_cpp;

static std::vector<uint8_t> read_file(const char* path, size_t n) {
  std::vector<uint8_t> v(n);
  FILE* f = std::fopen(path, "rb");
  if (!f || std::fread(v.data(), 1, n, f) != n) { std::fprintf(stderr, "cannot read %s\n", path); std::exit(2); }
  std::fclose(f);
  return v;
}

static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
  const uint8_t* b = (const uint8_t*)p;
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int W = std::atoi(argv[3]), H = std::atoi(argv[4]);
  std::vector<uint8_t> L = read_file(argv[1], (size_t)W * H), R = read_file(argv[2], (size_t)W * H);
  ORBextractor eL(600, 1.2f, 8, 20, 7), eR(600, 1.2f, 8, 20, 7);
  std::vector<KeyPoint> kL, kR;
  std::vector<uint8_t> dL, dR;
  std::thread tl([&] { eL(L.data(), W, H, W, kL, dL); });
  std::thread tr([&] { eR(R.data(), W, H, W, kR, dR); });
  tl.join();
  tr.join();
  std::vector<float> u, d;
  ComputeStereoMatches(eL, eR, kL, dL, kR, dR, 80.0f, 80.0f / 200.0f, u, d);
  const std::vector<Image>& pyr = eL.mvImagePyramid();
  uint64_t hp = 1469598103934665603ull;
  for (const Image& im : pyr) hp = fnv(im.data.data(), im.data.size(), hp);
  std::printf("nL=%zu nR=%zu kp=%016llx desc=%016llx u=%016llx d=%016llx pyr=%016llx levels=%d sf1=%.9g dist=%d\n",
              kL.size(), kR.size(), (unsigned long long)fnv(kL.data(), kL.size() * 28),
              (unsigned long long)fnv(dL.data(), dL.size()), (unsigned long long)fnv(u.data(), u.size() * 4),
              (unsigned long long)fnv(d.data(), d.size() * 4), (unsigned long long)hp, eL.GetLevels(),
              (double)eL.GetScaleFactors()[1], ORBmatcher::DescriptorDistance(dL.data(), dL.data() + 32));
  // tracking-thread searches on the pair taken as two consecutive monocular frames
  {
    FrameArrays F1(kL, dL, 0.0f, (float)W, 0.0f, (float)H), F2(kR, dR, 0.0f, (float)W, 0.0f, (float)H);
    std::vector<float> px(kL.size()), py(kL.size());
    for (size_t i = 0; i < kL.size(); i++) { px[i] = kL[i].x; py[i] = kL[i].y; }
    std::vector<int32_t> m12, mc;
    ORBmatcher init(0.9f, true);
    const int nInit = init.SearchForInitialization(F1, F2, px, py, m12, 100);
    std::vector<uint8_t> valid(kL.size(), 1);
    std::vector<int32_t> oct(kL.size());
    for (size_t i = 0; i < kL.size(); i++) oct[i] = kL[i].octave;
    const std::vector<float> sf = eL.GetScaleFactors();
    const int nProj = init.SearchByProjection(F2, sf, 0.0f, valid, F1.x, F1.y, std::vector<float>(), oct, F1.angle, dL, 0,
                                              15.0f, mc);
    const std::vector<size_t> area = F2.GetFeaturesInArea(160.0f, 100.0f, 40.0f, 0, 2);
    std::vector<int32_t> area32(area.begin(), area.end());
    std::printf("ninit=%d init=%016llx nproj=%d proj=%016llx area=%016llx narea=%zu\n", nInit,
                (unsigned long long)fnv(m12.data(), m12.size() * 4), nProj,
                (unsigned long long)fnv(mc.data(), mc.size() * 4),
                (unsigned long long)fnv(area32.data(), area32.size() * 4), area.size());
  }
  // empty image: silent return, outputs untouched
  std::vector<KeyPoint> k0(3);
  std::vector<uint8_t> d0(96);
  eL(nullptr, 0, 0, 0, k0, d0);
  if (k0.size() != 3 || d0.size() != 96) return 3;
  return 0;
}
