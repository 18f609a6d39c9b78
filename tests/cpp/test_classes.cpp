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
  // the remaining ORBmatcher searches through the C++ layer (LocalMapping / LoopClosing / relocalisation callers):
  // inputs derived from the two frames by fixed formulas the Python test repeats for the oracle
  {
    const std::vector<float> sf = eL.GetScaleFactors(), sig2 = eL.GetScaleSigmaSquares(), isig2 = eL.GetInverseScaleSigmaSquares();
    std::vector<float> urL(kL.size()), urR(kR.size());
    for (size_t i = 0; i < kL.size(); i++) urL[i] = (i % 5 == 0) ? kL[i].x - 3.0f : -1.0f;   // some stereo keypoints
    for (size_t i = 0; i < kR.size(); i++) urR[i] = (i % 7 == 0) ? kR[i].x - 2.0f : -1.0f;
    FrameArrays K1(kL, dL, 0.0f, (float)W, 0.0f, (float)H, urL), K2(kR, dR, 0.0f, (float)W, 0.0f, (float)H, urR);
    FrameArrays M2(kR, dR, 0.0f, (float)W, 0.0f, (float)H);  // monocular view of the right frame
    std::vector<uint8_t> has1(kL.size()), has2(kR.size());
    for (size_t i = 0; i < kL.size(); i++) has1[i] = i % 3 == 0;
    for (size_t i = 0; i < kR.size(); i++) has2[i] = i % 4 == 0;
    std::map<unsigned, std::vector<unsigned> > fm1, fm2;  // DBoW2::FeatureVector stand-in: node = f(descriptor)
    for (size_t i = 0; i < kL.size(); i++) fm1[(unsigned)((dL[32 * i] ^ dL[32 * i + 7]) % 37) * 5u + 2u].push_back((unsigned)i);
    for (size_t i = 0; i < kR.size(); i++) fm2[(unsigned)((dR[32 * i] ^ dR[32 * i + 7]) % 37) * 5u + 2u].push_back((unsigned)i);
    FeatureVectorCSR fv1(fm1), fv2(fm2);
    const float F12[9] = {0, 0, 0, 0, 0, -1, 0, 1, 0};  // rectified pair: epipolar lines are image rows
    std::vector<std::pair<size_t, size_t> > pairs;
    ORBmatcher tri(0.6f, false);
    const int nTri = tri.SearchForTriangulation(K1, has1, fv1, K2, has2, fv2, F12, 1000.0f, 100.0f, sf, sig2, pairs, false);
    std::vector<int32_t> flat;
    for (auto& pr : pairs) { flat.push_back((int32_t)pr.first); flat.push_back((int32_t)pr.second); }
    std::vector<std::pair<size_t, size_t> > pairsS;
    const int nTriS = tri.SearchForTriangulation(K1, has1, fv1, K2, has2, fv2, F12, 1000.0f, 100.0f, sf, sig2, pairsS, true);
    // relocalisation / loop-closing / fuse / Sim3 searches: the left frame's keypoints play the projected map points
    std::vector<uint8_t> valid(kL.size(), 1), validR(kR.size(), 1), none;
    std::vector<float> u(kL.size()), v(kL.size()), u2(kR.size()), v2(kR.size()), urp(kL.size(), -1.0f);
    std::vector<int32_t> lev(kL.size()), lev2(kR.size());
    for (size_t i = 0; i < kL.size(); i++) { u[i] = kL[i].x - 2.0f; v[i] = kL[i].y; lev[i] = kL[i].octave; valid[i] = i % 11 != 0; }
    for (size_t i = 0; i < kR.size(); i++) { u2[i] = kR[i].x + 2.0f; v2[i] = kR[i].y; lev2[i] = kR[i].octave; }
    ORBmatcher m(0.9f, true);
    std::vector<int32_t> mReloc, mSim, bestA, bestB, m12;
    const int nReloc = m.SearchByProjection(M2, sf, none, valid, u, v, lev, K1.angle, dL, 12.0f, 100, mReloc);
    const int nSim = m.SearchByProjection(M2, sf, none, valid, u, v, lev, dL, 10, mSim);
    m.Fuse(M2, sf, isig2, valid, u, v, urp, lev, dL, bestA, 12.0f);
    m.Fuse(M2, sf, valid, u, v, lev, dL, 12.0f, bestB);
    const int nS3 = m.SearchBySim3(K1, K2, sf, sf, valid, u, v, lev, dL, validR, u2, v2, lev2, dR, 12.0f, m12);
    std::printf("ntri=%d tri=%016llx ntris=%d nreloc=%d reloc=%016llx nsim=%d sim=%016llx fusea=%016llx fuseb=%016llx ns3=%d s3=%016llx\n",
                nTri, (unsigned long long)fnv(flat.data(), flat.size() * 4), nTriS, nReloc,
                (unsigned long long)fnv(mReloc.data(), mReloc.size() * 4), nSim,
                (unsigned long long)fnv(mSim.data(), mSim.size() * 4), (unsigned long long)fnv(bestA.data(), bestA.size() * 4),
                (unsigned long long)fnv(bestB.data(), bestB.size() * 4), nS3, (unsigned long long)fnv(m12.data(), m12.size() * 4));
  }
  // empty image: silent return, outputs untouched
  std::vector<KeyPoint> k0(3);
  std::vector<uint8_t> d0(96);
  eL(nullptr, 0, 0, 0, k0, d0);
  if (k0.size() != 3 || d0.size() != 96) return 3;
  return 0;
}
