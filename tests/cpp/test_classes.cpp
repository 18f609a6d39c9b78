// Builds against include/orbfe_classes.hpp + liborbfe.so and exercises the C++ host layer the way
// Frame does: two extractor instances on two threads (src/Frame.cc:78-81), getters, pyramid access,
// stereo matching, DescriptorDistance.  Prints a checksum the Python test compares with the oracle.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
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
  {  // the same stereo Frame in ONE call on one handle: identical keypoints, descriptors, mvuRight, mvDepth
    ORBextractor eS(600, 1.2f, 8, 20, 7);
    std::vector<KeyPoint> sL, sR;
    std::vector<uint8_t> sdL, sdR;
    std::vector<float> su, sd;
    eS.extractStereoFrame(L.data(), R.data(), W, H, W, 80.0f, 80.0f / 200.0f, sL, sdL, sR, sdR, su, sd);
    const bool same = sL.size() == kL.size() && sR.size() == kR.size() && !std::memcmp(sL.data(), kL.data(), kL.size() * 28) &&
                      !std::memcmp(sR.data(), kR.data(), kR.size() * 28) && sdL == dL && sdR == dR && su.size() == u.size() &&
                      !std::memcmp(su.data(), u.data(), u.size() * 4) && !std::memcmp(sd.data(), d.data(), d.size() * 4);
    if (!same) { std::printf("extractStereoFrame differs from two extractions + ComputeStereoMatches\n"); return 1; }
  }
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
  // ---- resident frames, the multi-neighbour calls and RE-ENTRANCY (Tracking, LocalMapping and LoopClosing
  //      each construct stack-local matchers and call them concurrently, src/LocalMapping.cc:261, src/LoopClosing.cc:294).
  //      Reference results are taken single-threaded first (the Python side checks those against the oracle); every
  //      threaded iteration must reproduce them bit for bit.
  {
    const std::vector<float> sf = eL.GetScaleFactors(), sig2 = eL.GetScaleSigmaSquares(), isig2 = eL.GetInverseScaleSigmaSquares();
    std::vector<float> urL(kL.size()), urR(kR.size());
    for (size_t i = 0; i < kL.size(); i++) urL[i] = (i % 5 == 0) ? kL[i].x - 3.0f : -1.0f;
    for (size_t i = 0; i < kR.size(); i++) urR[i] = (i % 7 == 0) ? kR[i].x - 2.0f : -1.0f;
    std::vector<uint8_t> has1(kL.size()), has2(kR.size()), all1(kL.size(), 1);
    for (size_t i = 0; i < kL.size(); i++) has1[i] = i % 3 == 0;
    for (size_t i = 0; i < kR.size(); i++) has2[i] = i % 4 == 0;
    std::map<unsigned, std::vector<unsigned> > fm1, fm2;
    for (size_t i = 0; i < kL.size(); i++) fm1[(unsigned)((dL[32 * i] ^ dL[32 * i + 7]) % 37) * 5u + 2u].push_back((unsigned)i);
    for (size_t i = 0; i < kR.size(); i++) fm2[(unsigned)((dR[32 * i] ^ dR[32 * i + 7]) % 37) * 5u + 2u].push_back((unsigned)i);
    FeatureVectorCSR fv1(fm1), fv2(fm2);
    FrameArrays K1(kL, dL, 0.0f, (float)W, 0.0f, (float)H, urL), K2(kR, dR, 0.0f, (float)W, 0.0f, (float)H, urR);
    FrameArrays H1(kL, dL, 0.0f, (float)W, 0.0f, (float)H, urL), H2(kR, dR, 0.0f, (float)W, 0.0f, (float)H, urR);  // host-array twins
    K1.makeResident(&fv1);
    K2.makeResident(&fv2);
    const float F12[9] = {0, 0, 0, 0, 0, -1, 0, 1, 0};
    std::vector<uint8_t> valid(kL.size(), 1);
    std::vector<float> u(kL.size()), v(kL.size()), urp(kL.size(), -1.0f);
    std::vector<int32_t> lev(kL.size());
    for (size_t i = 0; i < kL.size(); i++) { u[i] = kL[i].x - 2.0f; v[i] = kL[i].y; lev[i] = kL[i].octave; valid[i] = i % 11 != 0; }
    // single-threaded references on HOST arrays
    ORBmatcher bow(0.7f, true), tri(0.6f, false), prj(0.9f, true);
    std::vector<int32_t> refBow, refBowKF, refReloc, refFuse;
    std::vector<std::pair<size_t, size_t> > refTri;
    const int nBow = bow.SearchByBoW(dL.data(), all1.data(), H1.angle.data(), H1.N(), fv1, dR.data(), H2.angle.data(), H2.N(), fv2, refBow);
    const int nBowKF = bow.SearchByBoW(dL.data(), has1.data(), H1.angle.data(), H1.N(), fv1, dR.data(), has2.data(), H2.angle.data(), H2.N(), fv2, refBowKF);
    tri.SearchForTriangulation(H1, has1, fv1, H2, has2, fv2, F12, 1000.0f, 100.0f, sf, sig2, refTri, false);
    std::vector<uint8_t> none;
    const int nRel = prj.SearchByProjection(H2, sf, none, valid, u, v, lev, H1.angle, dL, 12.0f, 100, refReloc);
    prj.Fuse(H2, sf, isig2, valid, u, v, urp, lev, dL, refFuse, 12.0f);
    std::printf("nbow=%d bow=%016llx nbowkf=%d bowkf=%016llx\n", nBow, (unsigned long long)fnv(refBow.data(), refBow.size() * 4), nBowKF,
                (unsigned long long)fnv(refBowKF.data(), refBowKF.size() * 4));
    std::atomic<int> bad(0);
    auto expect = [&](bool ok, const char* what) { if (!ok) { bad++; std::fprintf(stderr, "MISMATCH %s\n", what); } };
    // resident frames reproduce the host-array results; the multi calls reproduce K single calls
    {
      std::vector<int32_t> m;
      expect(orbfe_search_by_bow_resident(K1.resident(), all1.data(), K2.resident(), 0.7f, 1, (m.assign(K2.N(), -1), m.data())) == nBow && m == refBow, "bow resident");
      expect(orbfe_search_by_bow_kf_resident(K1.resident(), has1.data(), K2.resident(), has2.data(), 0.7f, 1, (m.assign(K1.N(), -1), m.data())) == nBowKF && m == refBowKF, "bow kf resident");
      std::vector<std::vector<std::pair<size_t, size_t> > > multi;
      std::vector<const FrameArrays*> nb(3, &K2);
      std::vector<const std::vector<uint8_t>*> mk(3, &has2);
      std::vector<float> Fs, exs(3, 1000.0f), eys(3, 100.0f);
      for (int k = 0; k < 3; k++) Fs.insert(Fs.end(), F12, F12 + 9);
      tri.SearchForTriangulationMulti(K1, has1, nb, mk, Fs, exs, eys, sf, sig2, multi, false);
      for (int k = 0; k < 3; k++) expect(multi[k] == refTri, "triangulation multi");
      // round 4: one frame / key frame against K candidates in ONE call (src/Tracking.cc:1478-1498, src/LoopClosing.cc:294-321)
      {
        std::vector<std::vector<int32_t> > mm;
        std::vector<const FrameArrays*> cands(3, &K1);
        std::vector<const std::vector<uint8_t>*> cm(3, &all1);
        std::vector<int> c = bow.SearchByBoWMulti(cands, cm, K2, mm);
        for (int k = 0; k < 3; k++) expect(c[k] == nBow && mm[k] == refBow, "bow multi (KF_k, F)");
        std::vector<const FrameArrays*> cands2(3, &K2);
        std::vector<const std::vector<uint8_t>*> cm2(3, &has2);
        c = bow.SearchByBoWMulti(K1, has1, cands2, cm2, mm);
        for (int k = 0; k < 3; k++) expect(c[k] == nBowKF && mm[k] == refBowKF, "bow multi (KF, KF_k)");
        ORBmatcher::ProjectedKeyFrame pk;
        pk.valid = valid; pk.mpDescriptors = dL; pk.u = u; pk.v = v; pk.kfAngle = H1.angle; pk.level = lev; pk.th = 12.0f; pk.ORBdist = 100;
        std::vector<ORBmatcher::ProjectedKeyFrame> pks(3, pk);
        c = prj.SearchByProjectionMulti(K2, sf, pks, mm);
        for (int k = 0; k < 3; k++) expect(c[k] == nRel && mm[k] == refReloc, "reloc multi resident");
        c = prj.SearchByProjectionMulti(H2, sf, pks, mm);
        for (int k = 0; k < 3; k++) expect(c[k] == nRel && mm[k] == refReloc, "reloc multi host arrays");
        // Frame::Frame without the features travelling twice: eL's output block still holds kL / dL
        FrameArrays X1(kL, dL, 0.0f, (float)W, 0.0f, (float)H, urL);
        X1.makeResidentFromExtractor(eL.handle());
        X1.setFeatVec(fv1);  // Frame::ComputeBoW afterwards
        std::vector<int32_t> m;
        expect(orbfe_search_by_bow_resident(X1.resident(), all1.data(), K2.resident(), 0.7f, 1, (m.assign(K2.N(), -1), m.data())) == nBow && m == refBow, "bow on a frame built from the extractor's records");
        std::vector<int32_t> fx;
        prj.Fuse(X1, sf, isig2, valid, u, v, urp, lev, dL, fx, 12.0f);
        std::vector<int32_t> fh;
        prj.Fuse(H1, sf, isig2, valid, u, v, urp, lev, dL, fh, 12.0f);
        expect(fx == fh, "fuse on a frame built from the extractor's records");
      }
      std::vector<int32_t> mr, fu, fum;
      expect(prj.SearchByProjection(K2, sf, none, valid, u, v, lev, H1.angle, dL, 12.0f, 100, mr) == nRel && mr == refReloc, "reloc resident");
      prj.Fuse(K2, sf, isig2, valid, u, v, urp, lev, dL, fu, 12.0f);
      expect(fu == refFuse, "fuse resident");
      std::vector<const FrameArrays*> kfs = {&K2, &H2, &K2};
      std::vector<uint8_t> v3; std::vector<float> u3, vv3, ur3; std::vector<int32_t> l3;
      for (int k = 0; k < 3; k++) { v3.insert(v3.end(), valid.begin(), valid.end()); u3.insert(u3.end(), u.begin(), u.end()); vv3.insert(vv3.end(), v.begin(), v.end());
                                    ur3.insert(ur3.end(), urp.begin(), urp.end()); l3.insert(l3.end(), lev.begin(), lev.end()); }
      prj.FuseSearchMulti(kfs, sf, isig2, (int)kL.size(), v3, u3, vv3, ur3, l3, dL, 12.0f, fum);
      for (int k = 0; k < 3; k++) expect(std::vector<int32_t>(fum.begin() + k * kL.size(), fum.begin() + (k + 1) * kL.size()) == refFuse, "fuse multi");
    }
    // five threads at once: three matcher threads (BoW / triangulation / projection + Fuse, resident and host-array
    // operands alternating) and two extractor threads; the BoW thread ends half way and a NEW thread takes over
    // (thread-local arena teardown and re-creation).
    const uint64_t kpRef = fnv(kL.data(), kL.size() * 28), deRef = fnv(dL.data(), dL.size());
    auto bowLoop = [&](int iters) {
      ORBmatcher mm(0.7f, true);  // stack-local, as the reference's callers construct it
      for (int it = 0; it < iters; it++) {
        std::vector<int32_t> m;
        if (it & 1) expect(orbfe_search_by_bow_resident(K1.resident(), all1.data(), K2.resident(), 0.7f, 1, (m.assign(K2.N(), -1), m.data())) == nBow && m == refBow, "threaded bow resident");
        else expect(mm.SearchByBoW(dL.data(), all1.data(), H1.angle.data(), H1.N(), fv1, dR.data(), H2.angle.data(), H2.N(), fv2, m) == nBow && m == refBow, "threaded bow");
        expect(mm.SearchByBoW(dL.data(), has1.data(), H1.angle.data(), H1.N(), fv1, dR.data(), has2.data(), H2.angle.data(), H2.N(), fv2, m) == nBowKF && m == refBowKF, "threaded bow kf");
      }
    };
    std::thread tA([&] { bowLoop(20); });
    std::thread tB([&] {
      ORBmatcher mm(0.6f, false);
      for (int it = 0; it < 40; it++) {
        std::vector<std::pair<size_t, size_t> > pr;
        mm.SearchForTriangulation(H1, has1, fv1, H2, has2, fv2, F12, 1000.0f, 100.0f, sf, sig2, pr, false);
        expect(pr == refTri, "threaded triangulation");
        std::vector<std::vector<std::pair<size_t, size_t> > > multi;
        mm.SearchForTriangulationMulti(K1, has1, std::vector<const FrameArrays*>(2, &K2), std::vector<const std::vector<uint8_t>*>(2, &has2),
                                       std::vector<float>{0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0, 0, -1, 0, 1, 0}, std::vector<float>(2, 1000.0f),
                                       std::vector<float>(2, 100.0f), sf, sig2, multi, false);
        expect(multi.size() == 2 && multi[0] == refTri && multi[1] == refTri, "threaded triangulation multi");
      }
    });
    std::thread tC([&] {
      ORBmatcher mm(0.9f, true);
      for (int it = 0; it < 40; it++) {
        std::vector<int32_t> mr, fu;
        expect(mm.SearchByProjection((it & 1) ? K2 : H2, sf, none, valid, u, v, lev, H1.angle, dL, 12.0f, 100, mr) == nRel && mr == refReloc, "threaded reloc");
        mm.Fuse((it & 1) ? H2 : K2, sf, isig2, valid, u, v, urp, lev, dL, fu, 12.0f);
        expect(fu == refFuse, "threaded fuse");
      }
    });
    auto extLoop = [&](ORBextractor& e, const std::vector<uint8_t>& img, uint64_t kr, uint64_t dr, bool check) {
      for (int it = 0; it < 25; it++) {
        std::vector<KeyPoint> k;
        std::vector<uint8_t> d;
        e(img.data(), W, H, W, k, d);
        if (check) expect(fnv(k.data(), k.size() * 28) == kr && fnv(d.data(), d.size()) == dr, "threaded extraction");
      }
    };
    std::thread tD([&] { extLoop(eL, L, kpRef, deRef, true); });
    std::thread tE([&] { extLoop(eR, R, fnv(kR.data(), kR.size() * 28), fnv(dR.data(), dR.size()), true); });
    tA.join();                            // the first BoW thread is gone (its thread-local arena with it) ...
    std::thread tA2([&] { bowLoop(20); });  // ... a new one starts while the others still run
    tB.join(); tC.join(); tD.join(); tE.join(); tA2.join();
    std::printf("reentrancy_mismatches=%d\n", bad.load());
    if (bad.load()) return 4;
  }
  // empty image: silent return, outputs untouched
  std::vector<KeyPoint> k0(3);
  std::vector<uint8_t> d0(96);
  eL(nullptr, 0, 0, 0, k0, d0);
  if (k0.size() != 3 || d0.size() != 96) return 3;
  return 0;
}
