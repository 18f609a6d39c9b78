// Compiles the REAL drop-in classes -- include/ORBextractor.h, include/ORBmatcher.h + src/ORBmatcher_orbfe.cc -- against
// the functional test doubles of tests/cpp/doubles/ and RUNS all 12 ORBmatcher methods and ORBextractor::operator() on
// the GPU.  The scenario (poses, map points, key frames, frames) comes from tests/test_gpu_dropin.py as named arrays; the
// un-flattened MapPoint* results go back as named arrays of map-point ids, which the Python side compares with the
// oracle fed by an independent numpy-float32 prologue.  Nothing of the reference is compiled or linked here.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher.h"

using namespace ORB_SLAM2;

float Frame::mnMinX = 0, Frame::mnMaxX = 0, Frame::mnMinY = 0, Frame::mnMaxY = 0;

// ---- named arrays: [u32 name length][name][u8 kind: 0 u8, 1 i32, 2 f32][u32 count][data] ----
struct Arr { int kind = 0; std::vector<uint8_t> raw; size_t n = 0;
  const float* f() const { return reinterpret_cast<const float*>(raw.data()); }
  const int32_t* i() const { return reinterpret_cast<const int32_t*>(raw.data()); }
  const uint8_t* b() const { return raw.data(); } };
static std::map<std::string, Arr> g_in;
static FILE* g_out = nullptr;
static void load(const char* path) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(2); }
  uint32_t nl;
  while (std::fread(&nl, 4, 1, f) == 1) {
    std::string name(nl, ' ');
    uint8_t kind; uint32_t cnt;
    if (std::fread(&name[0], 1, nl, f) != nl || std::fread(&kind, 1, 1, f) != 1 || std::fread(&cnt, 4, 1, f) != 1) std::exit(2);
    Arr a; a.kind = kind; a.n = cnt; a.raw.resize((size_t)cnt * (kind == 0 ? 1 : 4) + 4);
    if (cnt && std::fread(a.raw.data(), kind == 0 ? 1 : 4, cnt, f) != cnt) std::exit(2);
    g_in[name] = a;
  }
  std::fclose(f);
}
static const Arr& in(const std::string& k) {
  if (!g_in.count(k)) { std::fprintf(stderr, "missing input %s\n", k.c_str()); std::exit(2); }
  return g_in[k];
}
static void put(const std::string& name, int kind, const void* data, size_t cnt) {
  const uint32_t nl = (uint32_t)name.size(), c = (uint32_t)cnt;
  const uint8_t k = (uint8_t)kind;
  std::fwrite(&nl, 4, 1, g_out); std::fwrite(name.data(), 1, nl, g_out); std::fwrite(&k, 1, 1, g_out); std::fwrite(&c, 4, 1, g_out);
  if (cnt) std::fwrite(data, kind == 0 ? 1 : 4, cnt, g_out);
}
static void put_i(const std::string& name, const std::vector<int32_t>& v) { put(name, 1, v.data(), v.size()); }
static void put_i1(const std::string& name, int v) { int32_t x = v; put(name, 1, &x, 1); }

static cv::Mat matf(const float* p, int r, int c) { cv::Mat m(r, c, CV_32F); std::memcpy(m.data, p, (size_t)r * c * 4); return m; }

// ---- scenario objects ----
static std::vector<MapPoint> g_mp;  // world points; ids = indices
static int id_of(MapPoint* p) { return p ? p->id : -1; }
static std::vector<int32_t> ids(const std::vector<MapPoint*>& v) { std::vector<int32_t> o(v.size()); for (size_t i = 0; i < v.size(); i++) o[i] = id_of(v[i]); return o; }

static void reset_points() {
  const Arr &X = in("mp_pos"), &Nn = in("mp_normal"), &D = in("mp_desc"), &mx = in("mp_maxraw"), &lo = in("mp_min"), &hi = in("mp_max"),
            &ob = in("mp_nobs"), &bd = in("mp_bad");
  const size_t M = ob.n;
  g_mp.assign(M, MapPoint());
  for (size_t i = 0; i < M; i++) {
    MapPoint& p = g_mp[i];
    p.id = (int)i;
    p.pos = matf(X.f() + 3 * i, 3, 1);
    p.normal = matf(Nn.f() + 3 * i, 3, 1);
    p.desc = cv::Mat(1, 32, CV_8U);
    std::memcpy(p.desc.data, D.b() + 32 * i, 32);
    p.maxDistRaw = mx.f()[i]; p.minDist = lo.f()[i]; p.maxDist = hi.f()[i];
    p.nObs = ob.i()[i]; p.bad = bd.b()[i] != 0;
  }
}
template <class T> static void fill_common(T& o, const std::string& c) {
  const Arr& K = in("K");
  o.fx = K.f()[0]; o.fy = K.f()[1]; o.cx = K.f()[2]; o.cy = K.f()[3]; o.mbf = K.f()[4]; o.mb = K.f()[5];
  const Arr& kp = in(c + "_kp");
  o.N = (int)(kp.n / 7);
  o.mvKeys.resize(o.N);
  std::memcpy(static_cast<void*>(o.mvKeys.data()), kp.raw.data(), (size_t)o.N * 28);
  o.mvKeysUn = o.mvKeys;
  const Arr& ur = in(c + "_ur");
  o.mvuRight.assign(ur.f(), ur.f() + ur.n);
  o.mDescriptors = cv::Mat(o.N, 32, CV_8U);
  std::memcpy(o.mDescriptors.data, in(c + "_desc").b(), (size_t)o.N * 32);
  const Arr& nd = in(c + "_node");
  o.mFeatVec.clear();
  for (int i = 0; i < o.N; i++) o.mFeatVec[(unsigned)nd.i()[i]].push_back((unsigned)i);
  const Arr &sf = in("sf"), &s2 = in("sigma2"), &is2 = in("invsigma2");
  o.mvScaleFactors.assign(sf.f(), sf.f() + sf.n);
  o.mvLevelSigma2.assign(s2.f(), s2.f() + s2.n);
  o.mvInvLevelSigma2.assign(is2.f(), is2.f() + is2.n);
  o.mnScaleLevels = (int)sf.n;
  o.mfLogScaleFactor = in("logsf").f()[0];
  const Arr& mp = in(c + "_mp");
  o.mvpMapPoints.assign(o.N, nullptr);
  for (int i = 0; i < o.N; i++) if (mp.i()[i] >= 0) o.mvpMapPoints[i] = &g_mp[mp.i()[i]];
}
static void make_kf(KeyFrame& k, const std::string& c) {
  fill_common(k, c);
  k.Tcw = matf(in(c + "_Tcw").f(), 4, 4);
  const Arr& b = in("bounds");
  k.mnMinX = (int)b.f()[0]; k.mnMaxX = (int)b.f()[1]; k.mnMinY = (int)b.f()[2]; k.mnMaxY = (int)b.f()[3];
  for (int i = 0; i < k.N; i++) if (k.mvpMapPoints[i]) k.mvpMapPoints[i]->obs[&k] = (size_t)i;  // (nObs comes from the scenario)
}
static void make_frame(Frame& f, const std::string& c) {
  fill_common(f, c);
  f.mTcw = matf(in(c + "_Tcw").f(), 4, 4);
  const Arr& ol = in(c + "_outlier");
  f.mvbOutlier.assign(f.N, false);
  for (int i = 0; i < f.N; i++) f.mvbOutlier[i] = ol.b()[i] != 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  load(argv[1]);
  g_out = std::fopen(argv[2], "wb");
  if (!g_out) return 2;
  const Arr& b = in("bounds");
  Frame::mnMinX = b.f()[0]; Frame::mnMaxX = b.f()[1]; Frame::mnMinY = b.f()[2]; Frame::mnMaxY = b.f()[3];

  {  // ---- ORBextractor::operator() + mvImagePyramid ----
    const Arr& img = in("image");
    const int W = in("image_wh").i()[0], H = in("image_wh").i()[1];
    ORBextractor ex(600, 1.2f, 8, 20, 7);
    cv::Mat im(H, W, CV_8UC1, (void*)img.b()), desc;
    std::vector<cv::KeyPoint> kps;
    ex(im, cv::Mat(), kps, desc);
    put("ex_kp", 0, kps.data(), kps.size() * 28);
    put("ex_desc", 0, desc.data, (size_t)desc.rows * 32);
    std::vector<uint8_t> pyr;
    for (size_t l = 0; l < ex.mvImagePyramid.size(); l++)
      for (int y = 0; y < ex.mvImagePyramid[l].rows; y++)
        pyr.insert(pyr.end(), ex.mvImagePyramid[l].ptr<uchar>(y), ex.mvImagePyramid[l].ptr<uchar>(y) + ex.mvImagePyramid[l].cols);
    put("ex_pyr", 0, pyr.data(), pyr.size());
    cv::Mat empty, d2;
    std::vector<cv::KeyPoint> k2(3);
    ex(empty, cv::Mat(), k2, d2);  // empty image: silent return (src/ORBextractor.cc:1122-1123)
    put_i1("ex_empty_untouched", (int)k2.size());
    put_i1("ex_levels", ex.GetLevels());
  }
  {  // (12) DescriptorDistance
    reset_points();
    put_i1("dd", ORBmatcher::DescriptorDistance(g_mp[0].desc, g_mp[1].desc));
  }
  {  // (1) SearchByProjection(Frame, MapPoints)
    reset_points();
    Frame F; make_frame(F, "C");
    const Arr &sel = in("m1_points"), &iv = in("m1_inview"), &lv = in("m1_level"), &vc = in("m1_viewcos"), &px = in("m1_px"), &py = in("m1_py"), &pxr = in("m1_pxr");
    std::vector<MapPoint*> pts;
    for (size_t i = 0; i < sel.n; i++) {
      MapPoint* p = &g_mp[sel.i()[i]];
      p->mbTrackInView = iv.b()[i] != 0; p->mnTrackScaleLevel = lv.i()[i]; p->mTrackViewCos = vc.f()[i];
      p->mTrackProjX = px.f()[i]; p->mTrackProjY = py.f()[i]; p->mTrackProjXR = pxr.f()[i];
      pts.push_back(p);
    }
    ORBmatcher m(0.8f, true);
    put_i1("m1_n", m.SearchByProjection(F, pts, in("m1_th").f()[0]));
    put_i("m1_out", ids(F.mvpMapPoints));
  }
  {  // (2) SearchByProjection(CurrentFrame, LastFrame)
    for (int mono = 0; mono < 2; mono++) {
      reset_points();
      Frame Cur, Last; make_frame(Cur, "C"); make_frame(Last, "D");
      ORBmatcher m(0.9f, true);
      const std::string t = mono ? "m2m" : "m2";
      put_i1(t + "_n", m.SearchByProjection(Cur, Last, in("m2_th").f()[0], mono != 0));
      put_i(t + "_out", ids(Cur.mvpMapPoints));
    }
  }
  {  // (3) SearchByProjection(CurrentFrame, KeyFrame, sAlreadyFound)
    reset_points();
    Frame Cur; make_frame(Cur, "C");
    KeyFrame KF; make_kf(KF, "A");
    std::set<MapPoint*> found;
    const Arr& af = in("m3_found");
    for (size_t i = 0; i < af.n; i++) found.insert(&g_mp[af.i()[i]]);
    ORBmatcher m(0.9f, true);
    put_i1("m3_n", m.SearchByProjection(Cur, &KF, found, in("m3_th").f()[0], 100));
    put_i("m3_out", ids(Cur.mvpMapPoints));
  }
  {  // (4) SearchByProjection(KeyFrame, Scw, vpPoints, vpMatched)
    reset_points();
    KeyFrame KF; make_kf(KF, "B");
    std::vector<MapPoint*> pts, matched(KF.N, nullptr);
    const Arr &sel = in("m4_points"), &pre = in("m4_matched");
    for (size_t i = 0; i < sel.n; i++) pts.push_back(&g_mp[sel.i()[i]]);
    for (int i = 0; i < KF.N; i++) if (pre.i()[i] >= 0) matched[i] = &g_mp[pre.i()[i]];
    ORBmatcher m(0.75f, true);
    put_i1("m4_n", m.SearchByProjection(&KF, matf(in("m4_Scw").f(), 4, 4), pts, matched, 10));
    put_i("m4_out", ids(matched));
  }
  {  // (5) SearchByBoW(KeyFrame, Frame)  (6) SearchByBoW(KeyFrame, KeyFrame)
    reset_points();
    KeyFrame A, B; make_kf(A, "A"); make_kf(B, "B");
    Frame F; make_frame(F, "C");
    std::vector<MapPoint*> out;
    ORBmatcher m(0.7f, true);
    put_i1("m5_n", m.SearchByBoW(&A, F, out));
    put_i("m5_out", ids(out));
    ORBmatcher m6(0.75f, true);
    put_i1("m6_n", m6.SearchByBoW(&A, &B, out));
    put_i("m6_out", ids(out));
  }
  {  // (7) SearchForInitialization
    reset_points();
    Frame F1, F2; make_frame(F1, "C"); make_frame(F2, "D");
    std::vector<cv::Point2f> prev(F1.N);
    for (int i = 0; i < F1.N; i++) prev[i] = F1.mvKeysUn[i].pt;
    std::vector<int> m12;
    ORBmatcher m(0.9f, true);
    put_i1("m7_n", m.SearchForInitialization(F1, F2, prev, m12, 100));
    put_i("m7_out", std::vector<int32_t>(m12.begin(), m12.end()));
    std::vector<float> pv;
    for (auto& p : prev) { pv.push_back(p.x); pv.push_back(p.y); }
    put("m7_prev", 2, pv.data(), pv.size());
  }
  {  // (8) SearchForTriangulation
    for (int only = 0; only < 2; only++) {
      reset_points();
      KeyFrame A, B; make_kf(A, "A"); make_kf(B, "B");
      std::vector<std::pair<size_t, size_t> > pairs;
      ORBmatcher m(0.6f, false);
      const int n = m.SearchForTriangulation(&A, &B, matf(in("m8_F12").f(), 3, 3), pairs, only != 0);
      std::vector<int32_t> flat;
      for (auto& p : pairs) { flat.push_back((int32_t)p.first); flat.push_back((int32_t)p.second); }
      put_i1(only ? "m8s_n" : "m8_n", n);
      put_i(only ? "m8s_out" : "m8_out", flat);
    }
  }
  {  // (9) SearchBySim3
    reset_points();
    KeyFrame A, B; make_kf(A, "A"); make_kf(B, "B");
    std::vector<MapPoint*> m12(A.N, nullptr);
    const Arr& pre = in("m9_pre");
    for (int i = 0; i < A.N; i++) if (pre.i()[i] >= 0) m12[i] = &g_mp[pre.i()[i]];
    ORBmatcher m(0.75f, true);
    const float s12 = in("m9_s12").f()[0];
    put_i1("m9_n", m.SearchBySim3(&A, &B, m12, s12, matf(in("m9_R12").f(), 3, 3), matf(in("m9_t12").f(), 3, 1), 7.5f));
    put_i("m9_out", ids(m12));
  }
  {  // (10) Fuse(KeyFrame, vpMapPoints)
    reset_points();
    KeyFrame KF; make_kf(KF, "B");
    std::vector<MapPoint*> pts;
    const Arr& sel = in("m10_points");
    for (size_t i = 0; i < sel.n; i++) pts.push_back(sel.i()[i] >= 0 ? &g_mp[sel.i()[i]] : nullptr);
    ORBmatcher m(0.6f, true);
    put_i1("m10_n", m.Fuse(&KF, pts, 3.0f));
    put_i("m10_kf", ids(KF.mvpMapPoints));
    std::vector<int32_t> rep(g_mp.size()), bad(g_mp.size()), nobs(g_mp.size());
    for (size_t i = 0; i < g_mp.size(); i++) { rep[i] = id_of(g_mp[i].replacedBy); bad[i] = g_mp[i].bad; nobs[i] = g_mp[i].nObs; }
    put_i("m10_replaced", rep); put_i("m10_bad", bad); put_i("m10_nobs", nobs);
  }
  {  // (11) Fuse(KeyFrame, Scw, vpPoints, th, vpReplacePoint)
    reset_points();
    KeyFrame KF; make_kf(KF, "B");
    std::vector<MapPoint*> pts;
    const Arr& sel = in("m11_points");
    for (size_t i = 0; i < sel.n; i++) pts.push_back(&g_mp[sel.i()[i]]);
    std::vector<MapPoint*> rep(pts.size(), nullptr);
    ORBmatcher m(0.8f, true);
    put_i1("m11_n", m.Fuse(&KF, matf(in("m4_Scw").f(), 4, 4), pts, 4.0f, rep));
    put_i("m11_rep", ids(rep));
    put_i("m11_kf", ids(KF.mvpMapPoints));
  }
  std::fclose(g_out);
  return 0;
}
