#include "octree_host.h"

#include <algorithm>
#include <cmath>

namespace orbfe {

namespace {
struct Node {
  int x0, x1, y0, y1;  // UL.x, UR.x, UL.y, BL.y (nodes are axis-aligned rectangles)
  int begin, end;      // key range inside the current key-index array
  int seq;             // creation order (stands in for the reference's pointer value)
  bool noMore;
};

struct Tree {
  const Candidate* cand;
  std::vector<int> keys, scratch;  // candidate indices grouped by node (stable order)
  std::vector<Node> list;          // current std::list<ExtractorNode> content, in list order
  int seq = 0;

  float kx(int id) const { return (float)(cand[id].xy & 0xffffu); }
  float ky(int id) const { return (float)(cand[id].xy >> 16); }

  // ExtractorNode::DivideNode (:498-558): stable 4-way partition of the parent's keys.
  // Children are returned in creation order n1..n4 (empty ones included, count 0).
  void divide(const Node& p, Node ch[4]) {
    const int halfX = (int)std::ceil((float)(p.x1 - p.x0) / 2);
    const int halfY = (int)std::ceil((float)(p.y1 - p.y0) / 2);
    const int mx = p.x0 + halfX, my = p.y0 + halfY;
    ch[0] = {p.x0, mx, p.y0, my, 0, 0, 0, false};
    ch[1] = {mx, p.x1, p.y0, my, 0, 0, 0, false};
    ch[2] = {p.x0, mx, my, p.y1, 0, 0, 0, false};
    ch[3] = {mx, p.x1, my, p.y1, 0, 0, 0, false};
    int cnt[4] = {0, 0, 0, 0};
    const float fmx = (float)mx, fmy = (float)my;
    auto quad = [&](int id) {
      if (kx(id) < fmx) return ky(id) < fmy ? 0 : 2;
      return ky(id) < fmy ? 1 : 3;
    };
    for (int i = p.begin; i < p.end; i++) cnt[quad(keys[i])]++;
    int pos[4];
    pos[0] = p.begin;
    for (int k = 1; k < 4; k++) pos[k] = pos[k - 1] + cnt[k - 1];
    for (int k = 0; k < 4; k++) { ch[k].begin = pos[k]; ch[k].end = pos[k] + cnt[k]; }
    for (int i = p.begin; i < p.end; i++) { const int id = keys[i]; scratch[pos[quad(id)]++] = id; }
    std::copy(scratch.begin() + p.begin, scratch.begin() + p.end, keys.begin() + p.begin);
    for (int k = 0; k < 4; k++) {
      ch[k].seq = seq++;
      ch[k].noMore = (cnt[k] == 1);
    }
  }
};
}  // namespace

int distribute_octree_host(const Candidate* cand, int n, int minX, int maxX, int minY, int maxY,
                           int N, LevelKp* out, int outCap) {
  if (n <= 0) return 0;
  const int nIni = (int)std::round((float)(maxX - minX) / (float)(maxY - minY));
  if (nIni <= 0) return 0;  // the reference assumes width > height (:560)
  const float hX = (float)(maxX - minX) / (float)nIni;
  Tree t;
  t.cand = cand;
  t.keys.resize(n);
  t.scratch.resize(n);
  // roots (:576-600): stable bucket of the keys by (int)(x / hX)
  std::vector<int> bucketCnt(nIni, 0);
  auto bucket = [&](int id) {
    int b = (int)(t.kx(id) / hX);
    return b >= nIni ? nIni - 1 : b;
  };
  for (int i = 0; i < n; i++) bucketCnt[bucket(i)]++;
  std::vector<int> pos(nIni, 0);
  for (int b = 1; b < nIni; b++) pos[b] = pos[b - 1] + bucketCnt[b - 1];
  std::vector<int> start = pos;
  for (int i = 0; i < n; i++) t.keys[pos[bucket(i)]++] = i;
  for (int b = 0; b < nIni; b++) {
    if (bucketCnt[b] == 0) continue;  // :608-619 erases empty roots
    Node nd{(int)(hX * (float)b), (int)(hX * (float)(b + 1)), 0, maxY - minY,
            start[b], start[b] + bucketCnt[b], t.seq++, bucketCnt[b] == 1};
    t.list.push_back(nd);
  }
  std::vector<Node> created, next;
  std::vector<int> expandable;  // indices into `created` of children with > 1 key
  bool finish = false;
  Node ch[4];
  while (!finish) {
    // ---- split-everything pass (:625-700) ----
    const int prevSize = (int)t.list.size();
    created.clear();
    next.clear();
    for (const Node& nd : t.list) {
      if (nd.noMore) { next.push_back(nd); continue; }
      t.divide(nd, ch);
      for (int k = 0; k < 4; k++)
        if (ch[k].end > ch[k].begin) created.push_back(ch[k]);
    }
    t.list.assign(created.rbegin(), created.rend());
    t.list.insert(t.list.end(), next.begin(), next.end());
    int nToExpand = 0;
    for (const Node& c : created) nToExpand += (c.end - c.begin > 1);
    const int size = (int)t.list.size();
    if (size >= N || size == prevSize) {
      finish = true;
    } else if (size + nToExpand * 3 > N) {
      // ---- largest-first passes (:714-782) ----
      // `cur` = nodes recorded in vSizeAndPointerToNode, identified by seq
      std::vector<Node> cur;
      for (const Node& c : created)
        if (c.end - c.begin > 1) cur.push_back(c);
      while (!finish) {
        const int prev2 = (int)t.list.size();
        std::sort(cur.begin(), cur.end(), [](const Node& a, const Node& b) {
          const int ca = a.end - a.begin, cb = b.end - b.begin;
          return ca != cb ? ca < cb : a.seq < b.seq;
        });
        std::vector<Node> made;
        std::vector<int> erasedSeq;
        int sz = prev2;
        for (int j = (int)cur.size() - 1; j >= 0; j--) {
          t.divide(cur[j], ch);
          for (int k = 0; k < 4; k++)
            if (ch[k].end > ch[k].begin) { made.push_back(ch[k]); sz++; }
          erasedSeq.push_back(cur[j].seq);
          sz--;
          if (sz >= N) break;
        }
        std::sort(erasedSeq.begin(), erasedSeq.end());
        next.clear();
        next.assign(made.rbegin(), made.rend());
        for (const Node& nd : t.list)
          if (!std::binary_search(erasedSeq.begin(), erasedSeq.end(), nd.seq)) next.push_back(nd);
        t.list.swap(next);
        cur.clear();
        for (const Node& c : made)
          if (c.end - c.begin > 1) cur.push_back(c);
        if ((int)t.list.size() >= N || (int)t.list.size() == prev2) finish = true;
      }
    }
  }
  // best response per node, first wins ties (:787-805); pt += minBorder (:909-916)
  int nout = 0;
  for (const Node& nd : t.list) {
    int best = t.keys[nd.begin];
    uint32_t bestScore = cand[best].score;
    for (int i = nd.begin + 1; i < nd.end; i++) {
      const int id = t.keys[i];
      if (cand[id].score > bestScore) { best = id; bestScore = cand[id].score; }
    }
    if (nout < outCap) {
      out[nout].x = (uint16_t)((cand[best].xy & 0xffffu) + minX);
      out[nout].y = (uint16_t)((cand[best].xy >> 16) + minY);
      out[nout].score = (uint16_t)bestScore;
      out[nout].rank = (uint16_t)nout;
    }
    nout++;
  }
  return nout;
}

}  // namespace orbfe
