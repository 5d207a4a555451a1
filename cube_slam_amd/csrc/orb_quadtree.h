// orb_quadtree.h -- host-side keypoint distribution for the ORB extractor.
//
// Same selection as ORB_SLAM2::ORBextractor::DistributeOctTree / ExtractorNode::DivideNode
// (reference orb_object_slam/src/ORBextractor.cc:483-538, :540-763) but index based: nodes live in a pool, the node
// "list" is an intrusive doubly linked list of pool indices, a node's keypoints are a contiguous slice of a permutation
// array (children = stable 4-way partition of the parent's slice).  The reference breaks equal-size ties of its
// largest-first phase by heap address (:685); here by node creation order.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace cs_orb_host {

struct Cand { float x, y, response; };

class QuadTree {
  public:
    // K: candidates (x, y relative to minX/minY).  Returns indices into K of the retained keypoints, in list order.
    void distribute(const Cand *K, int n, int minX, int maxX, int minY, int maxY, int N, std::vector<int> &result) {
        result.clear();
        const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
        if (nIni < 1 || n == 0) return;
        const float hX = static_cast<float>(maxX - minX) / nIni;
        nodes_.clear(); nodes_.reserve((size_t)4 * N + 64); head_ = tail_ = -1; size_ = 0;
        perm_.resize(n); tmp_.resize(n);
        // root nodes: bucket the candidates by x / hX (stable)
        cnt_.assign(nIni + 1, 0);
        std::vector<int> &cnt = cnt_;
        bucket_.resize(n);
        for (int i = 0; i < n; i++) { int b = (int)(K[i].x / hX); if (b >= nIni) b = nIni - 1; bucket_[i] = b; cnt[b + 1]++; }
        for (int b = 0; b < nIni; b++) cnt[b + 1] += cnt[b];
        { std::vector<int> pos(cnt.begin(), cnt.end() - 1); for (int i = 0; i < n; i++) perm_[pos[bucket_[i]]++] = i; }
        for (int b = 0; b < nIni; b++) {
            Node nd;
            nd.x0 = (int)(hX * static_cast<float>(b)); nd.x1 = (int)(hX * static_cast<float>(b + 1));
            nd.y0 = 0; nd.y1 = maxY - minY;
            nd.begin = cnt[b]; nd.end = cnt[b + 1];
            if (nd.end == nd.begin) continue;                 // empty root nodes are erased (:575-586)
            nd.no_more = (nd.end - nd.begin) == 1;
            push_back(add(nd));
        }
        std::vector<std::pair<int, int>> &expand = expand_, &prev_expand = prev_expand_; // (size, node id); node ids grow with creation order
        std::vector<int> &pass = pass_;
        bool finish = false;
        while (!finish) {
            const int prev_size = size_;
            int n_to_expand = 0;
            expand.clear();
            // one pass over the nodes that existed at pass start, in list order; children go to the front
            pass.clear();
            for (int id = head_; id >= 0; id = nodes_[id].next) pass.push_back(id);
            for (int id : pass) {
                if (nodes_[id].no_more) continue;
                split(K, id, expand, n_to_expand);
                erase(id);
            }
            if (size_ >= N || size_ == prev_size) finish = true;
            else if (size_ + n_to_expand * 3 > N) {
                while (!finish) {
                    const int ps = size_;
                    prev_expand = expand;
                    expand.clear();
                    std::sort(prev_expand.begin(), prev_expand.end()); // (size, creation order) ascending
                    for (int j = (int)prev_expand.size() - 1; j >= 0; j--) {
                        int dummy = 0;
                        split(K, prev_expand[j].second, expand, dummy);
                        erase(prev_expand[j].second);
                        if (size_ >= N) break;
                    }
                    if (size_ >= N || size_ == ps) finish = true;
                }
            }
        }
        result.reserve(size_);
        for (int id = head_; id >= 0; id = nodes_[id].next) { // best response per node, first wins ties (:744-760)
            const Node &nd = nodes_[id];
            int best = perm_[nd.begin];
            float mx = K[best].response;
            for (int p = nd.begin + 1; p < nd.end; p++)
                if (K[perm_[p]].response > mx) { best = perm_[p]; mx = K[best].response; }
            result.push_back(best);
        }
    }

  private:
    struct Node { int x0, y0, x1, y1, begin, end, prev, next; bool no_more; };
    std::vector<Node> nodes_;
    std::vector<int> perm_, tmp_, bucket_, cnt_, pass_;
    std::vector<std::pair<int, int>> expand_, prev_expand_;
    int head_ = -1, tail_ = -1, size_ = 0;

    int add(const Node &n) { nodes_.push_back(n); return (int)nodes_.size() - 1; }
    void push_back(int id) {
        nodes_[id].prev = tail_; nodes_[id].next = -1;
        if (tail_ >= 0) nodes_[tail_].next = id; else head_ = id;
        tail_ = id; size_++;
    }
    void push_front(int id) {
        nodes_[id].next = head_; nodes_[id].prev = -1;
        if (head_ >= 0) nodes_[head_].prev = id; else tail_ = id;
        head_ = id; size_++;
    }
    void erase(int id) {
        int p = nodes_[id].prev, q = nodes_[id].next;
        if (p >= 0) nodes_[p].next = q; else head_ = q;
        if (q >= 0) nodes_[q].prev = p; else tail_ = p;
        size_--;
    }
    // DivideNode (:483-538): children n1..n4 = (left,top) (right,top) (left,bottom) (right,bottom); stable partition
    void split(const Cand *K, int id, std::vector<std::pair<int, int>> &expand, int &n_to_expand) {
        const Node nd = nodes_[id];
        const int halfX = (int)std::ceil(static_cast<float>(nd.x1 - nd.x0) / 2);
        const int halfY = (int)std::ceil(static_cast<float>(nd.y1 - nd.y0) / 2);
        const int mx = nd.x0 + halfX, my = nd.y0 + halfY;
        int c[4] = {0, 0, 0, 0};
        for (int p = nd.begin; p < nd.end; p++) {
            const Cand &k = K[perm_[p]];
            int q = (k.x < mx) ? ((k.y < my) ? 0 : 2) : ((k.y < my) ? 1 : 3);
            bucket_[p] = q; c[q]++;
        }
        int off[4] = {nd.begin, nd.begin + c[0], nd.begin + c[0] + c[1], nd.begin + c[0] + c[1] + c[2]};
        int pos[4] = {off[0], off[1], off[2], off[3]};
        for (int p = nd.begin; p < nd.end; p++) tmp_[pos[bucket_[p]]++] = perm_[p];
        for (int p = nd.begin; p < nd.end; p++) perm_[p] = tmp_[p];
        const int bx[4][4] = {{nd.x0, nd.y0, mx, my}, {mx, nd.y0, nd.x1, my}, {nd.x0, my, mx, nd.y1}, {mx, my, nd.x1, nd.y1}};
        for (int q = 0; q < 4; q++) {
            if (c[q] == 0) continue;
            Node ch;
            ch.x0 = bx[q][0]; ch.y0 = bx[q][1]; ch.x1 = bx[q][2]; ch.y1 = bx[q][3];
            ch.begin = off[q]; ch.end = off[q] + c[q];
            ch.no_more = c[q] == 1;
            int cid = add(ch);
            push_front(cid);
            if (c[q] > 1) { n_to_expand++; expand.push_back(std::make_pair(c[q], cid)); }
        }
    }
};

} // namespace cs_orb_host
