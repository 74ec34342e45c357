// Partial-order alignment consensus for low-quality-region pseudo-seeds.
//
// Host-side (CPU) component of the consensus engine, <= 6 sequences per call from the engine (lib/nextcorrect.c:456-462), any
// number through the exported poa_to_consensus.  Behaviour follows the reference's lib/dag.c:658-694 `poa_to_consensus` and its
// callees (graph growth dag.c:345-401, group-wise DFS topological order dag.c:403-508, NW of sequence vs DAG dag.c:261-343,
// heaviest path dag.c:555-595), including its observable quirks:
//   * the unmatched tail inserted after the last matched query base is one
//     element longer than the tail, i.e. it also inserts the terminating NUL of
//     the query as a node (dag.c:354); callers strlen() the result;
//   * score ties: deletion wins over match only on ">=", evaluated per in-edge
//     in insertion order (dag.c:284-285);
//   * the consensus path picks the FIRST node in topological order with the
//     strictly greatest score (dag.c:579);
//   * the boundary cells of the alignment matrix point straight at the origin (dag.c:88-134): a leading run of unmatched query
//     bases is ONE route element.
// The data structures are our own.  Round 5: this function was 2.2 of the 4.2 host CPU-seconds of a config-2 step, and the box
// grants the process 16 CPUs -- the matrix is now two planes (scores, 4 bytes; where a cell came from, 2 bytes) instead of 8-byte
// cells, a row is filled in two passes -- the candidates through the row's in-edges, which depend only on finished rows and
// vectorise (AVX2 clone chosen at load time), then the horizontal move, the one dependence along the row -- the node records hold
// their few edges inline, and every buffer lives in a per-thread workspace that is reused from call to call.
#include "nd_host.h"

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace ndgpu {
namespace {

constexpr int32_t kGap = -2;

// a few ids inline, the rest on the heap (a node of a <= 6-sequence graph has a handful of edges)
template <typename T, int N>
struct SmallVec {
    T in[N];
    uint32_t n = 0;
    std::vector<T> more;
    void clear() { n = 0, more.clear(); }
    uint32_t size() const { return n; }
    bool empty() const { return n == 0; }
    void push_back(T v) {
        if (n < (uint32_t)N) in[n] = v;
        else more.push_back(v);
        n++;
    }
    T operator[](uint32_t i) const { return i < (uint32_t)N ? in[i] : more[i - N]; }
};

struct PNode {
    unsigned char base = 0;
    SmallVec<uint32_t, 6> in_e, out_e;      // edge ids in insertion order
    SmallVec<uint16_t, 4> aligned;          // nodes sharing this column
    void reset(unsigned char b) {
        base = b;
        in_e.clear(), out_e.clear(), aligned.clear();
    }
};

struct PEdge {
    uint16_t from = 0, to = 0;
    uint64_t labels = 0;                    // bit i: sequence i walks this edge
};

struct Route {
    int32_t node = -1;   // graph node matched (or -1)
    int32_t qpos = -1;   // query offset consumed (or -1)
};

// where a cell came from: [1:0] kind, [15:2] which in-edge of the row's node
enum : uint16_t { FROM_H = 0, FROM_MAT = 1, FROM_DEL = 2, FROM_ORIGIN = 3 };

// Candidates of one row through one in-edge (dag.c:281-290): del = the cell above-right + gap, mat = the cell above + substitution;
// del wins a tie between the two (">="); `first` starts the row's candidates, later in-edges replace them only when strictly better.
// prev = scores of the in-edge's row (Y + 1 of them), q = the query, out: cv / ct for columns 1..Y (index j = column j + 1).
#if defined(__x86_64__) && defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__) && !defined(NDGPU_NO_TARGET_CLONES)  // (host pass only)
__attribute__((target_clones("avx2", "default")))
#endif
void row_candidates(const int32_t *__restrict__ prev, const char *__restrict__ q, unsigned char base, int Y, uint16_t k, bool first,
                    int32_t *__restrict__ cv, uint16_t *__restrict__ ct) {
    const uint16_t t_del = (uint16_t)(k << 2 | FROM_DEL), t_mat = (uint16_t)(k << 2 | FROM_MAT);
    if (first) {
        for (int j = 0; j < Y; j++) {
            const int32_t del = prev[j + 1] + kGap;
            const int32_t mat = prev[j] + ((unsigned char)q[j] == base ? 1 : -2);
            const bool d = del >= mat;
            cv[j] = d ? del : mat;
            ct[j] = d ? t_del : t_mat;
        }
    } else {
        for (int j = 0; j < Y; j++) {
            const int32_t del = prev[j + 1] + kGap;
            const int32_t mat = prev[j] + ((unsigned char)q[j] == base ? 1 : -2);
            const bool d = del >= mat;
            const int32_t c = d ? del : mat;
            const bool better = c > cv[j];
            cv[j] = better ? c : cv[j];
            ct[j] = better ? (d ? t_del : t_mat) : ct[j];
        }
    }
}

class Graph {
  public:
    std::vector<PNode> nodes;   // (capacity kept between calls: n_nodes counts the live ones)
    size_t n_nodes = 0;
    std::vector<PEdge> edges;
    std::vector<uint16_t> order;            // topological order (node ids)

    void clear() {
        n_nodes = 0;
        edges.clear();
        order.clear();
    }
    int add_node(unsigned char b) {
        if (n_nodes == nodes.size()) nodes.emplace_back();
        nodes[n_nodes].reset(b);
        return (int)n_nodes++;
    }
    void add_edge(int from, int to, int seq) {
        PEdge e;
        e.from = (uint16_t)from;
        e.to = (uint16_t)to;
        e.labels = 1ull << seq;
        edges.push_back(e);
        uint32_t id = (uint32_t)edges.size() - 1;
        nodes[from].out_e.push_back(id);
        nodes[to].in_e.push_back(id);
    }
    // dag.c:223-234: mark every existing from->to edge; report whether none existed
    bool label_existing(int from, int to, int seq) {
        bool missing = true;
        const PNode &f = nodes[from];
        for (uint32_t x = 0; x < f.out_e.size(); x++) {
            const uint32_t id = f.out_e[x];
            if (edges[id].to == to) {
                edges[id].labels |= 1ull << seq;
                missing = false;
            }
        }
        return missing;
    }
    // dag.c:236-250
    void add_chain(int seq, const char *s, size_t n, int &first, int &head) {
        for (size_t i = 0; i < n; i++) {
            int id = add_node((unsigned char)s[i]);
            if (first == -1) first = id;
            else add_edge(head, id, seq);
            head = id;
        }
    }

    void toposort();
    void add_sequence(int seq, const char *s, int len);
    std::string heaviest_path(int nseq);

  private:
    // workspace, reused
    std::vector<int32_t> grp_of_;
    std::vector<uint16_t> grp_head_, stack_, rank_, ct_, from_;
    std::vector<int8_t> done_, started_;
    std::vector<int32_t> score_, cv_, preds_;
    std::vector<uint32_t> pred_off_;
    std::vector<Route> route_;
    std::vector<double> hp_sc_;
    std::vector<int32_t> hp_prev_;
};

// dag.c:469-508 + 403-467.  Aligned nodes form one group ("pnid"); groups are
// emitted by an iterative DFS whose stack discipline we keep verbatim because
// the resulting order breaks score ties downstream.
void Graph::toposort() {
    const int n = (int)n_nodes;
    std::vector<int32_t> &grp_of = grp_of_;
    std::vector<uint16_t> &grp_head = grp_head_;
    grp_of.assign(n, -1);
    grp_head.clear();
    for (int i = 0; i < n; i++) {
        if (grp_of[i] != -1) continue;
        int g = (int)grp_head.size();
        grp_head.push_back((uint16_t)i);
        grp_of[i] = g;
        const PNode &nd = nodes[i];
        for (uint32_t x = 0; x < nd.aligned.size(); x++) grp_of[nd.aligned[x]] = g;
    }
    const int ng = (int)grp_head.size();
    std::vector<int8_t> &done = done_, &started = started_;
    done.assign(ng, -1);
    order.assign(n, 0);
    int fill = n - 1;
    std::vector<uint16_t> &stack = stack_;

    auto has_pred = [&](int node) {
        const PNode &nd = nodes[node];
        unsigned c = nd.in_e.size();
        for (uint32_t j = 0; j < nd.aligned.size() && c == 0; j++) c += nodes[nd.aligned[j]].in_e.size();
        return c != 0;
    };

    int scan_from = 0;  // groups before this one are done or have a predecessor that no later step removes
    while (fill >= 0) {
        int root = -1;
        for (int g = scan_from; g < ng; g++)
            if (done[g] == -1 && !has_pred(grp_head[g])) {
                root = g;
                break;
            }
        if (root < 0) break;  // reference asserts; unreachable for DAGs built here
        scan_from = root + 1; // (the reference scans from 0 every time: the groups it passed over stay passed over -- `done` only grows
                              //  and has_pred does not change during a sort -- so the next root is the same one)
        started.assign(ng, -1);
        stack.clear();
        stack.push_back((uint16_t)root);
        while (!stack.empty()) {
            uint16_t g = stack.back();
            stack.pop_back();
            if (done[g] == 1) continue;
            const PNode &h = nodes[grp_head[g]];
            if (started[g] != -1) {
                done[g] = 1;
                order[fill--] = grp_head[g];
                for (uint32_t x = 0; x < h.aligned.size(); x++) order[fill--] = h.aligned[x];
                started[g] = -1;
                continue;
            }
            started[g] = 1;
            stack.push_back(g);
            for (uint32_t x = 0; x < h.out_e.size(); x++) stack.push_back((uint16_t)grp_of[edges[h.out_e[x]].to]);
            for (uint32_t x = 0; x < h.aligned.size(); x++) {
                const PNode &a = nodes[h.aligned[x]];
                for (uint32_t y = 0; y < a.out_e.size(); y++) stack.push_back((uint16_t)grp_of[edges[a.out_e[y]].to]);
            }
        }
    }
}

void Graph::add_sequence(int seq, const char *s, int len) {
    const int X = (int)n_nodes;
    const int Y = len;
    const size_t W = (size_t)Y + 1;
    // (not cleared: row 0, column 0 and every interior cell are written before they are read)
    if (score_.size() < (size_t)(X + 1) * W) score_.resize((size_t)(X + 1) * W), from_.resize((size_t)(X + 1) * W);
    int32_t *const S = score_.data();
    uint16_t *const F = from_.data();
    if (cv_.size() < W) cv_.resize(W), ct_.resize(W);
    std::vector<uint16_t> &rank = rank_;
    rank.resize(X);

    // in-edge rows of every node's row, in insertion order (row of node v = rank[v] + 1; a node without in-edges hangs on row 0)
    pred_off_.resize((size_t)X + 1);
    preds_.clear();
    for (int i = 0; i < X; i++) rank[order[i]] = (uint16_t)i;
    for (int i = 0; i < X; i++) {
        const PNode &nd = nodes[order[i]];
        pred_off_[i] = (uint32_t)preds_.size();
        for (uint32_t k = 0; k < nd.in_e.size(); k++) preds_.push_back((int32_t)rank[edges[nd.in_e[k]].from] + 1);
        if (nd.in_e.empty()) preds_.push_back(0);
    }
    pred_off_[X] = (uint32_t)preds_.size();

    // dag.c:88-134 boundary scores; the boundary cells point at the origin
    for (int c = 0; c <= Y; c++) S[c] = c * kGap, F[c] = FROM_ORIGIN;
    for (int i = 0; i < X; i++) {
        const PNode &nd = nodes[order[i]];
        int32_t b;
        if (nd.in_e.empty()) b = 0;
        else {
            b = S[(size_t)preds_[pred_off_[i]] * W];
            for (uint32_t k = pred_off_[i] + 1; k < pred_off_[i + 1]; k++) {
                const int32_t t = S[(size_t)preds_[k] * W];
                if (t > b) b = t;
            }
        }
        S[(size_t)(i + 1) * W] = b + kGap;
        F[(size_t)(i + 1) * W] = FROM_ORIGIN;
    }

    // dag.c:261-300 fill.  The reference starts a cell's best at the horizontal move and lets every in-edge's deletion / match replace
    // it when strictly better, in edge order: the first strict maximum of [horizontal, edge 0, edge 1, ...] wins.  The edges'
    // candidates depend on finished rows only (pass 1, vectorised); the horizontal move is the one dependence along the row (pass 2).
    int32_t *const cv = cv_.data();
    uint16_t *const ct = ct_.data();
    for (int i = 0; i < X; i++) {
        const unsigned char base = nodes[order[i]].base;
        const uint32_t p0 = pred_off_[i], p1 = pred_off_[i + 1];
        for (uint32_t k = p0; k < p1; k++)
            row_candidates(S + (size_t)preds_[k] * W, s, base, Y, (uint16_t)(k - p0), k == p0, cv, ct);
        int32_t *const row = S + (size_t)(i + 1) * W;
        uint16_t *const frow = F + (size_t)(i + 1) * W;
        int32_t left = row[0];
        for (int j = 0; j < Y; j++) {
            const int32_t h = left + kGap;
            const bool up = cv[j] > h;
            left = up ? cv[j] : h;
            row[j + 1] = left;
            frow[j + 1] = up ? ct[j] : (uint16_t)FROM_H;
        }
    }

    // dag.c:302-313 best sink row
    int bx = 0;
    {
        long bs = 0;
        bool any = false;
        for (int i = 0; i < X; i++)
            if (nodes[order[i]].out_e.empty()) {
                long v = S[(size_t)(i + 1) * W + Y];
                if (!any || v > bs) { bx = i + 1; bs = v; any = true; }
            }
    }
    // dag.c:327-343 route (collected backwards, then reversed)
    std::vector<Route> &route = route_;
    route.clear();
    long start_q = -1, end_q = -1;
    {
        int x = bx, y = Y;
        while (x != 0 || y != 0) {
            const uint16_t f = F[(size_t)x * W + y];
            int nx, ny;
            switch (f & 3u) {
                case FROM_H: nx = x, ny = y - 1; break;
                case FROM_MAT: nx = preds_[pred_off_[x - 1] + (f >> 2)], ny = y - 1; break;
                case FROM_DEL: nx = preds_[pred_off_[x - 1] + (f >> 2)], ny = y; break;
                default: nx = 0, ny = 0; break;
            }
            Route r;
            if (nx != x) r.node = order[x - 1];
            if (ny != y) {
                r.qpos = y - 1;
                start_q = y - 1;
                if (end_q == -1) end_q = r.qpos;
            }
            route.push_back(r);
            x = nx;
            y = ny;
        }
        for (size_t a = 0, b = route.size(); a + 1 < b; a++, b--) std::swap(route[a], route[b - 1]);
    }

    // dag.c:345-401 thread the sequence through the graph
    int first = -1, head = -1, tail_first = -1, tail_last = -1;
    if (start_q > 0) add_chain(seq, s, (size_t)start_q, first, head);
    if (end_q < Y - 1) add_chain(seq, s + end_q + 1, (size_t)(Y - end_q), tail_first, tail_last);  // includes s[Y] == NUL
    bool prev_new = true;
    for (const Route &r : route) {
        if (r.qpos == -1) continue;
        bool is_new = false;
        unsigned char b = (unsigned char)s[r.qpos];
        int node;
        if (r.node == -1) {
            node = add_node(b);
            is_new = true;
        } else if (nodes[r.node].base == b) {
            node = r.node;
        } else {
            int found = -1;
            {
                const PNode &rn = nodes[r.node];
                for (uint32_t x = 0; x < rn.aligned.size(); x++)
                    if (nodes[rn.aligned[x]].base == b) found = rn.aligned[x];
            }
            if (found != -1) node = found;
            else {
                node = add_node(b);   // (may move `nodes`: no reference into it is held across this call)
                is_new = true;
                // dag.c:190-202,369-372: new node joins the column of r.node
                PNode &nn = nodes[node];
                nn.aligned.push_back((uint16_t)r.node);
                const uint32_t na = nodes[r.node].aligned.size();
                for (uint32_t x = 0; x < na; x++) nn.aligned.push_back(nodes[r.node].aligned[x]);
                for (uint32_t x = 0; x < nn.aligned.size(); x++) nodes[nn.aligned[x]].aligned.push_back((uint16_t)node);
            }
        }
        if (head != -1) {
            if (is_new || prev_new) add_edge(head, node, seq);
            else if (label_existing(head, node, seq)) add_edge(head, node, seq);
        }
        head = node;
        prev_new = is_new;
        if (first == -1) first = head;
    }
    if (tail_first != -1) add_edge(head, tail_first, seq);
    toposort();
}

// dag.c:555-595
std::string Graph::heaviest_path(int nseq) {
    std::vector<double> &sc = hp_sc_;
    std::vector<int32_t> &prev = hp_prev_;
    sc.assign(n_nodes, 0);
    prev.assign(n_nodes, -1);
    const uint64_t mask = nseq >= 64 ? ~0ull : ((1ull << nseq) - 1);
    double best = -1, gbest = -1;
    int gnode = -1;
    for (size_t i = 0; i < order.size(); i++) {
        int v = order[i];
        int bp = -1;
        const PNode &nd = nodes[v];
        if (!nd.in_e.empty()) {
            for (uint32_t x = 0; x < nd.in_e.size(); x++) {
                const PEdge &e = edges[nd.in_e[x]];
                double s = sc[e.from] + __builtin_popcountll(e.labels & mask) - 0.5 * (double)nd.in_e.size();
                if (s > best || bp == -1) { best = s; bp = e.from; }
            }
        } else {
            best = 0;
            bp = -1;
        }
        sc[v] = best;
        prev[v] = bp;
        if (best > gbest) { gbest = best; gnode = v; }
    }
    std::string out;
    while (gnode != -1) {
        out.push_back((char)nodes[gnode].base);
        gnode = prev[gnode];
    }
    for (size_t a = 0, b = out.size(); a + 1 < b; a++, b--) std::swap(out[a], out[b - 1]);
    return out;
}

}  // namespace

// Returns the raw path (may contain an embedded NUL, see header comment); the
// caller truncates at the first NUL exactly as strlen() does in
// lib/nextcorrect.c:462.
std::string poa_consensus(const std::vector<std::string> &seqs) {
    static thread_local Graph g;   // (its buffers are the workspace of this thread's calls)
    g.clear();
    for (size_t i = 0; i < seqs.size(); i++) {
        const std::string &s = seqs[i];
        if (i == 0) {
            int first = -1, head = -1;
            g.add_chain(0, s.c_str(), s.size(), first, head);
            g.order.resize(g.n_nodes);
            for (size_t x = 0; x < g.n_nodes; x++) g.order[x] = (uint16_t)x;
        } else {
            g.add_sequence((int)i, s.c_str(), (int)s.size());
        }
    }
    std::string raw = g.heaviest_path((int)seqs.size());
    size_t z = raw.find('\0');
    if (z != std::string::npos) raw.resize(z);
    return raw;
}

}  // namespace ndgpu
