// Partial-order alignment consensus for low-quality-region pseudo-seeds.
//
// Host-side (CPU) component of the consensus engine: < 1 % of the reference's
// time, <= 6 sequences per call.  Behaviour follows
// the reference's lib/dag.c:658-694 `poa_to_consensus` and its callees
// (graph growth dag.c:345-401, group-wise DFS topological order dag.c:403-508,
// NW of sequence vs DAG dag.c:261-343, heaviest path dag.c:555-595), including
// its observable quirks:
//   * the unmatched tail inserted after the last matched query base is one
//     element longer than the tail, i.e. it also inserts the terminating NUL of
//     the query as a node (dag.c:354); callers strlen() the result;
//   * score ties: deletion wins over match only on ">=", evaluated per in-edge
//     in insertion order (dag.c:284-285);
//   * the consensus path picks the FIRST node in topological order with the
//     strictly greatest score (dag.c:579).
// The data structures are our own (index vectors instead of fixed 50-slot
// arrays); the reference's fixed-capacity overflow cases (SEQ_MAX_COUNT 50 edges
// per node) cannot be reached with <= 6 sequences.
#include "nd_host.h"

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace ndgpu {
namespace {

constexpr long kGap = -2;
inline long sub_score(char a, unsigned char b) { return (unsigned char)a == b ? 1 : -2; }

struct PNode {
    unsigned char base = 0;
    std::vector<uint32_t> in_e, out_e;      // edge ids in insertion order
    std::vector<uint16_t> aligned;          // nodes sharing this column
    int32_t best_prev = -1;
    double best_score = 0;
};

struct PEdge {
    uint16_t from = 0, to = 0;
    uint64_t labels = 0;                    // bit i: sequence i walks this edge
};

struct Cell {   // 8 bytes: the matrix of a 300-node graph and a 300-base sequence is what this function spends its time moving
    int32_t s;
    uint16_t px, py;
};

struct Route {
    int32_t node = -1;   // graph node matched (or -1)
    int32_t qpos = -1;   // query offset consumed (or -1)
};

class Graph {
  public:
    std::vector<PNode> nodes;
    std::vector<PEdge> edges;
    std::vector<uint16_t> order;            // topological order (node ids)

    int add_node(unsigned char b) {
        nodes.emplace_back();
        nodes.back().base = b;
        return (int)nodes.size() - 1;
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
        for (uint32_t id : nodes[from].out_e)
            if (edges[id].to == to) {
                edges[id].labels |= 1ull << seq;
                missing = false;
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
    std::string heaviest_path(int nseq) const;
};

// dag.c:469-508 + 403-467.  Aligned nodes form one group ("pnid"); groups are
// emitted by an iterative DFS whose stack discipline we keep verbatim because
// the resulting order breaks score ties downstream.
void Graph::toposort() {
    const int n = (int)nodes.size();
    std::vector<int32_t> grp_of(n, -1);
    std::vector<uint16_t> grp_head;
    for (int i = 0; i < n; i++) {
        if (grp_of[i] != -1) continue;
        int g = (int)grp_head.size();
        grp_head.push_back((uint16_t)i);
        grp_of[i] = g;
        for (uint16_t a : nodes[i].aligned) grp_of[a] = g;
    }
    const int ng = (int)grp_head.size();
    std::vector<int8_t> done(ng, -1);
    order.assign(n, 0);
    int fill = n - 1;
    std::vector<uint16_t> stack;
    std::vector<int8_t> started;

    auto has_pred = [&](int node) {
        unsigned c = (unsigned)nodes[node].in_e.size();
        for (size_t j = 0; j < nodes[node].aligned.size() && c == 0; j++)
            c += (unsigned)nodes[nodes[node].aligned[j]].in_e.size();
        return c != 0;
    };

    while (fill >= 0) {
        int root = -1;
        for (int g = 0; g < ng; g++)
            if (done[g] == -1 && !has_pred(grp_head[g])) {
                root = g;
                break;
            }
        if (root < 0) break;  // reference asserts; unreachable for DAGs built here
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
                for (uint16_t a : h.aligned) order[fill--] = a;
                started[g] = -1;
                continue;
            }
            started[g] = 1;
            stack.push_back(g);
            for (uint32_t id : h.out_e) stack.push_back((uint16_t)grp_of[edges[id].to]);
            for (uint16_t a : h.aligned)
                for (uint32_t id : nodes[a].out_e) stack.push_back((uint16_t)grp_of[edges[id].to]);
        }
    }
}

void Graph::add_sequence(int seq, const char *s, int len) {
    const int X = (int)nodes.size();
    const int Y = len;
    const size_t W = (size_t)Y + 1;
    // (one buffer per thread, not cleared: row 0, column 0 and every interior cell are written before they are read)
    static thread_local std::vector<Cell> dp_buf;
    if (dp_buf.size() < (size_t)(X + 1) * W) dp_buf.resize((size_t)(X + 1) * W);
    Cell *const dp = dp_buf.data();
    auto at = [&](int r, int c) -> Cell & { return dp[(size_t)r * W + c]; };
    std::vector<uint16_t> rank(X);

    // dag.c:88-134 boundary scores
    for (int c = 0; c <= Y; c++) at(0, c).s = (int32_t)(c * kGap), at(0, c).px = at(0, c).py = 0;
    for (int i = 0; i < X; i++) {
        int v = order[i];
        rank[v] = (uint16_t)i;
        long b;
        if (nodes[v].in_e.empty()) b = 0;
        else {
            b = at(rank[edges[nodes[v].in_e[0]].from] + 1, 0).s;
            for (size_t k = 1; k < nodes[v].in_e.size(); k++) {
                long t = at(rank[edges[nodes[v].in_e[k]].from] + 1, 0).s;
                if (t > b) b = t;
            }
        }
        at(i + 1, 0).s = (int32_t)(b + kGap), at(i + 1, 0).px = at(i + 1, 0).py = 0;
    }

    // dag.c:261-300 fill
    std::vector<int> preds;
    for (int i = 0; i < X; i++) {
        const PNode &nd = nodes[order[i]];
        preds.clear();
        for (uint32_t id : nd.in_e) preds.push_back(rank[edges[id].from] + 1);
        if (preds.empty()) preds.push_back(0);
        Cell *const row = dp + (size_t)(i + 1) * W;
        const size_t np = preds.size();
        for (int j = 0; j < Y; j++) {
            long best = row[j].s + kGap;
            int bx = i + 1, by = j;
            const long sub = sub_score(s[j], nd.base);
            for (size_t k = 0; k < np; k++) {   // per in-edge in insertion order (score ties: dag.c:284-285)
                const int pr = preds[k];
                const Cell *const pw = dp + (size_t)pr * W + j;
                const long del = pw[1].s + kGap;
                const long mat = pw[0].s + sub;
                if (del > best && del >= mat) { best = del; bx = pr; by = j + 1; }
                else if (mat > best && mat >= del) { best = mat; bx = pr; by = j; }
            }
            Cell &c = row[j + 1];
            c.s = (int32_t)best;
            c.px = (uint16_t)bx;
            c.py = (uint16_t)by;
        }
    }

    // dag.c:302-313 best sink row
    int bx = 0;
    {
        long bs = 0;
        bool any = false;
        for (int i = 0; i < X; i++)
            if (nodes[order[i]].out_e.empty()) {
                long v = at(i + 1, Y).s;
                if (!any || v > bs) { bx = i + 1; bs = v; any = true; }
            }
    }
    // dag.c:327-343 route (collected backwards, then reversed)
    std::vector<Route> route;
    long start_q = -1, end_q = -1;
    {
        int x = bx, y = Y;
        while (x != 0 || y != 0) {
            int nx = at(x, y).px, ny = at(x, y).py;
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
            for (uint16_t a : nodes[r.node].aligned)
                if (nodes[a].base == b) found = a;
            if (found != -1) node = found;
            else {
                node = add_node(b);
                is_new = true;
                // dag.c:190-202,369-372: new node joins the column of r.node
                std::vector<uint16_t> col;
                col.push_back((uint16_t)r.node);
                for (uint16_t a : nodes[r.node].aligned) col.push_back(a);
                nodes[node].aligned = col;
                for (uint16_t a : col) nodes[a].aligned.push_back((uint16_t)node);
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
std::string Graph::heaviest_path(int nseq) const {
    std::vector<double> sc(nodes.size(), 0);
    std::vector<int32_t> prev(nodes.size(), -1);
    const uint64_t mask = nseq >= 64 ? ~0ull : ((1ull << nseq) - 1);
    double best = -1, gbest = -1;
    int gnode = -1;
    for (size_t i = 0; i < order.size(); i++) {
        int v = order[i];
        int bp = -1;
        const PNode &nd = nodes[v];
        if (!nd.in_e.empty()) {
            for (uint32_t id : nd.in_e) {
                const PEdge &e = edges[id];
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
    Graph g;
    for (size_t i = 0; i < seqs.size(); i++) {
        const std::string &s = seqs[i];
        if (i == 0) {
            int first = -1, head = -1;
            g.add_chain(0, s.c_str(), s.size(), first, head);
            g.order.resize(g.nodes.size());
            for (size_t x = 0; x < g.nodes.size(); x++) g.order[x] = (uint16_t)x;
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
