// linkage.cpp -- see linkage.hpp.  Built with -ffp-contract=off: every update is a plain f32
// operation sequence, so the steps do not depend on the compiler.
#include "linkage.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <stdexcept>

namespace pnh {

bool parse_cluster_method(const std::string &s, ClusterMethod &m) {
    std::string t;
    for (char c : s) t += (char)std::tolower((unsigned char)c);
    static const char *names[] = {"single", "complete", "average", "weighted", "ward", "centroid", "median"};
    for (int k = 0; k < 7; ++k)
        if (t == names[k]) {
            m = (ClusterMethod)k;
            return true;
        }
    return false;
}

namespace {

// symmetric view of the condensed upper triangle
struct Cond {
    float *d;
    size_t n;
    float &at(size_t a, size_t b) {
        if (a > b) std::swap(a, b);
        return d[n * a - a * (a + 1) / 2 + (b - a - 1)];
    }
};

// Lance-Williams updates in the form of kodama's method.rs: `a` = d(x, removed), `b` = d(x, kept)
inline void update(ClusterMethod m, float a, float &b, float merged, size_t sa, size_t sb, size_t sx) {
    const float fa = (float)sa, fb = (float)sb, fx = (float)sx;
    switch (m) {
        case LINK_SINGLE: if (a < b) b = a; break;
        case LINK_COMPLETE: if (a > b) b = a; break;
        case LINK_AVERAGE: b = (fa * a + fb * b) / (fa + fb); break;
        case LINK_WEIGHTED: b = 0.5f * (a + b); break;
        case LINK_WARD: {
            const float num = ((fx + fa) * a) + ((fx + fb) * b) - (fx * merged);
            b = num / (fa + fb + fx);
            break;
        }
        case LINK_CENTROID: {
            const float fab = fa + fb;
            b = (((fa * a) + (fb * b)) / fab) - ((fa * fb * merged) / (fab * fab));
            break;
        }
        case LINK_MEDIAN: b = (0.5f * (a + b)) - (merged * 0.25f); break;
    }
}

struct State {
    Cond dis;
    std::vector<uint8_t> alive;
    std::vector<size_t> size;
    State(float *d, size_t n) : dis{d, n}, alive(n, 1), size(n, 1) {}
    // cluster a disappears into cluster b
    void merge(ClusterMethod m, size_t a, size_t b, float merged) {
        alive[a] = 0;
        for (size_t x = 0; x < dis.n; ++x)
            if (alive[x] && x != b) update(m, dis.at(x, a), dis.at(x, b), merged, size[a], size[b], size[x]);
        size[b] += size[a];
    }
};

// stable sort by dissimilarity, then SciPy labels through a union-find
void relabel(std::vector<LinkStep> &steps, size_t n) {
    std::stable_sort(steps.begin(), steps.end(),
                     [](const LinkStep &x, const LinkStep &y) { return x.dissimilarity < y.dissimilarity; });
    std::vector<size_t> parent(2 * n);
    for (size_t i = 0; i < parent.size(); ++i) parent[i] = i;
    auto find = [&](size_t x) {
        while (parent[x] != x) x = parent[x];
        return x;
    };
    size_t next = n;
    for (auto &s : steps) {
        const size_t r1 = find(s.c1), r2 = find(s.c2);
        parent[r1] = parent[r2] = next++;
        s.c1 = std::min(r1, r2);
        s.c2 = std::max(r1, r2);
    }
}

// Method::Single: Prim's minimum spanning tree grown from observation 0
std::vector<LinkStep> mst(State &st) {
    const size_t n = st.dis.n;
    std::vector<LinkStep> steps;
    std::vector<float> best(n, std::numeric_limits<float>::infinity());
    size_t tip = 0;
    st.alive[0] = 0;
    for (size_t it = 0; it + 1 < n; ++it) {
        size_t pick = n;
        for (size_t x = 0; x < n; ++x) {
            if (!st.alive[x]) continue;
            best[x] = std::min(best[x], st.dis.at(x, tip));
            if (pick == n || best[x] < best[pick]) pick = x;
        }
        steps.push_back({pick, tip, best[pick]});
        st.alive[pick] = 0;
        tip = pick;
    }
    return steps;
}

// Complete / Average / Weighted / Ward: nearest-neighbour chain
std::vector<LinkStep> nn_chain(State &st, ClusterMethod m) {
    const size_t n = st.dis.n;
    std::vector<LinkStep> steps;
    std::vector<size_t> chain;
    for (size_t it = 0; it + 1 < n; ++it) {
        size_t a, b;
        float best;
        if (chain.size() < 4) {
            a = 0;
            while (!st.alive[a]) ++a;
            chain.assign(1, a);
            b = n;
            best = 0.f;
            for (size_t i = a + 1; i < n; ++i)
                if (st.alive[i] && (b == n || st.dis.at(a, i) < best)) {
                    best = st.dis.at(a, i);
                    b = i;
                }
        } else {
            chain.pop_back();
            chain.pop_back();
            b = chain.back();
            chain.pop_back();
            a = chain.back();
            best = st.dis.at(a, b);
        }
        do {
            chain.push_back(b);
            for (size_t x = 0; x < n; ++x)
                if (st.alive[x] && x != b && st.dis.at(x, b) < best) {
                    best = st.dis.at(x, b);
                    a = x;
                }
            b = a;
            a = chain.back();
        } while (b != chain[chain.size() - 2]);
        steps.push_back({a, b, best});
        st.merge(m, std::min(a, b), std::max(a, b), best);
    }
    return steps;
}

// Centroid / Median (not reducible, no chain): merge the closest pair of live clusters at every
// step.  rowmin[i] caches the nearest live j > i; a row is rescanned only when its cached partner
// died or one of its distances was rewritten.
std::vector<LinkStep> closest_pair(State &st, ClusterMethod m) {
    const size_t n = st.dis.n;
    std::vector<LinkStep> steps;
    std::vector<size_t> label(n), partner(n, n);
    std::vector<float> rowmin(n, 0.f);
    std::vector<uint8_t> stale(n, 1);
    for (size_t i = 0; i < n; ++i) label[i] = i;
    for (size_t it = 0; it + 1 < n; ++it) {
        size_t a = n;
        for (size_t i = 0; i < n; ++i) {
            if (!st.alive[i]) continue;
            if (stale[i]) {
                partner[i] = n;
                for (size_t j = i + 1; j < n; ++j)
                    if (st.alive[j] && (partner[i] == n || st.dis.at(i, j) < rowmin[i])) {
                        rowmin[i] = st.dis.at(i, j);
                        partner[i] = j;
                    }
                stale[i] = 0;
            }
            if (partner[i] != n && (a == n || rowmin[i] < rowmin[a])) a = i;
        }
        const size_t b = partner[a];
        const float d = rowmin[a];
        steps.push_back({std::min(label[a], label[b]), std::max(label[a], label[b]), d});
        st.merge(m, a, b, d);
        label[b] = n + it;
        // every distance to b changed, and rows that pointed at a lost their partner
        for (size_t i = 0; i < n; ++i)
            if (st.alive[i] && (i <= b || partner[i] == a)) stale[i] = 1;
    }
    return steps;
}

}  // namespace

std::vector<LinkStep> linkage(std::vector<float> &condensed, size_t n, ClusterMethod method) {
    if (condensed.size() != n * (n ? n - 1 : 0) / 2) throw std::runtime_error("linkage: condensed matrix has the wrong size");
    if (n < 2) return {};
    const bool squared = method == LINK_WARD || method == LINK_CENTROID || method == LINK_MEDIAN;
    if (squared)
        for (float &x : condensed) x = x * x;
    State st(condensed.data(), n);
    std::vector<LinkStep> steps;
    if (method == LINK_SINGLE) steps = mst(st);
    else if (method == LINK_CENTROID || method == LINK_MEDIAN) steps = closest_pair(st, method);
    else steps = nn_chain(st, method);
    if (method != LINK_CENTROID && method != LINK_MEDIAN) relabel(steps, n);
    if (squared)
        for (auto &s : steps) s.dissimilarity = std::sqrt(s.dissimilarity);
    return steps;
}

std::vector<size_t> similarity_order(const std::vector<float> &table, size_t n, ClusterMethod method) {
    if (n == 0) throw std::runtime_error("similarity: no groups (the reference's calculate_distances underflows, similarity.rs:248)");
    // calculate_distances + euclidean (similarity.rs:238-254)
    std::vector<float> condensed;
    condensed.reserve(n * (n - 1) / 2);
    for (size_t r = 0; r + 1 < n; ++r)
        for (size_t c = r + 1; c < n; ++c) {
            float sum = 0.0f;
            for (size_t k = 0; k < n; ++k) {
                const float d = table[r * n + k] - table[c * n + k];
                sum += d * d;  // powf(2.0) of an f32 = its correctly rounded square
            }
            condensed.push_back(std::sqrt(sum));
        }
    const std::vector<LinkStep> steps = linkage(condensed, n, method);
    // get_order_from_dendrogram (:205-217)
    std::vector<size_t> leaves;
    for (const auto &s : steps) {
        if (s.c1 < n) leaves.push_back(s.c1);
        if (s.c2 < n) leaves.push_back(s.c2);
    }
    // enumerate + sort_by_key(observation) + positions (:172-174)
    std::vector<size_t> indices(leaves.size());
    for (size_t k = 0; k < leaves.size(); ++k) indices[leaves[k]] = k;
    // sort_by_indices (:194-203), applied to a list that starts as 0..n
    std::vector<size_t> perm(n);
    for (size_t i = 0; i < n; ++i) perm[i] = i;
    for (size_t i = 0; i < indices.size(); ++i)
        while (i != indices[i]) {
            const size_t new_i = indices[i];
            std::swap(indices[i], indices[new_i]);
            std::swap(perm[i], perm[new_i]);
        }
    return perm;
}

}  // namespace pnh
