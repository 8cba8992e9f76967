#include "bvh.h"

#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cstring>
#include <functional>
#include <limits>
#include <optional>
#include <queue>
#include <stack>
#include <stdexcept>
#include <string>

namespace igh {

// ---------------------------------------------------------------- BVH2 (sweep SAH)

namespace {

struct SweepBuilder {
    const std::vector<BBox>& bboxes;
    const std::vector<V3>& centers;
    size_t min_leaf_size = 1;
    size_t max_leaf_size = 8;

    std::vector<size_t> prim_ids[3];
    std::vector<bool> marks;
    std::vector<float> accum;

    struct Split {
        size_t pos;
        float cost;
        size_t axis;
    };

    SweepBuilder(const std::vector<BBox>& b, const std::vector<V3>& c, size_t max_leaf)
        : bboxes(b)
        , centers(c)
        , max_leaf_size(max_leaf)
    {
        const size_t n = bboxes.size();
        marks.resize(n);
        accum.resize(n);
        for (int axis = 0; axis < 3; ++axis) {
            prim_ids[axis].resize(n);
            for (size_t i = 0; i < n; ++i)
                prim_ids[axis][i] = i;
            std::stable_sort(prim_ids[axis].begin(), prim_ids[axis].end(),
                             [&](size_t i, size_t j) { return centers[i][axis] < centers[j][axis]; });
        }
    }

    // SplitHeuristic with log_cluster_size 0, cost_ratio 1
    static float leafCost(size_t begin, size_t end, const BBox& bbox) { return bbox.halfArea() * (float)(end - begin); }
    static float nonSplitCost(size_t begin, size_t end, const BBox& bbox) { return bbox.halfArea() * ((float)(end - begin) - 1.0f); }

    BBox computeBBox(size_t begin, size_t end) const
    {
        BBox b;
        for (size_t i = begin; i < end; ++i)
            b.extend(bboxes[prim_ids[0][i]]);
        return b;
    }

    void findBestSplit(size_t axis, size_t begin, size_t end, Split& best)
    {
        size_t first_right = begin;

        // Sweep from the right to the left, computing the partial SAH cost
        BBox right_bbox;
        for (size_t i = end - 1; i > begin;) {
            constexpr size_t chunk_size = 32;
            const size_t next           = i - std::min(i - begin, chunk_size);
            float right_cost            = 0;
            for (; i > next; --i) {
                right_bbox.extend(bboxes[prim_ids[axis][i]]);
                accum[i] = right_cost = leafCost(i, end, right_bbox);
            }
            // Every chunk, check that we are not above the maximum cost
            if (right_cost > best.cost) {
                first_right = i;
                break;
            }
        }

        // Sweep from the left to the right, computing the full cost
        BBox left_bbox;
        for (size_t i = begin; i < first_right; ++i)
            left_bbox.extend(bboxes[prim_ids[axis][i]]);
        for (size_t i = first_right; i < end - 1; ++i) {
            left_bbox.extend(bboxes[prim_ids[axis][i]]);
            const float left_cost = leafCost(begin, i + 1, left_bbox);
            const float cost      = left_cost + accum[i + 1];
            if (cost < best.cost)
                best = Split{ i + 1, cost, axis };
            else if (left_cost > best.cost)
                break;
        }
    }

    std::optional<size_t> trySplit(const BBox& bbox, size_t begin, size_t end)
    {
        const float leaf_cost = nonSplitCost(begin, end, bbox);
        Split best{ (begin + end + 1) / 2, leaf_cost, 0 };
        for (size_t axis = 0; axis < 3; ++axis)
            findBestSplit(axis, begin, end, best);

        if (best.cost >= leaf_cost) {
            if (end - begin <= max_leaf_size)
                return std::nullopt;
            // Too many primitives for a leaf: median split on the largest axis
            best.pos   = (begin + end + 1) / 2;
            const V3 d = bbox.diameter();
            best.axis  = (d.x > d.y) ? (d.x > d.z ? 0 : 2) : (d.y > d.z ? 1 : 2);
        }

        // Partition, keeping the per-axis orders intact
        for (size_t i = begin; i < best.pos; ++i)
            marks[prim_ids[best.axis][i]] = true;
        for (size_t i = best.pos; i < end; ++i)
            marks[prim_ids[best.axis][i]] = false;
        for (size_t axis = 0; axis < 3; ++axis) {
            if (axis == best.axis)
                continue;
            std::stable_partition(prim_ids[axis].begin() + begin, prim_ids[axis].begin() + end,
                                  [&](size_t i) { return (bool)marks[i]; });
        }
        return best.pos;
    }
};

inline void setBounds(Bvh2Node& n, const BBox& b)
{
    n.bounds[0] = b.min.x;
    n.bounds[1] = b.max.x;
    n.bounds[2] = b.min.y;
    n.bounds[3] = b.max.y;
    n.bounds[4] = b.min.z;
    n.bounds[5] = b.max.z;
}

inline BBox getBounds(const Bvh2Node& n)
{
    BBox b;
    b.min = V3(n.bounds[0], n.bounds[2], n.bounds[4]);
    b.max = V3(n.bounds[1], n.bounds[3], n.bounds[5]);
    return b;
}

} // namespace

Bvh2 build_bvh2(const std::vector<BBox>& bboxes, const std::vector<V3>& centers, size_t max_leaf_size, size_t min_leaf_size)
{
    const size_t prim_count = bboxes.size();
    if (prim_count == 0)
        throw std::runtime_error("build_bvh2: no primitives");

    SweepBuilder builder(bboxes, centers, max_leaf_size);
    builder.min_leaf_size = std::max<size_t>(1, min_leaf_size);

    struct WorkItem {
        size_t node_id, begin, end;
        size_t size() const { return end - begin; }
    };

    Bvh2 bvh;
    bvh.nodes.reserve(2 * prim_count);
    bvh.nodes.emplace_back();
    setBounds(bvh.nodes.back(), builder.computeBBox(0, prim_count));

    std::stack<WorkItem> stack;
    stack.push(WorkItem{ 0, 0, prim_count });
    while (!stack.empty()) {
        const WorkItem item = stack.top();
        stack.pop();

        if (item.size() > builder.min_leaf_size) {
            const BBox node_bbox = getBounds(bvh.nodes[item.node_id]);
            if (auto split_pos = builder.trySplit(node_bbox, item.begin, item.end)) {
                const size_t first_child        = bvh.nodes.size();
                bvh.nodes[item.node_id].first      = (uint32_t)first_child;
                bvh.nodes[item.node_id].prim_count = 0;
                bvh.nodes.resize(first_child + 2);

                BBox first_bbox   = builder.computeBBox(item.begin, *split_pos);
                BBox second_bbox  = builder.computeBBox(*split_pos, item.end);
                auto first_range  = std::make_pair(item.begin, *split_pos);
                auto second_range = std::make_pair(*split_pos, item.end);

                // Largest-area child first (surface-area traversal order for any-hit queries)
                if (first_bbox.halfArea() < second_bbox.halfArea()) {
                    std::swap(first_bbox, second_bbox);
                    std::swap(first_range, second_range);
                }

                WorkItem first_item{ first_child + 0, first_range.first, first_range.second };
                WorkItem second_item{ first_child + 1, second_range.first, second_range.second };
                setBounds(bvh.nodes[first_child + 0], first_bbox);
                setBounds(bvh.nodes[first_child + 1], second_bbox);

                if (first_item.size() < second_item.size())
                    std::swap(first_item, second_item);
                stack.push(first_item);
                stack.push(second_item);
                continue;
            }
        }

        bvh.nodes[item.node_id].first      = (uint32_t)item.begin;
        bvh.nodes[item.node_id].prim_count = (uint32_t)item.size();
    }

    bvh.prim_ids = builder.prim_ids[0];
    return bvh;
}

// ---------------------------------------------------------------- reinsertion optimiser
//
// What bvh::v2::DefaultBuilder runs after the sweep at its default quality (High): Meister & Bittner, "Parallel Reinsertion for
// Bounding Volume Hierarchy Optimization" (Eurographics 2018), in the batched sequential form — per iteration the nodes with
// the largest surface area (a fixed share of the tree) each look for the position that shrinks the tree's summed area most
// when the node's subtree is cut out and hung in there, the candidates are applied in order of gain, and a move that touches a
// node an earlier move of the iteration touched is skipped. The search for one node is branch and bound: going up from the
// node's parent it keeps what removing the node has saved so far, and descends into each sibling subtree on the way only
// while that bound can still beat the best position found. Leaves keep their primitive ranges; only inner structure changes.
namespace {

struct Reinserter {
    std::vector<Bvh2Node>& nodes;
    std::vector<uint32_t> parent;

    explicit Reinserter(Bvh2& bvh)
        : nodes(bvh.nodes)
        , parent(bvh.nodes.size(), 0)
    {
        for (uint32_t i = 0; i < (uint32_t)nodes.size(); ++i)
            if (!nodes[i].isLeaf())
                parent[nodes[i].first] = parent[nodes[i].first + 1] = i;
    }

    // (the root is node 0 and every split appends its two children together: pairs start at odd indices)
    static uint32_t siblingOf(uint32_t i) { return (i & 1u) ? i + 1 : i - 1; }
    float areaOf(uint32_t i) const { return getBounds(nodes[i]).halfArea(); }

    struct Move {
        uint32_t from = 0, to = 0;
        float gain = 0;
    };

    mutable std::vector<std::pair<float, uint32_t>> open; // bestMove's work list (kept across calls: one allocation per pass, not per node)

    Move bestMove(uint32_t node) const
    {
        Move best;
        best.from              = node;
        const BBox node_box    = getBounds(nodes[node]);
        const float node_area  = node_box.halfArea();
        const uint32_t up      = parent[node];
        float saved            = areaOf(up); // the parent disappears: its sibling child takes its place
        uint32_t side          = siblingOf(node);
        BBox shrunk            = getBounds(nodes[side]); // box of the ancestor reached so far, without `node`
        uint32_t pivot         = up;
        open.clear();
        do {
            // positions inside the subtree that hangs beside the path at this height
            open.emplace_back(saved, side);
            while (!open.empty()) {
                const auto [bound, at] = open.back();
                open.pop_back();
                if (bound - node_area <= best.gain)
                    continue; // even a position whose box already holds `node` could not beat the best
                const Bvh2Node& dst = nodes[at];
                BBox merged         = getBounds(dst);
                merged.extend(node_box);
                const float here = bound - merged.halfArea(); // new parent of {dst, node} at dst's place
                if (here > best.gain) {
                    best.to   = at;
                    best.gain = here;
                }
                if (!dst.isLeaf()) {
                    const float below = here + getBounds(dst).halfArea(); // dst grows to `merged` instead of getting a new parent
                    open.emplace_back(below, dst.first);
                    open.emplace_back(below, dst.first + 1);
                }
            }
            // one level up: that ancestor shrinks to what is left below it
            if (pivot != up) {
                shrunk.extend(getBounds(nodes[side]));
                saved += areaOf(pivot) - shrunk.halfArea();
            }
            side  = siblingOf(pivot);
            pivot = parent[pivot];
        } while (pivot != 0);
        if (best.to == siblingOf(best.from) || best.to == parent[best.from])
            return Move{}; // the same tree
        return best;
    }

    void refitFrom(uint32_t i)
    {
        do {
            Bvh2Node& n = nodes[i];
            if (!n.isLeaf()) {
                BBox b = getBounds(nodes[n.first]);
                b.extend(getBounds(nodes[n.first + 1]));
                setBounds(n, b);
            }
            i = parent[i];
        } while (i != 0);
    }

    // cut `from` out (its sibling moves into the parent's slot) and hang it beside `to`: the freed pair of slots {from, sibling} holds
    // `from` and the old `to`, the slot of `to` becomes their parent
    void apply(const Move& m)
    {
        const uint32_t sib = siblingOf(m.from), up = parent[m.from];
        const Bvh2Node moved_sibling = nodes[sib], target = nodes[m.to];
        nodes[m.to].first      = std::min(m.from, sib);
        nodes[m.to].prim_count = 0;
        nodes[sib]             = target;
        nodes[up]              = moved_sibling;
        if (!target.isLeaf())
            parent[target.first] = parent[target.first + 1] = sib;
        if (!moved_sibling.isLeaf())
            parent[moved_sibling.first] = parent[moved_sibling.first + 1] = up;
        parent[sib]    = m.to;
        parent[m.from] = m.to;
        refitFrom(m.to);
        refitFrom(up);
    }

    void run(float batch_ratio, int iterations)
    {
        const size_t count = nodes.size();
        if (count < 4)
            return;
        const size_t batch = std::max<size_t>(1, (size_t)((float)count * batch_ratio));
        std::vector<uint32_t> order(count - 1);
        std::vector<float> area(count);
        std::vector<Move> moves;
        std::vector<char> touched(count);
        const bool debug = std::getenv("IGH_BVH_DEBUG") != nullptr;
        for (int it = 0; it < iterations; ++it) {
            // the `batch` nodes of largest area (never the root)
            for (uint32_t i = 0; i < (uint32_t)count; ++i)
                area[i] = areaOf(i);
            for (uint32_t i = 1; i < (uint32_t)count; ++i)
                order[i - 1] = i;
            const size_t take = std::min(batch, order.size());
            std::partial_sort(order.begin(), order.begin() + (ptrdiff_t)take, order.end(), [&](uint32_t a, uint32_t b) { return area[a] > area[b] || (area[a] == area[b] && a < b); });
            moves.clear();
            for (size_t k = 0; k < take; ++k) {
                const Move m = bestMove(order[k]);
                if (m.gain > 0)
                    moves.push_back(m);
            }
            std::stable_sort(moves.begin(), moves.end(), [](const Move& a, const Move& b) { return a.gain > b.gain; });
            std::fill(touched.begin(), touched.end(), 0);
            size_t applied = 0;
            double promised = 0;
            for (const Move& m : moves) {
                const uint32_t involved[5] = { m.to, m.from, siblingOf(m.from), parent[m.to], parent[m.from] };
                bool clash = false;
                for (uint32_t i : involved)
                    clash |= touched[i] != 0;
                if (clash)
                    continue;
                for (uint32_t i : involved)
                    touched[i] = 1;
                apply(m);
                ++applied;
                promised += m.gain;
            }
            if (debug) {
                double inner = 0;
                for (const Bvh2Node& n : nodes)
                    if (!n.isLeaf())
                        inner += getBounds(n).halfArea();
                std::fprintf(stderr, "reinsertion pass %d: %zu of %zu nodes looked at, %zu moves found, %zu applied (gain %.6g), inner area now %.6g\n", it, take, count, moves.size(), applied, promised, inner);
            }
        }
    }
};

} // namespace

float bvh2_sah_cost(const Bvh2& bvh)
{
    // SAH cost with unit node and primitive costs, relative to the root's area
    double cost = 0;
    for (const Bvh2Node& n : bvh.nodes)
        cost += (double)getBounds(n).halfArea() * (n.isLeaf() ? (double)n.prim_count : 1.0);
    return (float)(cost / (double)getBounds(bvh.nodes[0]).halfArea());
}

void optimize_bvh2(Bvh2& bvh)
{
    // defaults of bvh::v2::ReinsertionOptimizer::Config (batch_size_ratio 0.05, max_iter_count 3); IGH_BVH_REINSERT=0 switches the pass off,
    // IGH_BVH_REINSERT_ITERS / IGH_BVH_REINSERT_RATIO override (experiments, tools/bvh_visits.py). Load-time cost: per iteration a
    // partial sort of the nodes by area and a branch-and-bound search for 5 % of them (16 M triangles: a few seconds on top of the sweep).
    int iterations = 3;
    float ratio    = 0.05f;
    if (const char* e = std::getenv("IGH_BVH_REINSERT"))
        if (std::atoi(e) == 0)
            return;
    if (const char* e = std::getenv("IGH_BVH_REINSERT_ITERS"))
        iterations = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("IGH_BVH_REINSERT_RATIO"))
        ratio = std::min(1.0f, std::max(0.0f, (float)std::atof(e)));
    Reinserter(bvh).run(ratio, iterations);
    // the builder's convention again (largest-area child first: the order any-hit queries visit the children in), which the moves do not keep
    for (Bvh2Node& n : bvh.nodes)
        if (!n.isLeaf() && getBounds(bvh.nodes[n.first]).halfArea() < getBounds(bvh.nodes[n.first + 1]).halfArea())
            std::swap(bvh.nodes[n.first], bvh.nodes[n.first + 1]);
}

// ---------------------------------------------------------------- N-ary collapse

namespace {

constexpr size_t N = 8;

struct NNode {
    float bounds[6];
    int32_t primitive_or_child_count; // < 0: leaf with -count primitives
    uint32_t first_child_or_primitive;
    bool isLeaf() const { return primitive_or_child_count < 0; }
    uint32_t primitiveCount() const { return (uint32_t)-primitive_or_child_count; }
};

struct NBvh {
    std::vector<NNode> nodes;
    std::vector<size_t> primitive_indices;
};

inline NNode cloneNode(const Bvh2Node& o)
{
    NNode n;
    std::memcpy(n.bounds, o.bounds, sizeof(n.bounds));
    n.primitive_or_child_count = o.isLeaf() ? -(int32_t)o.prim_count : (int32_t)N;
    n.first_child_or_primitive = o.first;
    return n;
}

// Tuned collapse (default): open the inner child with the largest surface area until the node has N children or only
// leaves are left. The reference's breadth-first collapse below opens at most four nodes and spends openings on leaves,
// which leaves its "8-wide" nodes with 3.5 children on average (measured on Diamond.ply and on the stand-in terrain).
bool referenceCollapse()
{
    static const bool v = [] { const char* e = std::getenv("IGH_BVH_REFERENCE"); return e && *e && *e != '0'; }();
    return v;
}

void convertNode(const Bvh2& original, const Bvh2Node& node, NBvh& bvh, uint32_t cur_id);

void convertNodeGreedy(const Bvh2& original, const Bvh2Node& node, NBvh& bvh, uint32_t cur_id)
{
    if (node.isLeaf())
        return;
    std::vector<Bvh2Node> children{ original.nodes[node.first + 0], original.nodes[node.first + 1] };
    while (children.size() < N) {
        int best        = -1;
        float best_area = -1;
        for (size_t i = 0; i < children.size(); ++i) {
            if (children[i].isLeaf())
                continue;
            const float* b = children[i].bounds;
            const float dx = b[1] - b[0], dy = b[3] - b[2], dz = b[5] - b[4];
            const float area = dx * dy + dy * dz + dz * dx;
            if (area > best_area)
                best_area = area, best = (int)i;
        }
        if (best < 0)
            break;
        const Bvh2Node open = children[(size_t)best];
        children[(size_t)best] = original.nodes[open.first + 0]; // keeps the order of the remaining children
        children.insert(children.begin() + best + 1, original.nodes[open.first + 1]);
    }
    bvh.nodes[cur_id].primitive_or_child_count = (int32_t)children.size();
    bvh.nodes[cur_id].first_child_or_primitive = (uint32_t)bvh.nodes.size();
    for (const auto& child : children)
        bvh.nodes.push_back(cloneNode(child));
    const uint32_t first = bvh.nodes[cur_id].first_child_or_primitive;
    for (size_t i = 0; i < children.size(); ++i)
        convertNodeGreedy(original, children[i], bvh, first + (uint32_t)i);
}

void convertNode(const Bvh2& original, const Bvh2Node& node, NBvh& bvh, uint32_t cur_id)
{
    constexpr size_t MaxIter = (size_t)1 << (3 /*log2(8)*/ - 1);
    if (node.isLeaf())
        return;

    std::queue<Bvh2Node> queue;
    queue.push(original.nodes[node.first + 0]);
    queue.push(original.nodes[node.first + 1]);

    std::vector<Bvh2Node> children;
    for (size_t k = 0; k < MaxIter && !queue.empty(); ++k) {
        const Bvh2Node cur = queue.front();
        queue.pop();
        if (cur.isLeaf()) {
            children.push_back(cur);
        } else {
            queue.push(original.nodes[cur.first + 0]);
            queue.push(original.nodes[cur.first + 1]);
        }
    }
    while (!queue.empty()) {
        children.push_back(queue.front());
        queue.pop();
    }

    bvh.nodes[cur_id].primitive_or_child_count = (int32_t)children.size();
    bvh.nodes[cur_id].first_child_or_primitive = (uint32_t)bvh.nodes.size();
    for (const auto& child : children)
        bvh.nodes.push_back(cloneNode(child));

    const uint32_t first = bvh.nodes[cur_id].first_child_or_primitive;
    for (size_t i = 0; i < children.size(); ++i)
        convertNode(original, children[i], bvh, first + (uint32_t)i);
}

// Collapse that minimises the summed surface area of the wide inner nodes (the SAH cost of visiting them; the leaves are the binary
// tree's and cost the same under every collapse): the dynamic programme of Ylitie, Karras and Laine, "Efficient Incoherent Ray Traversal
// on GPUs Through Compressed Wide BVHs" (HPG 2017, section 3.1), without its leaf merging. T(n, i) = least cost of the subtree of n
// presented to a parent as at most i children; T(n, 1) = area(n) + D(n, 8) makes n a wide node, D(n, j) = min over k of T(left, k) +
// T(right, j - k) hands j slots to its two sides. The greedy collapse fills the nodes near the root and leaves the bottom of the tree as
// nodes of two leaves (3.3 children per node on the stand-in terrain); this one trades slots between the sides.
struct CollapsePlan {
    struct Entry {
        float t[N + 1];       // t[i], i = 1 .. N - 1 (t[N] unused)
        uint8_t split[N + 1]; // for j = 2 .. N: slots handed to the left side by D(n, j)
        uint8_t use[N + 1];   // for i = 1 .. N - 1: the number of slots T(n, i) really uses (1 = n is a wide node itself)
    };
    std::vector<Entry> e;

    explicit CollapsePlan(const Bvh2& bvh)
        : e(bvh.nodes.size())
    {
        // Bottom-up = a pre-order walk from the root, backwards. (build_bvh2 appends children behind their parent, but the reinsertion
        // pass re-links nodes — a moved pair can sit below its new parent in the array — so index order is not a topological order.)
        std::vector<uint32_t> order;
        order.reserve(bvh.nodes.size());
        {
            std::vector<uint32_t> open{ 0u };
            while (!open.empty()) {
                const uint32_t n = open.back();
                open.pop_back();
                order.push_back(n);
                if (!bvh.nodes[n].isLeaf())
                    open.push_back(bvh.nodes[n].first), open.push_back(bvh.nodes[n].first + 1);
            }
        }
        for (size_t at = order.size(); at-- > 0;) {
            const uint32_t n     = order[at];
            const Bvh2Node& node = bvh.nodes[n];
            Entry& en            = e[n];
            if (node.isLeaf()) {
                for (size_t i = 0; i <= N; ++i)
                    en.t[i] = 0, en.split[i] = 0, en.use[i] = 1;
                continue;
            }
            const Entry &l = e[node.first], &r = e[node.first + 1];
            float d[N + 1];
            for (size_t j = 2; j <= N; ++j) {
                d[j] = std::numeric_limits<float>::infinity();
                for (size_t k = 1; k < j; ++k) {
                    const size_t kl = std::min(k, N - 1), kr = std::min(j - k, N - 1);
                    const float c = l.t[kl] + r.t[kr];
                    if (c < d[j])
                        d[j] = c, en.split[j] = (uint8_t)k;
                }
            }
            const float* b = node.bounds;
            const float dx = b[1] - b[0], dy = b[3] - b[2], dz = b[5] - b[4];
            en.t[1]   = (dx * dy + dy * dz + dz * dx) + d[N];
            en.use[1] = 1;
            for (size_t i = 2; i < N; ++i) {
                if (d[i] < en.t[i - 1])
                    en.t[i] = d[i], en.use[i] = (uint8_t)i;
                else
                    en.t[i] = en.t[i - 1], en.use[i] = en.use[i - 1];
            }
            en.t[N] = en.t[N - 1], en.use[N] = en.use[N - 1];
        }
    }

    // the children node `n` contributes to its parent when it is given `slots` of them
    void gather(const Bvh2& bvh, uint32_t n, size_t slots, std::vector<uint32_t>& out) const
    {
        const Bvh2Node& node = bvh.nodes[n];
        const size_t used    = node.isLeaf() ? 1 : e[n].use[std::min(slots, N - 1)];
        if (used <= 1) {
            out.push_back(n);
            return;
        }
        const size_t k = e[n].split[used];
        gather(bvh, node.first, k, out);
        gather(bvh, node.first + 1, used - k, out);
    }
    void childrenOf(const Bvh2& bvh, uint32_t n, std::vector<uint32_t>& out) const
    {
        const Bvh2Node& node = bvh.nodes[n];
        const size_t k       = e[n].split[N];
        gather(bvh, node.first, k, out);
        gather(bvh, node.first + 1, N - k, out);
    }
};

void convertNodeOptimal(const Bvh2& original, const CollapsePlan& plan, uint32_t n, NBvh& bvh, uint32_t cur_id)
{
    if (original.nodes[n].isLeaf())
        return;
    std::vector<uint32_t> children;
    plan.childrenOf(original, n, children);
    bvh.nodes[cur_id].primitive_or_child_count = (int32_t)children.size();
    bvh.nodes[cur_id].first_child_or_primitive = (uint32_t)bvh.nodes.size();
    for (uint32_t c : children)
        bvh.nodes.push_back(cloneNode(original.nodes[c]));
    const uint32_t first = bvh.nodes[cur_id].first_child_or_primitive;
    for (size_t i = 0; i < children.size(); ++i)
        convertNodeOptimal(original, plan, children[i], bvh, first + (uint32_t)i);
}

// T(n, 1) by plain recursion over the tree as it is linked (no assumption about the order of the array): what CollapsePlan's table must hold
void recursiveCollapseCost(const Bvh2& bvh, uint32_t n, float t[N + 1])
{
    const Bvh2Node& node = bvh.nodes[n];
    if (node.isLeaf()) {
        for (size_t i = 0; i <= N; ++i)
            t[i] = 0;
        return;
    }
    float l[N + 1], r[N + 1], d[N + 1];
    recursiveCollapseCost(bvh, node.first, l);
    recursiveCollapseCost(bvh, node.first + 1, r);
    for (size_t j = 2; j <= N; ++j) {
        d[j] = std::numeric_limits<float>::infinity();
        for (size_t k = 1; k < j; ++k)
            d[j] = std::min(d[j], l[std::min(k, N - 1)] + r[std::min(j - k, N - 1)]);
    }
    const float* b = node.bounds;
    const float dx = b[1] - b[0], dy = b[3] - b[2], dz = b[5] - b[4];
    t[1] = (dx * dy + dy * dz + dz * dx) + d[N];
    for (size_t i = 2; i < N; ++i)
        t[i] = std::min(d[i], t[i - 1]);
    t[N] = t[N - 1];
}

bool greedyCollapse()
{
    static const bool v = [] { const char* e = std::getenv("IGH_COLLAPSE"); return e && std::string(e) == "greedy"; }();
    return v;
}

NBvh convertToNArity(const Bvh2& original)
{
    NBvh bvh;
    bvh.nodes.reserve((original.nodes.size() - 1) / (N / 2) + 1);
    bvh.nodes.push_back(cloneNode(original.nodes[0]));
    if (referenceCollapse())
        convertNode(original, original.nodes[0], bvh, 0);
    else if (greedyCollapse())
        convertNodeGreedy(original, original.nodes[0], bvh, 0);
    else
        convertNodeOptimal(original, CollapsePlan(original), 0, bvh, 0);
    bvh.primitive_indices = original.prim_ids;
    return bvh;
}

// write_node with a leaf callback (BvhNAdapter.h:37-93)
using LeafWriter = std::function<void(const NBvh&, const NNode&, size_t parent, size_t child)>;

void writeNode(std::vector<ig_node8>& nodes, const NBvh& bvh, const NNode& node, int parent, size_t child, const LeafWriter& writeLeaf)
{
    const size_t node_id = nodes.size();
    if (parent >= 0)
        nodes[(size_t)parent].child[child] = (int32_t)node_id + 1;
    nodes.emplace_back();
    std::memset(&nodes.back(), 0, sizeof(ig_node8));

    const size_t count = (size_t)node.primitive_or_child_count;
    if (count == 0 || count > N)
        throw std::runtime_error("BVH collapse produced an invalid node");

    // Which child goes into which slot. The traversal pushes the children it hits in slot order; an any-hit query makes each of them the
    // new top of the stack, i.e. looks at the LAST slot first, a closest-hit query keeps the nearest on top and the others in slot order.
    // IGH_CHILD_ORDER: asis (the collapse's order) | reverse | area_asc (largest box in the last slot: first for a shadow ray) | area_desc
    size_t order[N];
    for (size_t i = 0; i < count; ++i)
        order[i] = i;
    static const std::string child_order = [] { const char* e = std::getenv("IGH_CHILD_ORDER"); return std::string(e ? e : "asis"); }();
    if (child_order != "asis" && !referenceCollapse()) {
        auto area = [&](size_t i) {
            const float* b = bvh.nodes.at(node.first_child_or_primitive + i).bounds;
            const float dx = b[1] - b[0], dy = b[3] - b[2], dz = b[5] - b[4];
            return dx * dy + dy * dz + dz * dx;
        };
        if (child_order == "reverse")
            std::reverse(order, order + count);
        else if (child_order == "area_asc")
            std::stable_sort(order, order + count, [&](size_t a, size_t b) { return area(a) < area(b); });
        else if (child_order == "area_desc")
            std::stable_sort(order, order + count, [&](size_t a, size_t b) { return area(a) > area(b); });
    }
    for (size_t i = 0; i < count; ++i) {
        const NNode src = bvh.nodes.at(node.first_child_or_primitive + order[i]);
        for (int k = 0; k < 6; ++k)
            nodes[node_id].bounds[k][i] = src.bounds[k];
        if (src.isLeaf())
            writeLeaf(bvh, src, node_id, i);
        else
            writeNode(nodes, bvh, src, (int)node_id, i, writeLeaf);
    }

    const float inf = std::numeric_limits<float>::infinity();
    for (size_t i = count; i < N; ++i) {
        ig_node8& dst    = nodes[node_id];
        dst.bounds[0][i] = inf;
        dst.bounds[2][i] = inf;
        dst.bounds[4][i] = inf;
        dst.bounds[1][i] = -inf;
        dst.bounds[3][i] = -inf;
        dst.bounds[5][i] = -inf;
        dst.child[i]     = 0;
    }
}

void adapt(std::vector<ig_node8>& nodes, const Bvh2& bvh2, const LeafWriter& writeLeaf)
{
    const NBvh nbvh = convertToNArity(bvh2);
    if (nbvh.nodes[0].isLeaf()) {
        // Root is already a leaf: a node with one child (BvhNAdapter.h:26-31)
        NNode root                    = nbvh.nodes[0];
        root.primitive_or_child_count = 1;
        root.first_child_or_primitive = 0;
        writeNode(nodes, nbvh, root, -1, 0, writeLeaf);
    } else {
        writeNode(nodes, nbvh, nbvh.nodes[0], -1, 0, writeLeaf);
    }
}

struct TriangleProxy {
    V3 p0, e1, e2, n;
    TriangleProxy(V3 a, V3 b, V3 c)
        : p0(a)
        , e1(c - a)
        , e2(a - b)
    {
        // compute_stable_triangle_normal(e1, e2, p1 - p2), TriBVHAdapter.h:41-50
        const V3 A = e1, B = e2, C = b - c;
        const float ab_x = A[2] * B[1], ab_y = A[0] * B[2], ab_z = A[1] * B[0];
        const float bc_x = B[2] * C[1], bc_y = B[0] * C[2], bc_z = B[1] * C[0];
        const V3 cross_ab(A[1] * B[2] - ab_x, A[2] * B[0] - ab_y, A[0] * B[1] - ab_z);
        const V3 cross_bc(B[1] * C[2] - bc_x, B[2] * C[0] - bc_y, B[0] * C[1] - bc_z);
        n = V3(std::abs(ab_x) < std::abs(bc_x) ? cross_ab[0] : cross_bc[0],
               std::abs(ab_y) < std::abs(bc_y) ? cross_ab[1] : cross_bc[1],
               std::abs(ab_z) < std::abs(bc_z) ? cross_ab[2] : cross_bc[2]);
    }
    V3 p1() const { return p0 - e2; }
    V3 p2() const { return p0 + e1; }
    BBox bbox() const
    {
        BBox b;
        b.extend(p0);
        b.extend(p1());
        b.extend(p2());
        return b;
    }
    V3 center() const { return (p0 + p1() + p2()) * (1.0f / 3); }
};

} // namespace

void build_tri_bvh8(const TriMesh& mesh, std::vector<ig_node8>& nodes, std::vector<ig_tri4>& tris)
{
    constexpr size_t M = 4;
    std::vector<TriangleProxy> triangles;
    triangles.reserve(mesh.faceCount());
    for (size_t f = 0; f < mesh.faceCount(); ++f)
        triangles.emplace_back(mesh.vertices.at(mesh.indices[f * 4 + 0]), mesh.vertices.at(mesh.indices[f * 4 + 1]), mesh.vertices.at(mesh.indices[f * 4 + 2]));

    std::vector<BBox> bboxes;
    std::vector<V3> centers;
    for (const auto& t : triangles) {
        bboxes.push_back(t.bbox());
        centers.push_back(t.center());
    }

    // Tuned build (default): a leaf is at least one full Tri4 packet — four triangles are not split further, since
    // a packet costs one fetch whether it holds two triangles or four. The library defaults the reference ends up
    // with (min_leaf_size 1) give leaves of 2.0 - 2.3 triangles.
    size_t min_leaf = M;
    if (const char* e = std::getenv("IGH_MIN_LEAF"))
        min_leaf = (size_t)std::max(1, std::atoi(e)); // experiments
    size_t max_leaf = 8;
    if (const char* e = std::getenv("IGH_MAX_LEAF"))
        max_leaf = (size_t)std::max(1, std::atoi(e)); // experiments
    Bvh2 bvh2 = build_bvh2(bboxes, centers, std::max<size_t>(max_leaf, min_leaf), referenceCollapse() ? 1 : min_leaf);
    optimize_bvh2(bvh2);

    adapt(nodes, bvh2, [&](const NBvh& bvh, const NNode& node, size_t parent, size_t child) {
        nodes[parent].child[child] = ~(int32_t)tris.size();
        const size_t ref_count     = node.primitiveCount();
        for (size_t i = 0; i < ref_count; i += M) {
            const size_t c = i + M <= ref_count ? M : ref_count - i;
            ig_tri4 tri;
            std::memset(&tri, 0, sizeof(tri));
            for (size_t j = 0; j < c; ++j) {
                const int id            = (int)bvh.primitive_indices.at(node.first_child_or_primitive + i + j);
                const TriangleProxy& in = triangles[(size_t)id];
                for (int k = 0; k < 3; ++k) {
                    tri.v0[k][j] = in.p0[k];
                    tri.e1[k][j] = in.e1[k];
                    tri.e2[k][j] = in.e2[k];
                    tri.n[k][j]  = in.n[k];
                }
                tri.prim_id[j] = id;
            }
            for (size_t j = c; j < M; ++j)
                tri.prim_id[j] = (int32_t)0xFFFFFFFF;
            tris.push_back(tri);
        }
        tris.back().prim_id[M - 1] |= (int32_t)0x80000000;
    });
}

void build_scene_bvh8(const std::vector<EntityObject>& objs, std::vector<ig_node8>& nodes, std::vector<ig_entity_leaf1>& leaves)
{
    std::vector<BBox> bboxes;
    std::vector<V3> centers;
    for (const auto& o : objs) {
        bboxes.push_back(o.bbox);
        centers.push_back(o.bbox.center());
    }

    // Leaves of the scene BVH. The sweep builder stops splitting where the SAH sees no gain, which for the boxes of a room's walls (they
    // all span the room) is at once: a leaf of several entities whose boxes are then scanned in storage order by every ray that reaches
    // it. With one entity per leaf the wide node above them orders the entities by entry distance and the cull against the current hit
    // works per entity; up to two per leaf where the SAH keeps them together (diamond_scene: 1 / 2 / 3 / 4 / 8 entities per leaf = 7 740 / 7 845 /
    // 7 840 / 7 270 / 7 240 Mrays/s, profiles/r03_experiment_ab.txt). IGH_SCENE_MAX_LEAF overrides (8 = the library default the reference ends up with).
    size_t max_leaf = referenceCollapse() ? 8 : 2;
    if (const char* e = std::getenv("IGH_SCENE_MAX_LEAF"))
        max_leaf = (size_t)std::max(1, std::atoi(e));
    Bvh2 bvh2 = build_bvh2(bboxes, centers, max_leaf);
    optimize_bvh2(bvh2);

    adapt(nodes, bvh2, [&](const NBvh& bvh, const NNode& node, size_t parent, size_t child) {
        nodes[parent].child[child] = ~(int32_t)leaves.size();
        for (size_t i = 0; i < node.primitiveCount(); ++i) {
            const int id           = (int)bvh.primitive_indices.at(node.first_child_or_primitive + i);
            const EntityObject& in = objs[(size_t)id];
            ig_entity_leaf1 l;
            l.min[0] = in.bbox.min.x, l.min[1] = in.bbox.min.y, l.min[2] = in.bbox.min.z;
            l.max[0] = in.bbox.max.x, l.max[1] = in.bbox.max.y, l.max[2] = in.bbox.max.z;
            l.entity_id = in.entity_id;
            l.shape_id  = in.shape_id;
            std::memcpy(l.local, in.local, sizeof(l.local));
            l.flags   = in.flags;
            l.mat_id  = in.material_id;
            l.user[0] = in.user1;
            l.user[1] = in.user2;
            leaves.push_back(l);
        }
        leaves.back().entity_id |= (int32_t)0x80000000;
    });
}

// Child boxes on a per-node 8-bit grid (Ylitie, Karras, Laine 2017, section 3.2: origin = the node's lower corner, one power-of-two
// scale per axis, lower planes rounded down and upper planes up), written back as the floats they decode to: plane =
// fmaf(q, 2^e, origin). The tables stay ordinary Node8 records — any traversal (the oracle's included) reads slightly larger child
// boxes — and the HIP device can hold such a node in 128 bytes without loss (igd_assign_scene, DevScene::node_format). The grid is
// left in the record's padding words: pad[0..2] = origin, pad[3] = the three biased exponents | IG_NODE8_QUANT_MARK.
void quantise_node8(ig_node8* nodes, size_t count)
{
    for (size_t n = 0; n < count; ++n) {
        ig_node8& nd = nodes[n];
        // Everything into temporaries first: a node is rewritten on all three axes or not at all. A node whose planes do not fit a grid —
        // only one with a non-finite plane can fail to: the extent of finite floats is below 255 x 2^122 — keeps its floats and gets no
        // mark (clamping the byte of an upper plane would write a box that no longer contains the child's); the device then
        // keeps such a scene on Node8 records (packQuantisedNodes: lossless or not at all).
        float planes[6][8];
        float origins[3];
        uint32_t exps = 0;
        bool node_fits = true;
        for (int a = 0; a < 3 && node_fits; ++a) {
            float origin = std::numeric_limits<float>::infinity(), top = -std::numeric_limits<float>::infinity();
            bool finite  = true;
            for (int i = 0; i < 8; ++i)
                if (nd.child[i] != 0) {
                    finite = finite && std::isfinite(nd.bounds[2 * a][i]) && std::isfinite(nd.bounds[2 * a + 1][i]);
                    origin = std::min(origin, nd.bounds[2 * a][i]), top = std::max(top, nd.bounds[2 * a + 1][i]);
                }
            if (!finite) {
                node_fits = false;
                break;
            }
            if (!(origin <= top)) // (no children: never written by the builder)
                origin = top = 0;
            const double extent = (double)top - (double)origin;
            int e               = extent > 0 ? (int)std::ceil(std::log2(extent / 255.0)) : -126;
            e                   = std::max(-126, std::min(127, e));
            uint8_t qlo[8], qhi[8];
            bool fits = false;
            for (; e <= 127; ++e) {
                const float s = std::ldexp(1.0f, e);
                fits          = true;
                for (int i = 0; i < 8 && fits; ++i) {
                    qlo[i] = qhi[i] = 0;
                    if (nd.child[i] == 0)
                        continue;
                    const float lo = nd.bounds[2 * a][i], hi = nd.bounds[2 * a + 1][i];
                    double ql = std::floor(((double)lo - (double)origin) / (double)s), qh = std::ceil(((double)hi - (double)origin) / (double)s);
                    ql = std::min(255.0, std::max(0.0, ql)), qh = std::max(0.0, qh);
                    while (ql > 0 && std::fmaf((float)ql, s, origin) > lo)
                        ql -= 1;
                    while (qh <= 255 && std::fmaf((float)qh, s, origin) < hi)
                        qh += 1;
                    // (the decoded planes must be finite and on the right side: fmaf can overflow at the top of the float range)
                    fits   = qh <= 255 && std::fmaf((float)ql, s, origin) <= lo && std::isfinite(std::fmaf((float)std::min(255.0, qh), s, origin));
                    qlo[i] = (uint8_t)ql, qhi[i] = (uint8_t)std::min(255.0, qh);
                }
                if (fits)
                    break;
            }
            if (!fits) {
                node_fits = false;
                break;
            }
            const float s = std::ldexp(1.0f, e);
            for (int i = 0; i < 8; ++i) {
                planes[2 * a][i]     = nd.child[i] != 0 ? std::fmaf((float)qlo[i], s, origin) : nd.bounds[2 * a][i];
                planes[2 * a + 1][i] = nd.child[i] != 0 ? std::fmaf((float)qhi[i], s, origin) : nd.bounds[2 * a + 1][i];
            }
            origins[a] = origin;
            exps |= (uint32_t)(e + 127) << (8 * a);
        }
        if (!node_fits) {
            nd.pad[0] = nd.pad[1] = nd.pad[2] = nd.pad[3] = 0;
            continue;
        }
        std::memcpy(nd.bounds, planes, sizeof(planes));
        std::memcpy(&nd.pad[0], origins, sizeof(origins));
        nd.pad[3] = (int32_t)(exps | IG_NODE8_QUANT_MARK);
    }
}

// Diagnostics (tests/test_bvh_builder.py through igh_test_collapse_plan): sweep build over the boxes, the reinsertion pass with the given
// parameters, then out[0] = the collapse plan's cost of the root as a wide node, out[1] = the same by plain recursion, out[2] = inner
// nodes with a child below them in the array (what the reinsertion pass leaves behind), out[3] = inner nodes whose box does not
// contain a child's, out[4] = summed area of the wide nodes convertToNArity really emits.
void collapse_plan_check(const std::vector<BBox>& boxes, float ratio, int iterations, double out[5])
{
    std::vector<V3> centers(boxes.size());
    for (size_t i = 0; i < boxes.size(); ++i)
        centers[i] = (boxes[i].min + boxes[i].max) * 0.5f;
    Bvh2 bvh = build_bvh2(boxes, centers, 8, 1);
    Reinserter(bvh).run(ratio, iterations);
    for (Bvh2Node& n : bvh.nodes)
        if (!n.isLeaf() && getBounds(bvh.nodes[n.first]).halfArea() < getBounds(bvh.nodes[n.first + 1]).halfArea())
            std::swap(bvh.nodes[n.first], bvh.nodes[n.first + 1]);
    const CollapsePlan plan(bvh);
    float t[N + 1];
    recursiveCollapseCost(bvh, 0, t);
    out[0] = bvh.nodes[0].isLeaf() ? 0.0 : (double)plan.e[0].t[1];
    out[1] = bvh.nodes[0].isLeaf() ? 0.0 : (double)t[1];
    out[2] = out[3] = out[4] = 0;
    for (uint32_t i = 0; i < (uint32_t)bvh.nodes.size(); ++i) {
        const Bvh2Node& n = bvh.nodes[i];
        if (n.isLeaf())
            continue;
        out[2] += n.first < i ? 1 : 0;
        const BBox b = getBounds(n);
        for (uint32_t c = n.first; c < n.first + 2; ++c) {
            const BBox cb = getBounds(bvh.nodes[c]);
            const bool inside = b.min.x <= cb.min.x && b.min.y <= cb.min.y && b.min.z <= cb.min.z && b.max.x >= cb.max.x && b.max.y >= cb.max.y && b.max.z >= cb.max.z;
            out[3] += inside ? 0 : 1;
        }
    }
    const NBvh wide = convertToNArity(bvh);
    for (const NNode& n : wide.nodes)
        if (!n.isLeaf()) {
            const float dx = n.bounds[1] - n.bounds[0], dy = n.bounds[3] - n.bounds[2], dz = n.bounds[5] - n.bounds[4];
            out[4] += (double)(dx * dy + dy * dz + dz * dx);
        }
}

} // namespace igh
