// scene_host.cpp -- Object/Scene/Camera constructors and the BVH builder + flattener.
//
// The builder takes the same decisions as the reference's BvhTree::build_sah /
// build_midpoint (rayrs-lib/src/bvh.rs:227-389) -- same longest-axis rule, same
// stable sort by bbox centre, same 1..=splits candidate planes, same strict "<"
// on the SAH cost, same median fallback, same leaf threshold of 4, same child
// order -- but evaluates all candidate planes of a node from one prefix and one
// suffix sweep of surface areas instead of 2*splits from_object_list folds.
// min/max are exact, so the swept boxes are bit-identical to the folded ones and
// the chosen split is the same (tests/test_bvh_builder.py checks the exported
// tree against the oracle's literal restatement).
#include "scene_host.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "../../include/rayrs_numeric.h"

namespace rayrs {

// ---------------------------------------------------------------- surfaces

static bool in01(const double* c) {
    return c[0] >= 0.0 && c[0] <= 1.0 && c[1] >= 0.0 && c[1] <= 1.0 && c[2] >= 0.0 && c[2] <= 1.0;
}

// ctor asserts: material.rs:609, :630, :651-653, :674-676, :706-708, :833-837,
// :864-868, :888-893, :1068-1072
int ObjectList::add_surface(const rayrs_material* m, const rayrs_emission* e) {
    if (!m) return RAYRS_INVALID_ARG;
    SurfaceDev s;
    std::memset(&s, 0, sizeof(s));
    s.kind = m->kind;
    if (m->kind < 0 || m->kind > RAYRS_MAT_NO_REFLECT) return RAYRS_INVALID_ARG;
    if (m->kind != RAYRS_MAT_NO_REFLECT) {
        if (!in01(m->color)) return RAYRS_INVALID_ARG;
        for (int i = 0; i < 3; i++) s.color[i] = m->color[i];
        const bool needs_ior = m->kind == RAYRS_MAT_REFRACT || m->kind == RAYRS_MAT_GLASS ||
                               m->kind == RAYRS_MAT_COOK_TORRANCE_REFRACT ||
                               m->kind == RAYRS_MAT_COOK_TORRANCE_GLASS || m->kind == RAYRS_MAT_PLASTIC;
        if (needs_ior && !(m->ior > 0.0 && std::isfinite(m->ior))) return RAYRS_INVALID_ARG;
        s.ior = m->ior;
        if (m->kind >= RAYRS_MAT_COOK_TORRANCE && m->kind <= RAYRS_MAT_PLASTIC) {
            if (!(m->alpha > 0.0 && std::isfinite(m->alpha))) return RAYRS_INVALID_ARG;
            s.ct_alpha2 = m->alpha * m->alpha;
            for (int i = 0; i < 3; i++) s.ct_color[i] = m->color[i];
            if (m->kind == RAYRS_MAT_COOK_TORRANCE) {
                s.metallic = m->metallic ? 1 : 0;
                if (s.metallic) {
                    for (int i = 0; i < 3; i++) s.ct_r0[i] = m->r0[i];
                } else {
                    s.ct_ior = m->ior;
                }
            } else {
                s.metallic = 0;
                s.ct_ior = m->ior;
                if (m->kind == RAYRS_MAT_PLASTIC) {
                    if (!in01(m->spec_color)) return RAYRS_INVALID_ARG;
                    for (int i = 0; i < 3; i++) s.ct_color[i] = m->spec_color[i];
                }
            }
        }
    }
    if (e && e->emissive) {
        if (!(e->strength >= 0.0) || !in01(e->color)) return RAYRS_INVALID_ARG;
        s.emissive = 1;
        for (int i = 0; i < 3; i++) s.emit[i] = e->strength * e->color[i];  // material.rs:1080
    }
    // share identical rows (Object::from_triangles clones one pair per triangle)
    for (size_t i = 0; i < surfaces.size(); i++)
        if (std::memcmp(&surfaces[i], &s, sizeof(s)) == 0) return (int)i;
    surfaces.push_back(s);
    return (int)surfaces.size() - 1;
}

// ------------------------------------------------------------------- boxes

// From<&Sphere/&Plane/&Triangle> for AxisAlignedBoundingBox, geometry.rs:686-733
Aabb shape_bbox(const Shape& s) {
    Aabb b;
    if (s.kind == PRIM_SPHERE) {
        const double radius = rr_sqrt(s.radius2);
        b.xmin = s.origin.x - radius;
        b.xmax = s.origin.x + radius;
        b.ymin = s.origin.y - radius;
        b.ymax = s.origin.y + radius;
        b.zmin = s.origin.z - radius;
        b.zmax = s.origin.z + radius;
    } else if (s.kind == PRIM_PLANE) {
        if (s.axis == RAYRS_AXIS_X || s.axis == RAYRS_AXIS_XREV) {
            b = {s.pos, s.pos, s.u0, s.u1, s.v0, s.v1};
        } else if (s.axis == RAYRS_AXIS_Y || s.axis == RAYRS_AXIS_YREV) {
            b = {s.u0, s.u1, s.pos, s.pos, s.v0, s.v1};
        } else {
            b = {s.u0, s.u1, s.v0, s.v1, s.pos, s.pos};
        }
    } else {
        b.xmin = rr_min(s.p1.x, rr_min(s.p2.x, s.p3.x));
        b.ymin = rr_min(s.p1.y, rr_min(s.p2.y, s.p3.y));
        b.zmin = rr_min(s.p1.z, rr_min(s.p2.z, s.p3.z));
        b.xmax = rr_max(s.p1.x, rr_max(s.p2.x, s.p3.x));
        b.ymax = rr_max(s.p1.y, rr_max(s.p2.y, s.p3.y));
        b.zmax = rr_max(s.p1.z, rr_max(s.p2.z, s.p3.z));
    }
    return b;
}

static inline Aabb merge(const Aabb& a, const Aabb& o) {  // expand, geometry.rs:674-683
    return {rr_min(a.xmin, o.xmin), rr_max(a.xmax, o.xmax), rr_min(a.ymin, o.ymin),
            rr_max(a.ymax, o.ymax), rr_min(a.zmin, o.zmin), rr_max(a.zmax, o.zmax)};
}
static inline double area(const Aabb& b) {  // surface_area, geometry.rs:640-645
    const double x = b.xmax - b.xmin, y = b.ymax - b.ymin, z = b.zmax - b.zmin;
    return 2. * x * y + 2. * y * z + 2. * x * z;
}
static inline double centre_of(double lo, double hi) { return (hi - lo) / 2. + lo; }  // geometry.rs:577-582

// ----------------------------------------------------------------- builder

namespace {

struct Builder {
    const ObjectList& list;
    FlatScene& flat;
    const bool sah;
    const uint32_t splits;
    std::vector<Aabb> boxes;
    std::vector<double> centre[3];
    std::vector<uint32_t> order;  // objects in their current (sorted) order
    std::vector<std::pair<double, uint32_t>> keyed;
    std::vector<double> area_left, area_right;
    uint32_t next_prim = 0;

    Builder(const ObjectList& l, FlatScene& f, bool use_sah, uint32_t n_splits)
        : list(l), flat(f), sah(use_sah), splits(n_splits) {
        const size_t n = l.objs.size();
        boxes.resize(n);
        for (auto& c : centre) c.resize(n);
        order.resize(n);
        keyed.resize(n);
        area_left.resize(n);
        area_right.resize(n);
        for (size_t i = 0; i < n; i++) {
            boxes[i] = shape_bbox(l.objs[i].geom);
            centre[0][i] = centre_of(boxes[i].xmin, boxes[i].xmax);
            centre[1][i] = centre_of(boxes[i].ymin, boxes[i].ymax);
            centre[2][i] = centre_of(boxes[i].zmin, boxes[i].zmax);
            order[i] = (uint32_t)i;
        }
        flat.prim_object.assign(n, 0);
    }

    Aabb bounds(size_t lo, size_t hi) const {
        Aabb b = boxes[order[lo]];
        for (size_t i = lo + 1; i < hi; i++) b = merge(b, boxes[order[i]]);
        return b;
    }

    // BvhData::sort: a stable sort of the node's objects by centre on `axis`
    void sort_range(size_t lo, size_t hi, int axis) {
        const std::vector<double>& c = centre[axis];
        bool sorted = true;
        for (size_t i = lo + 1; i < hi; i++) {
            if (c[order[i - 1]] > c[order[i]]) {
                sorted = false;
                break;
            }
        }
        if (sorted) return;
        for (size_t i = lo; i < hi; i++) keyed[i] = {c[order[i]], order[i]};
        std::stable_sort(keyed.begin() + (std::ptrdiff_t)lo, keyed.begin() + (std::ptrdiff_t)hi,
                         [](const std::pair<double, uint32_t>& a, const std::pair<double, uint32_t>& b) {
                             return a.first < b.first;
                         });
        for (size_t i = lo; i < hi; i++) order[i] = keyed[i].second;
    }

    uint32_t emit_single(uint32_t obj) {
        const uint32_t p = next_prim++;
        flat.prim_object[p] = obj;
        return (REF_SINGLE << 30) | (p << 2);
    }

    // Node over order[lo, hi), hi - lo >= 1 handled by the caller for singles.
    // Returns the child reference and the node's box.
    uint32_t build(size_t lo, size_t hi, uint32_t level, Aabb* box_out) {
        const size_t len = hi - lo;
        const Aabb box = bounds(lo, hi);
        *box_out = box;
        if (len <= 4) {  // bvh.rs:306-316 / :378-387
            const uint32_t first = next_prim;
            for (size_t i = lo; i < hi; i++) flat.prim_object[next_prim++] = order[i];
            return (REF_RANGE << 30) | (first << 2) | (uint32_t)(len - 1);
        }
        const double ex = box.xmax - box.xmin, ey = box.ymax - box.ymin, ez = box.zmax - box.zmin;
        int axis;
        double lo_edge, extent;
        if (ex >= ey && ex >= ez) {  // bvh.rs:248-257
            axis = 0, lo_edge = box.xmin, extent = ex;
        } else if (ey >= ez) {
            axis = 1, lo_edge = box.ymin, extent = ey;
        } else {
            axis = 2, lo_edge = box.zmin, extent = ez;
        }
        sort_range(lo, hi, axis);
        const std::vector<double>& c = centre[axis];

        bool found = false;
        size_t ind = 0;
        if (sah) {
            const double total_area = area(box);
            // area_left[k]  = SA(objects lo .. lo+k-1), k >= 1
            // area_right[k] = SA(objects lo+k .. hi-1)
            Aabb acc = boxes[order[lo]];
            for (size_t k = 1; k < len; k++) {
                area_left[lo + k] = area(acc);
                acc = merge(acc, boxes[order[lo + k]]);
            }
            acc = boxes[order[hi - 1]];
            area_right[lo + len - 1] = area(acc);
            for (size_t k = len - 1; k-- > 0;) {
                acc = merge(boxes[order[lo + k]], acc);
                area_right[lo + k] = area(acc);
            }
            const double step = extent / (double)(splits - 1u);  // bvh.rs:259
            double best = std::numeric_limits<double>::infinity();
            size_t k = 0;
            for (uint32_t i = 1; i < splits + 1u; i++) {  // bvh.rs:262
                const double plane = lo_edge + (double)i * step;
                while (k < len && !(c[order[lo + k]] > plane)) k++;  // split_ind, bvh.rs:7-13
                if (k == len) break;                                 // None for this and all later planes
                const double p_left = k > 0 ? area_left[lo + k] / total_area : 0.;
                const double p_right = area_right[lo + k] / total_area;
                const double cost = 0.3 + 1. * (p_left * (double)k + p_right * (double)(len - k));  // bvh.rs:36-37
                if (cost < best) {
                    best = cost;
                    ind = k;
                    found = true;
                }
            }
        } else {
            const double plane = axis == 0 ? centre_of(box.xmin, box.xmax)
                                           : (axis == 1 ? centre_of(box.ymin, box.ymax) : centre_of(box.zmin, box.zmax));
            size_t k = 0;
            while (k < len && !(c[order[lo + k]] > plane)) k++;
            found = k < len;
            ind = k;
        }
        if (!found || ind == 0 || ind == len - 1) ind = len / 2;  // bvh.rs:279-287

        const uint32_t rec = (uint32_t)(flat.child_ref.size() / 2);
        flat.child_ref.resize(flat.child_ref.size() + 2);
        flat.child_box.resize(flat.child_box.size() + 12);
        if (level + 1 > flat.depth) flat.depth = level + 1;

        const size_t mid = lo + ind;
        const size_t range[2][2] = {{lo, mid}, {mid, hi}};
        for (int ch = 0; ch < 2; ch++) {
            Aabb cb;
            uint32_t ref;
            if (range[ch][1] - range[ch][0] > 1) {  // bvh.rs:294-303
                ref = build(range[ch][0], range[ch][1], level + 1, &cb);
            } else {
                const uint32_t obj = order[range[ch][0]];
                cb = boxes[obj];
                ref = emit_single(obj);
            }
            flat.child_ref[(size_t)rec * 2 + ch] = ref;
            double* bx = &flat.child_box[((size_t)rec * 2 + ch) * 6];
            bx[0] = cb.xmin, bx[1] = cb.xmax, bx[2] = cb.ymin, bx[3] = cb.ymax, bx[4] = cb.zmin, bx[5] = cb.zmax;
        }
        return (REF_INTERIOR << 30) | rec;
    }
};

inline bool f32_exact(double v) { return (double)(float)v == v; }

// Surface area of the box around a wide record's tested slots.
double record_area(const WalkTree& f, uint32_t rec) {
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    bool any = false;
    for (int i = 0; i < 4; i++) {
        const uint32_t kind = f.ref[(size_t)rec * 4 + i] >> 30;
        if (kind != REF_INTERIOR && kind != REF_RANGE) continue;
        const double* b = &f.box[((size_t)rec * 4 + i) * 6];
        for (int a = 0; a < 3; a++) {
            if (!any || b[2 * a] < lo[a]) lo[a] = b[2 * a];
            if (!any || b[2 * a + 1] > hi[a]) hi[a] = b[2 * a + 1];
        }
        any = true;
    }
    if (!any) return 0.0;
    const double ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    return 2.0 * (ex * ey + ey * ez + ex * ez);
}

// Renumbers the wide records so that the WIDE_FRONT records with the largest boxes come
// first, largest first (ties: lower index first); the others keep their depth-first order
// behind them.  A ray meets a box with probability proportional to its surface area, so
// these are the records most queries read, and the traversal kernel keeps the first of them
// in LDS (wavefront.hip) instead of asking the vector L1 for them.
void front_largest(WalkTree& f) {
    const uint32_t n = f.n();
    const uint32_t k = n < WIDE_FRONT ? n : WIDE_FRONT;
    if (k == 0) return;
    std::vector<double> area(n);
    for (uint32_t r = 0; r < n; r++) area[r] = record_area(f, r);
    std::vector<uint32_t> order(n);
    for (uint32_t r = 0; r < n; r++) order[r] = r;
    auto larger = [&](uint32_t a, uint32_t b) { return area[a] > area[b] || (area[a] == area[b] && a < b); };
    std::partial_sort(order.begin(), order.begin() + k, order.end(), larger);
    std::vector<uint8_t> in_front(n, 0);
    for (uint32_t i = 0; i < k; i++) in_front[order[i]] = 1;
    uint32_t at = k;
    for (uint32_t r = 0; r < n; r++)
        if (!in_front[r]) order[at++] = r;  // order[new] = old
    std::vector<uint32_t> new_of(n);
    for (uint32_t i = 0; i < n; i++) new_of[order[i]] = i;
    std::vector<double> box(f.box.size());
    std::vector<uint32_t> ref(f.ref.size());
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t old = order[i];
        for (int c = 0; c < 4; c++) {
            uint32_t r = f.ref[(size_t)old * 4 + c];
            if ((r >> 30) == REF_INTERIOR) r = (REF_INTERIOR << 30) | new_of[r & 0x3fffffffu];
            ref[(size_t)i * 4 + c] = r;
        }
        for (int c = 0; c < 24; c++) box[(size_t)i * 24 + c] = f.box[(size_t)old * 24 + c];
    }
    f.box.swap(box);
    f.ref.swap(ref);
    f.root_ref = (REF_INTERIOR << 30) | new_of[f.root_ref & 0x3fffffffu];
}

// ------------------------------------------------------------ the walk trees
//
// What decides the reference's answer is, per primitive, ONE box: a primitive is reached by
// BvhTree::intersect (bvh.rs:391-415) iff every box on its root path passes the slab test,
// those boxes nest exactly (a Node's box is the min/max of everything below it) and every
// operation of the slab test is monotone in the bounds (NaN-ignoring max/min included), so
// passing the innermost of them -- the box of the primitive's parent Node, its *gating box* --
// implies passing all the others.  The closest hit is then the smallest accepted t, the
// first primitive in depth-first order on exact ties (bvh.rs:62), whatever order the
// primitives are visited in.  So the kernels need not walk the reference's topology: they
// walk trees built here for traversal speed.  The GATE tree's leaf slots are the reference's
// *groups* (the 1..4 leaves that share a parent Node, contiguous in depth-first order) behind
// their exact gating boxes, and its interior boxes are unions of those, i.e. supersets -- a ray
// that misses a superset misses every gating box inside it, so skipping the subtree skips
// nothing the reference reaches: this tree reaches exactly what the reference reaches
// (the default walk, and the local-pool route's gates).  On the benchmark
// scenes it removes the chain of sixteen levels down which the reference carries the 50 x 50
// floor (every query used to read all eight folded records of it) and halves the records a
// query visits.  The FAST walk's tree's leaf slots are single primitives behind boxes of their own
// inside the gating box (tight_box below): a subset of what the reference reaches.

struct WalkGroup {
    double box[6];
    uint32_t ref;  // REF_RANGE: first primitive << 2 | count - 1
};

// The groups of the two-child tree in depth-first order.  `gate` is the box of record n itself.
void collect_groups(const FlatScene& f, uint32_t n, const double* gate, std::vector<WalkGroup>& out) {
    for (int c = 0; c < 2; c++) {
        const uint32_t r = f.child_ref[(size_t)n * 2 + c];
        const double* box = &f.child_box[((size_t)n * 2 + c) * 6];
        const uint32_t kind = r >> 30;
        if (kind == REF_INTERIOR) {
            collect_groups(f, r & 0x3fffffffu, box, out);
        } else {
            WalkGroup g;
            // a direct leaf has no box of its own in the reference (bvh.rs:297, :302): what gates it
            // is the box of the Node it hangs under
            const double* gb = kind == REF_SINGLE ? gate : box;
            for (int k = 0; k < 6; k++) g.box[k] = gb[k];
            g.ref = (REF_RANGE << 30) | (r & 0x3fffffffu);
            out.push_back(g);
        }
    }
}

struct WalkNode {  // binary tree over the groups
    double box[6];
    int32_t left, right;  // -1: leaf
    uint32_t ref;         // leaf: the group's reference
};

inline double box_area(const double* b) {
    const double x = b[1] - b[0], y = b[3] - b[2], z = b[5] - b[4];
    return 2.0 * (x * y + y * z + x * z);
}
inline void box_merge(double* a, const double* b) {
    for (int k = 0; k < 6; k += 2) {
        if (b[k] < a[k]) a[k] = b[k];
        if (b[k + 1] > a[k + 1]) a[k + 1] = b[k + 1];
    }
}

// Top-down binned surface-area-heuristic build (32 bins on each axis of the centroid bounds);
// every group ends in a leaf of its own.
// Top-down build with the exact surface-area heuristic: the groups are sorted by centroid once per axis;
// a node prices every one of its n - 1 split positions on every axis from one prefix and one suffix sweep
// over its boxes, and the chosen split partitions the three sorted sequences stably, so they stay sorted
// and the whole build is O(n log n).  (Round 2 began with 32 centroid bins per axis: on the headline scene,
// whose floor stretches the root's centroid range to fifty times the mesh, the exact sweep needs 4.5 instead
// of 5.9 records per query.)
struct WalkBuilder {
    const std::vector<WalkGroup>& groups;
    std::vector<WalkNode> nodes;
    std::vector<uint32_t> ord[3];  // the groups of the current ranges, sorted by centroid along x / y / z
    std::vector<double> cen[3];
    std::vector<double> suffix_;
    std::vector<uint32_t> tmp_;
    std::vector<uint8_t> left_;

    explicit WalkBuilder(const std::vector<WalkGroup>& g) : groups(g) {
        const size_t n = g.size();
        for (int a = 0; a < 3; a++) {
            cen[a].resize(n);
            for (size_t i = 0; i < n; i++) cen[a][i] = 0.5 * g[i].box[2 * a] + 0.5 * g[i].box[2 * a + 1];
            ord[a].resize(n);
            for (size_t i = 0; i < n; i++) ord[a][i] = (uint32_t)i;
            const std::vector<double>& c = cen[a];
            std::stable_sort(ord[a].begin(), ord[a].end(), [&](uint32_t x, uint32_t y) { return c[x] < c[y]; });
        }
        suffix_.resize(n);
        tmp_.resize(n);
        left_.assign(n, 0);
        nodes.reserve(2 * n);
    }

    int32_t build() {
        struct Work {
            int32_t node;
            size_t lo, hi;
        };
        std::vector<Work> todo;
        nodes.push_back(WalkNode());
        todo.push_back({0, 0, groups.size()});
        while (!todo.empty()) {
            const Work w = todo.back();
            todo.pop_back();
            double box[6];
            for (int k = 0; k < 6; k++) box[k] = groups[ord[0][w.lo]].box[k];
            for (size_t i = w.lo + 1; i < w.hi; i++) box_merge(box, groups[ord[0][i]].box);
            for (int k = 0; k < 6; k++) nodes[w.node].box[k] = box[k];
            if (w.hi - w.lo == 1) {
                nodes[w.node].left = nodes[w.node].right = -1;
                nodes[w.node].ref = groups[ord[0][w.lo]].ref;
                continue;
            }
            const size_t mid = split(w.lo, w.hi);
            const int32_t l = (int32_t)nodes.size();
            nodes.push_back(WalkNode());
            nodes.push_back(WalkNode());
            nodes[w.node].left = l, nodes[w.node].right = l + 1, nodes[w.node].ref = 0;
            todo.push_back({l + 1, mid, w.hi});
            todo.push_back({l, w.lo, mid});
        }
        return 0;
    }

    // Splits the range [lo, hi) of all three sequences and returns the split position (lo < mid < hi).
    size_t split(size_t lo, size_t hi) {
        const size_t n = hi - lo;
        double best_cost = std::numeric_limits<double>::infinity();
        int best_axis = -1;
        size_t best_pos = 0;
        for (int a = 0; a < 3; a++) {
            const uint32_t* o = ord[a].data() + lo;
            if (!(cen[a][o[n - 1]] > cen[a][o[0]])) continue;
            double acc[6];
            for (int k = 0; k < 6; k++) acc[k] = groups[o[n - 1]].box[k];
            suffix_[n - 1] = box_area(acc);
            for (size_t i = n - 1; i-- > 1;) {
                box_merge(acc, groups[o[i]].box);
                suffix_[i] = box_area(acc);
            }
            for (int k = 0; k < 6; k++) acc[k] = groups[o[0]].box[k];
            for (size_t i = 1; i < n; i++) {  // left = o[0, i), right = o[i, n)
                const double cost = box_area(acc) * (double)i + suffix_[i] * (double)(n - i);
                if (cost < best_cost) best_cost = cost, best_axis = a, best_pos = i;
                box_merge(acc, groups[o[i]].box);
            }
        }
        if (best_axis < 0) best_axis = 0, best_pos = n / 2;  // coincident centroids: any split will do
        const uint32_t* o = ord[best_axis].data() + lo;
        for (size_t i = 0; i < best_pos; i++) left_[o[i]] = 1;
        for (int a = 0; a < 3; a++) {
            if (a == best_axis) continue;
            uint32_t* q = ord[a].data() + lo;
            size_t nl = 0, nr = 0;
            for (size_t i = 0; i < n; i++) {
                if (left_[q[i]]) q[nl++] = q[i];
                else tmp_[nr++] = q[i];
            }
            for (size_t i = 0; i < nr; i++) q[nl + i] = tmp_[i];
        }
        for (size_t i = 0; i < best_pos; i++) left_[o[i]] = 0;
        return lo + best_pos;
    }
};

// children before parents
std::vector<int32_t> post_order(const std::vector<WalkNode>& nodes) {
    std::vector<int32_t> order, stack{0};
    order.reserve(nodes.size());
    while (!stack.empty()) {  // pre-order with right before left, reversed below
        const int32_t n = stack.back();
        stack.pop_back();
        order.push_back(n);
        if (nodes[n].left >= 0) stack.push_back(nodes[n].left), stack.push_back(nodes[n].right);
    }
    std::reverse(order.begin(), order.end());
    return order;
}

// Which nodes of the binary tree become four-slot records, and which are opened inside their parent's
// record, is chosen to minimise the expected number of records a ray reads: a record behind a box of
// area A is read with probability ~ A / A(root), so cost(n as a record) = A(n) + the best way to spend
// four slots on n's two subtrees, where a subtree given j slots either is one record (one slot) or is
// opened and splits its j slots between its own children -- the dynamic programme of Ylitie et al. 2017,
// with zero cost for the leaf slots (a group's gating box is tested in the record that holds it).
struct WideCollapse {
    const std::vector<WalkNode>& nodes;
    std::vector<float> cost;    // [n * 4 + (j - 1)]: subtree n in at most j slots
    std::vector<uint8_t> used;  // [n * 4 + (j - 1)]: slots it actually takes (1 = a record of its own)
    std::vector<uint8_t> argk;  // [n * 4 + (j - 1)]: of `used` slots, how many go to the left child

    explicit WideCollapse(const std::vector<WalkNode>& nd) : nodes(nd) {
        const size_t n = nd.size();
        cost.assign(n * 4, 0.f), used.assign(n * 4, 1), argk.assign(n * 4, 0);
        for (const int32_t ni : post_order(nd)) {
            const size_t i = (size_t)ni;
            const WalkNode& w = nd[i];
            if (w.left < 0) continue;  // a group: no record, no cost
            float d[5];
            uint8_t dk[5];
            for (int j = 2; j <= 4; j++) {
                d[j] = std::numeric_limits<float>::infinity(), dk[j] = 1;
                for (int k = 1; k < j; k++) {
                    const float c = cost[(size_t)w.left * 4 + (k - 1)] + cost[(size_t)w.right * 4 + (j - k - 1)];
                    if (c < d[j]) d[j] = c, dk[j] = (uint8_t)k;
                }
            }
            cost[i * 4] = (float)box_area(w.box) + d[4];
            used[i * 4] = 1, argk[i * 4] = dk[4];  // as a record: its four slots split dk[4] : 4 - dk[4]
            for (int j = 2; j <= 4; j++) {
                if (d[j] < cost[i * 4 + (j - 2)]) cost[i * 4 + (j - 1)] = d[j], used[i * 4 + (j - 1)] = (uint8_t)j, argk[i * 4 + (j - 1)] = dk[j];
                else cost[i * 4 + (j - 1)] = cost[i * 4 + (j - 2)], used[i * 4 + (j - 1)] = used[i * 4 + (j - 2)], argk[i * 4 + (j - 1)] = argk[i * 4 + (j - 2)];
            }
        }
    }

    // the slots subtree n fills when it is given j of them
    void expand(int32_t n, int j, int32_t* slots, int& ns) const {
        const WalkNode& w = nodes[n];
        if (w.left < 0 || used[(size_t)n * 4 + (j - 1)] == 1) {
            slots[ns++] = n;
            return;
        }
        const int u = used[(size_t)n * 4 + (j - 1)], k = argk[(size_t)n * 4 + (j - 1)];
        expand(w.left, k, slots, ns);
        expand(w.right, u - k, slots, ns);
    }
};

// One record per call: the slots the collapse gives node n's two subtrees.  Returns the record's reference;
// *stack_need is the number of stack entries a traversal below it can have pending.
uint32_t emit_wide(WalkTree& f, const WideCollapse& wc, int32_t n, uint32_t* stack_need) {
    const std::vector<WalkNode>& nodes = wc.nodes;
    const uint32_t rec = f.n();
    f.ref.resize(f.ref.size() + 4, REF_NONE << 30);
    f.box.resize(f.box.size() + 24, 0.0);
    int32_t slots[4] = {-1, -1, -1, -1};
    int ns = 0;
    const int k = wc.argk[(size_t)n * 4];
    wc.expand(nodes[n].left, k, slots, ns);
    wc.expand(nodes[n].right, 4 - k, slots, ns);
    uint32_t below = 0;
    for (int i = 0; i < ns; i++) {
        const WalkNode& c = nodes[slots[i]];
        uint32_t ref = c.ref;
        if (c.left >= 0) {
            uint32_t need = 0;
            ref = emit_wide(f, wc, slots[i], &need);
            if (need > below) below = need;
        }
        f.ref[(size_t)rec * 4 + i] = ref;
        for (int q = 0; q < 6; q++) f.box[((size_t)rec * 4 + i) * 6 + q] = c.box[q];
    }
    *stack_need = (uint32_t)(ns - 1) + below;
    return (REF_INTERIOR << 30) | rec;
}

// The fast walk's tree's leaf slots: every primitive alone behind a box of its own -- its bounding box (the one
// Bvh::build computes for it, geometry.rs bbox) widened on every side by LEAF_MARGIN of its largest extent,
// rounded outwards to f32 and clipped to its group's gating box.
//   Nothing is tested that the reference does not reach: the box lies inside the gating box, and the slab test is
// monotone in the bounds, so passing it implies passing the gating box and everything around that.
//   What is no longer tested is a primitive whose widened box the ray misses.  In exact arithmetic such a ray
// misses the primitive by more than the widening; the reference's own test (Moeller-Trumbore, the sphere's
// quadratic, the plane's rectangle) then rejects it unless its rounding moves the hit POINT by more than
// LEAF_MARGIN of the primitive's size.  That error is about eps * |origin - primitive| / angle to the
// primitive's plane: from 20 units away it takes a ray within 1e-11 rad of the plane of a primitive 0.02 across
// that it passes beside (1e-8 rad from 1e4 scene sizes away, and so on) -- like closest-hit culling a bet on the
// reference's arithmetic, measured (profiles/r04_tight_leaves.txt), not a construction;
// the default walk is over the gate tree, which makes neither bet (rayrs_render_params.fast_traversal asks for this one).
constexpr double LEAF_MARGIN = 0x1p-6;

void tight_box(const Aabb& b, const double* gate, double* out) {
    const double pb[6] = {b.xmin, b.xmax, b.ymin, b.ymax, b.zmin, b.zmax};
    double ext = 0.0;
    for (int a = 0; a < 3; a++) ext = std::max(ext, pb[2 * a + 1] - pb[2 * a]);
    const double m = ext * LEAF_MARGIN;
    for (int a = 0; a < 3; a++) {
        double lo = pb[2 * a] - m, hi = pb[2 * a + 1] + m;
        float lf = (float)lo, hf = (float)hi;
        if ((double)lf > lo) lf = std::nextafterf(lf, -std::numeric_limits<float>::infinity());
        if ((double)hf < hi) hf = std::nextafterf(hf, std::numeric_limits<float>::infinity());
        lo = (double)lf, hi = (double)hf;
        // (a bound that is not a number, from an object whose own box is not, leaves the gating box's bound)
        out[2 * a] = lo > gate[2 * a] ? lo : gate[2 * a];
        out[2 * a + 1] = hi < gate[2 * a + 1] ? hi : gate[2 * a + 1];
    }
}

void split_groups(const std::vector<WalkGroup>& groups, const std::vector<Aabb>& prim_box, std::vector<WalkGroup>& out) {
    out.reserve(groups.size() * 3);
    for (const WalkGroup& g : groups) {
        const uint32_t first = (g.ref & 0x3fffffffu) >> 2, count = (g.ref & 3u) + 1u;
        for (uint32_t i = 0; i < count; i++) {
            WalkGroup r;
            tight_box(prim_box[first + i], g.box, r.box);
            r.ref = (REF_RANGE << 30) | ((first + i) << 2);
            out.push_back(r);
        }
    }
}

void build_tree_over(const std::vector<WalkGroup>& leaves, WalkTree& t) {
    WalkBuilder b(leaves);
    b.build();
    t.box.reserve(leaves.size() * 12);
    t.ref.reserve(leaves.size() * 2);
    const WideCollapse wc(b.nodes);
    t.root_ref = emit_wide(t, wc, 0, &t.depth);
    front_largest(t);
}

// The hot group (layout.h HotGroupDev): the group whose gating box has the largest surface area, if that is at least
// HOT_MIN_AREA of the root Node's box -- a ray that enters the root box enters a box inside it with probability about
// the ratio of their areas, so such a group is one most rays test whatever else they meet, and its slot in the tree's
// top record is a test every ray makes anyway.  Taken out of the tree, the rest is built as before (gate_hot): the
// leaves of gate_hot and the hot group together are exactly the groups of the gate tree, each behind exactly its
// gating box (tests/test_bvh_builder.py checks that from the exports).  Scenes of a handful of groups keep the one tree:
// they take the local-pool route or walk two or three records.
constexpr double HOT_MIN_AREA = 0.25;
constexpr size_t HOT_MIN_GROUPS = 8;

void pick_hot_group(FlatScene& f, const std::vector<WalkGroup>& groups) {
    f.has_hot = false;
    f.gate_hot = WalkTree();
    std::memset(&f.hot, 0, sizeof(f.hot));
    if (groups.size() < HOT_MIN_GROUPS) return;
    size_t best = 0;
    double best_area = -1.0;
    for (size_t i = 0; i < groups.size(); i++) {
        const double a = box_area(groups[i].box);
        if (a > best_area) best_area = a, best = i;  // (not a number: never larger; ties: the first in depth-first order)
    }
    const double root_area = box_area(f.root_box);
    if (!(best_area >= HOT_MIN_AREA * root_area) || !std::isfinite(best_area)) return;
    std::vector<WalkGroup> rest;
    rest.reserve(groups.size() - 1);
    for (size_t i = 0; i < groups.size(); i++)
        if (i != best) rest.push_back(groups[i]);
    build_tree_over(rest, f.gate_hot);
    const WalkGroup& g = groups[best];
    for (int k = 0; k < 6; k++) f.hot.box[k] = g.box[k];
    f.hot.first = (g.ref & 0x3fffffffu) >> 2;
    f.hot.count = (g.ref & 3u) + 1u;
    f.has_hot = true;  // (the primitives' values are filled in by build_flat_scene, which has the objects)
}

// prim_box[p]: the reference's bounding box of the object behind primitive record p
void build_walk_trees(FlatScene& f, const std::vector<Aabb>& prim_box) {
    f.walk = WalkTree();
    f.gate = WalkTree();
    if ((f.root_ref >> 30) != REF_INTERIOR) {  // one bottom Node: its box is root_box, tested by trav_init
        f.walk.root_ref = f.gate.root_ref = f.root_ref;
        return;
    }
    std::vector<WalkGroup> groups;
    collect_groups(f, f.root_ref & 0x3fffffffu, f.root_box, groups);
    build_tree_over(groups, f.gate);
    pick_hot_group(f, groups);
    std::vector<WalkGroup> singles;
    split_groups(groups, prim_box, singles);
    build_tree_over(singles, f.walk);
}

inline bool boxes_f32_exact(const WalkTree& t) {
    for (size_t r = 0; r < t.ref.size(); r++) {
        if ((t.ref[r] >> 30) == REF_NONE) continue;  // box never read
        for (int k = 0; k < 6; k++)
            if (!f32_exact(t.box[r * 6 + k])) return false;
    }
    return true;
}

// The records as the kernels read them.  An unused slot gets the inverted box [+inf, -inf], which no ray
// enters, so that the kernel's slab test says the right thing without looking at the kind.
template <typename NODE, typename F>
void fill_nodes(WalkTree& t) {
    const uint32_t n = t.n();
    const double inf = std::numeric_limits<double>::infinity();
    t.node_bytes.assign((size_t)std::max(n, 1u) * sizeof(NODE), 0);
    NODE* nodes = reinterpret_cast<NODE*>(t.node_bytes.data());
    for (uint32_t r = 0; r < n; r++)
        for (int ch = 0; ch < 4; ch++) {
            const uint32_t ref = t.ref[(size_t)r * 4 + ch];
            nodes[r].ref[ch] = ref;
            for (int k = 0; k < 6; k++)
                nodes[r].box[ch][k] = (ref >> 30) == REF_NONE ? ((k & 1) ? (F)-inf : (F)inf) : (F)t.box[((size_t)r * 4 + ch) * 6 + k];
        }
}

void put_f64(uint32_t* dst, double v) { std::memcpy(dst, &v, 8); }
void put_f32(uint32_t* dst, float v) { std::memcpy(dst, &v, 4); }

}  // namespace

int build_flat_scene(const ObjectList& objs, double z_near, double z_far, int heuristic, uint32_t splits,
                     uint32_t hdri_w, uint32_t hdri_h, const float* hdri_rgb, FlatScene* out) {
    if (!(z_near >= 0.0) || !(z_far > z_near)) return RAYRS_INVALID_ARG;  // lib.rs:234-235
    if (objs.objs.empty()) return RAYRS_INVALID_ARG;                      // bvh.rs:229
    if (heuristic != RAYRS_BVH_MIDPOINT && heuristic != RAYRS_BVH_SAH) return RAYRS_INVALID_ARG;
    if (heuristic == RAYRS_BVH_SAH && splits < 2) return RAYRS_INVALID_ARG;
    if (hdri_w < 2 || hdri_h < 2 || !hdri_rgb) return RAYRS_INVALID_ARG;
    if (objs.objs.size() >= (1u << 28)) return RAYRS_UNSUPPORTED;

    const auto t_begin = std::chrono::steady_clock::now();
    FlatScene& f = *out;
    f = FlatScene();
    f.t0 = z_near;
    f.t1 = z_far;

    Builder b(objs, f, heuristic == RAYRS_BVH_SAH, splits);
    Aabb root;
    const size_t n = objs.objs.size();
    if (n == 1) {
        // Node(bbox, [Leaf]) -- bvh.rs:306-316 with one object
        root = b.boxes[0];
        f.prim_object[0] = 0;
        b.next_prim = 1;
        f.root_ref = (REF_RANGE << 30) | 0u;
    } else {
        f.root_ref = b.build(0, n, 0, &root);
    }
    f.root_box[0] = root.xmin, f.root_box[1] = root.xmax, f.root_box[2] = root.ymin;
    f.root_box[3] = root.ymax, f.root_box[4] = root.zmin, f.root_box[5] = root.zmax;

    // ---- the trees the kernels walk
    {
        std::vector<Aabb> prim_box(n);
        for (size_t p = 0; p < n; p++) prim_box[p] = b.boxes[f.prim_object[p]];
        build_walk_trees(f, prim_box);
        std::vector<double> ext(n);
        for (size_t p = 0; p < n; p++) {
            const Aabb& pb = prim_box[p];
            const double e = std::max(pb.xmax - pb.xmin, std::max(pb.ymax - pb.ymin, pb.zmax - pb.zmin));
            ext[p] = e > 0.0 ? e : std::numeric_limits<double>::infinity();  // (degenerate or not a number: not "small")
        }
        std::nth_element(ext.begin(), ext.begin() + (std::ptrdiff_t)(n / 20), ext.end());
        f.small_extent = std::isfinite(ext[n / 20]) ? ext[n / 20] : 0.0;
    }

    // ---- choose the layout
    bool compact = boxes_f32_exact(f.gate) && boxes_f32_exact(f.walk) && boxes_f32_exact(f.gate_hot);
    for (size_t i = 0; i < n && compact; i++) {
        const Shape& s = objs.objs[i].geom;
        if (s.kind != PRIM_TRIANGLE) continue;
        const double v[9] = {s.p1.x, s.p1.y, s.p1.z, s.p2.x, s.p2.y, s.p2.z, s.p3.x, s.p3.y, s.p3.z};
        for (double c : v)
            if (!f32_exact(c)) {
                compact = false;
                break;
            }
    }
    f.compact = compact;
    for (WalkTree* t : {&f.gate, &f.walk, &f.gate_hot}) {
        if (compact) fill_nodes<Node4F32, float>(*t);
        else fill_nodes<Node4F64, double>(*t);
    }

    // ---- primitive records in DFS order
    const uint32_t dw = compact ? PRIM_DWORDS_COMPACT : PRIM_DWORDS_FULL;
    f.prim_bytes.assign((size_t)n * dw * 4, 0);
    uint32_t* prims = reinterpret_cast<uint32_t*>(f.prim_bytes.data());
    for (size_t p = 0; p < n; p++) {
        const Object& o = objs.objs[f.prim_object[p]];
        const Shape& s = o.geom;
        uint32_t* rec = prims + p * dw;
        if (s.kind == PRIM_SPHERE) {
            put_f64(rec + 0, s.radius2);
            put_f64(rec + 2, s.origin.x);
            put_f64(rec + 4, s.origin.y);
            put_f64(rec + 6, s.origin.z);
        } else if (s.kind == PRIM_PLANE) {
            put_f64(rec + 0, s.u0);
            put_f64(rec + 2, s.u1);
            put_f64(rec + 4, s.v0);
            put_f64(rec + 6, s.v1);
            put_f64(rec + 8, s.pos);
        } else {
            const double v[9] = {s.p1.x, s.p1.y, s.p1.z, s.p2.x, s.p2.y, s.p2.z, s.p3.x, s.p3.y, s.p3.z};
            for (int k = 0; k < 9; k++) {
                if (compact)
                    put_f32(rec + k, (float)v[k]);
                else
                    put_f64(rec + 2 * k, v[k]);
            }
        }
        rec[dw - 1] = s.kind | (s.axis << 2) | (o.surface << 8);
    }

    // ---- the hot group's primitives as f64 values (Triangle::new's e1, e2: geometry.rs:342-343)
    if (f.has_hot) {
        for (uint32_t k = 0; k < f.hot.count; k++) {
            const Object& o = objs.objs[f.prim_object[f.hot.first + k]];
            const Shape& s = o.geom;
            HotPrim& hp = f.hot.prim[k];
            if (s.kind == PRIM_SPHERE) {
                hp.v[0] = s.radius2, hp.v[1] = s.origin.x, hp.v[2] = s.origin.y, hp.v[3] = s.origin.z;
            } else if (s.kind == PRIM_PLANE) {
                hp.v[0] = s.u0, hp.v[1] = s.u1, hp.v[2] = s.v0, hp.v[3] = s.v1, hp.v[4] = s.pos;
            } else {
                hp.v[0] = s.p1.x, hp.v[1] = s.p1.y, hp.v[2] = s.p1.z;
                hp.v[3] = s.p2.x - s.p1.x, hp.v[4] = s.p2.y - s.p1.y, hp.v[5] = s.p2.z - s.p1.z;
                hp.v[6] = s.p3.x - s.p1.x, hp.v[7] = s.p3.y - s.p1.y, hp.v[8] = s.p3.z - s.p1.z;
            }
            hp.tag = s.kind | (s.axis << 2) | (o.surface << 8);
            if (s.kind == PRIM_SPHERE) f.hot.n_sphere++;
            else if (s.kind == PRIM_PLANE) f.hot.n_plane++;
            else f.hot.n_tri++;
        }
        // the root record of the tree without the group, as the kernels' records hold it (fill_nodes)
        const WalkTree& t = f.gate_hot;
        const uint32_t r = t.root_ref & 0x3fffffffu;
        const double inf = std::numeric_limits<double>::infinity();
        for (int ch = 0; ch < 4; ch++) {
            const uint32_t ref = t.ref[(size_t)r * 4 + ch];
            f.hot.root_ref[ch] = ref;
            for (int k = 0; k < 6; k++)
                f.hot.root_box[ch][k] = (ref >> 30) == REF_NONE ? ((k & 1) ? -inf : inf) : t.box[((size_t)r * 4 + ch) * 6 + k];
        }
    }

    // ---- HDRI: clip(0, 3) (main.rs:43; clip = min(max).max(min), vecmath.rs:388-396), then one record per
    // texel (i, j) holding the four texels Scene::background reads for it -- (i, j), (i+1, j), (i, j+1),
    // (i+1, j+1), RGBA f32 each, out-of-range neighbours clamped exactly as device_path.h hdri_texel did --
    // so that a lookup is ONE aligned 64-byte read instead of two or three 128-byte lines
    f.hdri_w = hdri_w;
    f.hdri_h = hdri_h;
    {
        std::vector<float> rgb((size_t)hdri_w * hdri_h * 3);
        for (size_t t = 0; t < rgb.size(); t++) {
            const double v = rr_max(rr_min((double)hdri_rgb[t], 3.0), 0.0);
            rgb[t] = (float)v;  // v is an f32 value or 0 or 3: exact
        }
        f.hdri_quads.assign((size_t)hdri_w * hdri_h * 16, 0.f);
        for (uint32_t i = 0; i < hdri_h; i++) {
            const uint32_t i1 = i + 1 < hdri_h ? i + 1 : hdri_h - 1;
            for (uint32_t j = 0; j < hdri_w; j++) {
                const uint32_t j1 = j + 1 < hdri_w ? j + 1 : hdri_w - 1;
                const size_t src[4] = {(size_t)i * hdri_w + j, (size_t)i1 * hdri_w + j, (size_t)i * hdri_w + j1,
                                       (size_t)i1 * hdri_w + j1};
                float* q = &f.hdri_quads[((size_t)i * hdri_w + j) * 16];
                for (int k = 0; k < 4; k++)
                    for (int c = 0; c < 3; c++) q[k * 4 + c] = rgb[src[k] * 3 + c];
            }
        }
    }

    f.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    return RAYRS_OK;
}

// ------------------------------------------------------------------ camera

static inline Vec3 sub(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline Vec3 cross(Vec3 a, Vec3 b) {  // vecmath.rs:565-577
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline Vec3 unit(Vec3 a) {  // vecmath.rs:525-527 with Div<f64> :690-698
    const double inv = 1.0 / rr_sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return {a.x * inv, a.y * inv, a.z * inv};
}

int camera_new(const double origin[3], const double up[3], const double lookat[3], double fov, double width,
               double height, uint32_t ppi, rayrs_camera* out) {
    if (!origin || !up || !lookat || !out) return RAYRS_INVALID_ARG;
    if (!(fov > 0. && fov < 180.)) return RAYRS_INVALID_ARG;  // lib.rs:108
    if (!(width > 0.) || !(height > 0.)) return RAYRS_INVALID_ARG;
    const Vec3 o{origin[0], origin[1], origin[2]}, u{up[0], up[1], up[2]}, la{lookat[0], lookat[1], lookat[2]};
    if (o.x == la.x && o.y == la.y && o.z == la.z) return RAYRS_INVALID_ARG;  // lib.rs:111
    const uint32_t ppc = (uint32_t)std::round((double)ppi * 2.54);            // lib.rs:113
    const Vec3 z = unit(sub(la, o));
    const Vec3 x = unit(cross(u, z));
    const Vec3 y = unit(cross(z, x));
    const double rad = fov * (RR_PI / 180.0);  // f64::to_radians
    const double focal = width / rr_tan(rad / 2.);  // lib.rs:131
    out->origin[0] = o.x, out->origin[1] = o.y, out->origin[2] = o.z;
    out->e_x[0] = x.x, out->e_x[1] = x.y, out->e_x[2] = x.z;
    out->e_y[0] = y.x, out->e_y[1] = y.y, out->e_y[2] = y.z;
    out->z[0] = focal * z.x, out->z[1] = focal * z.y, out->z[2] = focal * z.z;
    out->width = width;
    out->height = height;
    out->ppc = ppc;
    out->x_pixels = (uint32_t)std::round(width * (double)ppc);   // lib.rs:153-155
    out->y_pixels = (uint32_t)std::round(height * (double)ppc);  // lib.rs:175-177
    return RAYRS_OK;
}

}  // namespace rayrs
