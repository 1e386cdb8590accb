// rtx_wide.hpp — from a binary BVH to the slots of an 8-wide node: which binary subtrees become wide nodes or leaf slots (surface-area-heuristic dynamic program after Ylitie,
// Karras, Laine, HPG 2017, section 3.1), the children of one wide node, and their assignment to octant slots.
//
// ONE source for two builders: the host collapse (csrc/rtx_scene_host.cpp: collapse_bvh8) and the GPU build (csrc/rtx_build.hip) include this file, so that both take the same
// decisions from the same inputs — same operations in the same order, double arithmetic where the host always had it, no contraction (-ffp-contract=off on both sides).
// A tree built on the GPU can therefore be compared with its host twin node for node (tests: test_gpu_build_equals_its_host_twin).
#pragma once
#include <stdint.h>
#include <math.h>
#include "rtx_math.hpp"

namespace rtx {

struct WBox { float mn[3], mx[3]; };
RTX_HD float wbox_area(const WBox& b) { const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2]; return dx * dy + dy * dz + dz * dx; }   // (half the surface area)
RTX_HD WBox wbox_union(const WBox& a, const WBox& b) {
    WBox u;
    for (int k = 0; k < 3; k++) { u.mn[k] = b.mn[k] < a.mn[k] ? b.mn[k] : a.mn[k]; u.mx[k] = a.mx[k] < b.mx[k] ? b.mx[k] : a.mx[k]; }      // std::min(a, b) / std::max(a, b): the first operand on a tie (+0 / -0)
    return u;
}

// The dynamic program's record of one binary node n.  cost[i], i = 1 .. 7: the cheapest way to represent the subtree with at most i child slots of its wide parent — as ONE
// slot (a leaf slot holding all its <= 4 triangles, or an internal slot = a wide node of its own with 8 slots to distribute) or split between its two children.
// choice[1]: 0 leaf slot / 1 internal slot; choice[i >= 2]: 0 = as with i - 1 slots, k = the left child gets k of the i slots; choice[8]: the split of the node's own 8 slots.
struct WideDp { double area; double cost[8]; uint32_t prims; uint8_t choice[9]; uint8_t pad_[3]; };

// one child of a binary node as the program sees it: a binary LEAF with `leaf_cnt` triangles (dp == nullptr) or an internal binary node with its record
struct WideDpChild { float area; uint32_t leaf_cnt; const WideDp* dp; };
RTX_HD double wide_child_cost(const WideDpChild& c, int i, double tri_cost) {
    if (!c.dp) return (double)c.area * tri_cost * c.leaf_cnt;
    return c.dp->cost[i < 7 ? i : 7];
}
RTX_HD uint32_t wide_child_prims(const WideDpChild& c) { return c.dp ? c.dp->prims : c.leaf_cnt; }
// record of the node whose children are L and R and whose (padded) box has half-area `area_u`; node_cost = 1: a node step, tri_cost: a triangle test relative to it
RTX_HD void wide_dp_combine(const WideDpChild& L, const WideDpChild& R, float area_u, double node_cost, double tri_cost, WideDp& out) {
    out.area = (double)area_u; out.prims = wide_child_prims(L) + wide_child_prims(R);
    for (int i = 0; i < 3; i++) out.pad_[i] = 0;
    out.cost[0] = 0.0; out.choice[0] = 0;
    double c8 = INFINITY; uint8_t k8 = 1;
    for (int k = 1; k < 8; k++) { const double c = wide_child_cost(L, k, tri_cost) + wide_child_cost(R, 8 - k, tri_cost); if (c < c8) { c8 = c; k8 = (uint8_t)k; } }
    out.choice[8] = k8;
    const double c_int = c8 + out.area * node_cost;
    const double c_leaf = out.prims <= 4u ? out.area * tri_cost * out.prims : (double)INFINITY;
    out.cost[1] = c_int < c_leaf ? c_int : c_leaf; out.choice[1] = c_leaf <= c_int ? 0 : 1;      // std::min(c_leaf, c_int)
    for (int i = 2; i <= 7; i++) {
        double best = INFINITY; uint8_t kb = 1;
        for (int k = 1; k < i; k++) { const double c = wide_child_cost(L, k, tri_cost) + wide_child_cost(R, i - k, tri_cost); if (c < best) { best = c; kb = (uint8_t)k; } }
        if (best < out.cost[i - 1]) { out.cost[i] = best; out.choice[i] = kb; } else { out.cost[i] = out.cost[i - 1]; out.choice[i] = 0; }
    }
}

// ---- the children of the wide node made from one binary node, following the recorded decisions ----
// A builder describes its binary tree through an accessor `A`:
//   struct Ref { WBox box; <code> c; }                        a child reference: its (padded) box and the builder's own code for it
//   bool         A.is_leaf(const Ref&)                        a binary leaf (becomes a leaf slot as it is)
//   void         A.children(const Ref&, Ref& L, Ref& R)       the two children of an internal binary node
//   uint8_t      A.choice(const Ref&, int i)                  the dynamic program's decision i of that node
//   Ref          A.merged(const Ref&)                         the same node as ONE leaf slot holding all its (<= 4) triangles
// out[]: the wide node's children in the order the recursion "left subtree first" meets them (at most 8); internal[k]: child k becomes a wide node of its own
template <class A, class Ref>
RTX_HD int wide_children(const A& acc, const Ref& L, const Ref& R, int k_left, Ref* out, bool* internal) {
    struct It { Ref r; int budget; };
    It st[16]; int sp = 0, m = 0;
    st[sp++] = It{R, 8 - k_left}; st[sp++] = It{L, k_left};
    while (sp > 0) {
        const It it = st[--sp];
        if (acc.is_leaf(it.r)) { if (m < 8) { out[m] = it.r; internal[m] = false; } m++; continue; }
        int i = it.budget < 7 ? it.budget : 7;
        while (i > 1 && acc.choice(it.r, i) == 0) i--;
        if (i == 1) {
            const bool leaf_slot = acc.choice(it.r, 1) == 0;
            if (m < 8) { out[m] = leaf_slot ? acc.merged(it.r) : it.r; internal[m] = !leaf_slot; }
            m++; continue;
        }
        const int k = acc.choice(it.r, i);
        Ref a, b; acc.children(it.r, a, b);
        if (sp + 2 > 16) return 9;                           // (cannot happen: a budget of 8 opens at most 7 nodes)
        st[sp++] = It{b, i - k}; st[sp++] = It{a, k};
    }
    return m;
}

// ---- slots: the child with the largest projection on an octant's diagonal gets that octant's slot (greedy assignment).  Visiting hit children in increasing (slot ^ octant)
//      then approximates front-to-back order without sorting.  bmn / bmx: the union of the children's boxes (the wide node's own box). ----
RTX_HD void wide_assign_slots(const WBox* ch, int m, float bmn[3], float bmx[3], int* slot_of) {
    for (int a = 0; a < 3; a++) {
        float lo = INFINITY, hi = -INFINITY;
        for (int k = 0; k < m; k++) { lo = ch[k].mn[a] < lo ? ch[k].mn[a] : lo; hi = hi < ch[k].mx[a] ? ch[k].mx[a] : hi; }      // std::min(lo, x) / std::max(hi, x)
        if (m == 0) { lo = 0.0f; hi = 0.0f; }
        bmn[a] = lo; bmx[a] = hi;
    }
    double cost[8][8];
    for (int k = 0; k < m; k++) for (int sl = 0; sl < 8; sl++) {
        double c = 0.0;
        for (int a = 0; a < 3; a++) {
            const double rel = 0.5 * ((double)ch[k].mn[a] + (double)ch[k].mx[a]) - 0.5 * ((double)bmn[a] + (double)bmx[a]);
            c += ((sl >> a) & 1) ? rel : -rel;
        }
        cost[k][sl] = c;
    }
    bool done[8] = {false, false, false, false, false, false, false, false}, used[8] = {false, false, false, false, false, false, false, false};
    for (int it = 0; it < m; it++) {
        int bk = -1, bs = -1; double bc = 0.0;
        for (int k = 0; k < m; k++) if (!done[k]) for (int sl = 0; sl < 8; sl++) if (!used[sl]) if (bk < 0 || cost[k][sl] > bc) { bc = cost[k][sl]; bk = k; bs = sl; }
        done[bk] = true; used[bs] = true; slot_of[bk] = bs;
    }
}

// ---- PLOC (Meister & Bittner 2018), the pieces host twin and device share (rtx_scene_host.cpp: ploc_clusters; rtx_build.hip: k_ploc_*) ----
RTX_HD uint64_t ploc_spread21(uint32_t v) {            // 21 bits -> every third bit of 63
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull; x = (x | x << 16) & 0x1f0000ff0000ffull; x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull; x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
// Morton key of a box's centre on the 2^21 grid over the scene's box [lo, lo + ext]: inv[a] = ext[a] > 0 ? 2097151 / ext[a] : 0
RTX_HD void ploc_grid(const float smn[3], const float smx[3], float lo[3], float inv[3]) {
    for (int a = 0; a < 3; a++) { lo[a] = smn[a]; const float ext = smx[a] - smn[a]; inv[a] = ext > 0.0f ? 2097151.0f / ext : 0.0f; }
}
RTX_HD uint64_t ploc_morton(const WBox& b, const float lo[3], const float inv[3]) {
    uint64_t code = 0;
    for (int a = 0; a < 3; a++) {
        const float c = 0.5f * (b.mn[a] + b.mx[a]);
        float q = (c - lo[a]) * inv[a]; q = q < 0.0f ? 0.0f : (q > 2097151.0f ? 2097151.0f : q);
        code |= ploc_spread21((uint32_t)q) << a;
    }
    return code;
}
RTX_HD float ploc_union_area(const WBox& a, const WBox& b) {          // half the surface area of the union (0 for an inverted box, as the builder's half_area)
    const WBox u = wbox_union(a, b);
    const float dx = u.mx[0] - u.mn[0], dy = u.mx[1] - u.mn[1], dz = u.mx[2] - u.mn[2];
    if (dx < 0) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
}
// nearest neighbour of cluster i among the places i - radius .. i + radius of a list of m: the smallest union area, scanned outwards (i - 1, i + 1, i - 2, ...) so that of equally
// good partners the nearest place along the curve wins — on regular tessellations most candidates tie, and "lowest place wins" pairs almost nobody.  box_at(j): box of place j.  -1: m == 1
template <class BoxAt>
RTX_HD int ploc_nearest(int i, int m, int radius, const BoxAt& box_at) {
    const WBox me = box_at(i);
    float best = INFINITY; int bj = -1;
    for (int d = 1; d <= radius; d++) {
        if (i - d >= 0) { const float a = ploc_union_area(me, box_at(i - d)); if (a < best) { best = a; bj = i - d; } }
        if (i + d < m) { const float a = ploc_union_area(me, box_at(i + d)); if (a < best) { best = a; bj = i + d; } }
    }
    return bj;
}

}  // namespace rtx
