/* walk_sched_sim.c -- development aid, not product code: simulates how wf_trav_kernel's waves schedule their lanes
 * (rayrs_amd/csrc/wavefront.hip) on the rays a real frame makes, to price scheduling policies on the CPU before a
 * kernel is built for them.  Input: the walk tree the default walk reads (rayrs_scene_export_hot_tree), the rays of a
 * set of paths in item order (oracle: orc_set_ray_dump), a pool of slots.  Output: wave steps and lane counts per
 * phase kind.  Policies: 0 = the kernel as it is (a lane stands on ONE reference; interior or leaf phase by vote);
 * 1 = leaves set aside per lane (a lane walks on while its leaf groups wait in a queue of its own; leaf phases serve
 * one queued group per lane).  Build: gcc -O2 -shared -fPIC -ffp-contract=off -o walk_sched_sim.so walk_sched_sim.c */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t policy, refill_min, leaf_min, blocked_max, queue_cap, windows_per_wave, n_slots, int_min;
} sim_params;

typedef struct {
    uint64_t rounds, rays, walk_rays, int_steps, int_lanes, leaf_steps, leaf_lanes, leaf_prim_steps, refills, refill_lanes,
        records, prims, max_queue, queue_full_waits, leaf_groups;
} sim_stats;

static const double* g_box;
static const uint32_t* g_ref;
static double g_t0, g_t1;

static inline double fmax_(double a, double b) { return a > b ? a : b; } /* operands are never NaN here */
static inline double fmin_(double a, double b) { return a < b ? a : b; }

static int slab(const double* b, const double* o, const double* inv) {
    double tmin = g_t0, tmax = g_t1;
    for (int a = 0; a < 3; a++) {
        const double lo = b[2 * a], hi = b[2 * a + 1];
        const double n = inv[a] < 0.0 ? hi : lo, f = inv[a] < 0.0 ? lo : hi;
        tmin = fmax_(tmin, (n - o[a]) * inv[a]);
        tmax = fmin_(tmax, (f - o[a]) * inv[a]);
    }
    return !(tmax <= tmin);
}

static uint32_t record_mask(uint32_t rec, const double* o, const double* inv) {
    uint32_t m = 0;
    for (int c = 0; c < 4; c++)
        if ((g_ref[4 * rec + c] >> 30) != 3u && slab(g_box + (size_t)(rec * 4 + c) * 6, o, inv)) m |= 1u << c;
    return m;
}

#define NONE 0xffffffffu
typedef struct {
    int active;
    uint32_t cur;
    uint32_t st[128];
    int sp;
    uint32_t lq[64];
    int nlq;
    double o[3], inv[3];
} lane_t;

typedef struct {
    const double* ray; /* 7 doubles */
    uint32_t mask;
} ready_t;

static inline int is_int(uint32_t r) { return r != NONE && (r >> 30) == 0u; }
static inline int is_leaf(uint32_t r) { return r != NONE && (r >> 30) == 1u; }

/* the entered slots of `rec` for the lane: policy 0 -- first becomes cur, others stacked to pop in slot order;
 * policy 1 -- leaves to the lane's queue, the first interior slot becomes cur, the others stacked */
static void interior_step(lane_t* L, int policy, sim_stats* st) {
    const uint32_t rec = L->cur & 0x3fffffffu;
    const uint32_t m = record_mask(rec, L->o, L->inv);
    st->records++;
    uint32_t ent[4];
    int n = 0;
    for (int c = 0; c < 4; c++)
        if (m & (1u << c)) ent[n++] = g_ref[4 * rec + c];
    if (policy == 0) {
        if (n == 0) {
            L->cur = L->sp ? L->st[--L->sp] : NONE;
        } else {
            for (int k = n - 1; k >= 1; k--) L->st[L->sp++] = ent[k];
            L->cur = ent[0];
        }
    } else {
        uint32_t first = NONE;
        for (int k = n - 1; k >= 0; k--) {
            if (is_leaf(ent[k])) {
                L->lq[L->nlq++] = ent[k];
                if ((uint64_t)L->nlq > st->max_queue) st->max_queue = (uint64_t)L->nlq;
            } else {
                if (first != NONE) L->st[L->sp++] = first;
                first = ent[k];
            }
        }
        L->cur = first != NONE ? first : (L->sp ? L->st[--L->sp] : NONE);
    }
}

static void start_ray(lane_t* L, const ready_t* r, uint32_t root_rec, int policy, sim_stats* st) {
    for (int a = 0; a < 3; a++) L->o[a] = r->ray[a], L->inv[a] = 1.0 / r->ray[3 + a];
    L->sp = 0, L->nlq = 0, L->active = 1;
    uint32_t ent[4];
    int n = 0;
    for (int c = 0; c < 4; c++)
        if (r->mask & (1u << c)) ent[n++] = g_ref[4 * root_rec + c];
    if (policy == 0) {
        for (int k = n - 1; k >= 1; k--) L->st[L->sp++] = ent[k];
        L->cur = ent[0];
    } else {
        uint32_t first = NONE;
        for (int k = n - 1; k >= 0; k--) {
            if (is_leaf(ent[k])) L->lq[L->nlq++] = ent[k];
            else {
                if (first != NONE) L->st[L->sp++] = first;
                first = ent[k];
            }
        }
        L->cur = first;
    }
    st->walk_rays++;
}

/* one wave works through `n` ready rays (its windows' lists, concatenated) */
static void run_wave(const ready_t* list, size_t n, uint32_t root_rec, const sim_params* P, sim_stats* st) {
    static lane_t lanes[64];
    for (int l = 0; l < 64; l++) lanes[l].active = 0, lanes[l].cur = NONE, lanes[l].nlq = 0, lanes[l].sp = 0;
    size_t pos = 0;
    const int pol = (int)P->policy;
    for (;;) {
        int n_int = 0, n_leaf = 0, n_work = 0, n_blocked = 0;
        for (int l = 0; l < 64; l++) {
            lane_t* L = &lanes[l];
            if (!L->active) continue;
            if (pol == 0) {
                if (is_int(L->cur)) n_int++, n_work++;
                else if (is_leaf(L->cur)) n_leaf++, n_work++;
                else L->active = 0;
            } else {
                const int hi = L->cur != NONE, hl = L->nlq > 0;
                const int room = L->nlq + 4 <= (int)P->queue_cap;
                if (!hi && !hl) { L->active = 0; continue; }
                n_work++;
                if (hi && room) n_int++;
                if (hi && !room) st->queue_full_waits++;
                if (hl) n_leaf++;
                if (hl && !(hi && room)) n_blocked++;
            }
        }
        const int no_more = pos >= n;
        if ((n_work < (int)P->refill_min && !no_more) || n_work == 0) {
            if (no_more) break;
            int got = 0;
            for (int l = 0; l < 64 && pos < n; l++)
                if (!lanes[l].active) start_ray(&lanes[l], &list[pos++], root_rec, pol, st), got++;
            st->refills++, st->refill_lanes += (uint64_t)got;
            continue;
        }
        int leaf_phase;
        if (pol == 0) leaf_phase = n_leaf >= (int)P->leaf_min || n_int == 0;
        else leaf_phase = n_int == 0 || n_leaf >= (int)P->leaf_min || n_blocked >= (int)P->blocked_max ||
                          (n_int < (int)P->int_min && n_leaf > n_int);
        if (leaf_phase) {
            uint32_t maxc = 0;
            for (int l = 0; l < 64; l++) {
                lane_t* L = &lanes[l];
                if (!L->active) continue;
                uint32_t r = NONE;
                if (pol == 0) {
                    if (!is_leaf(L->cur)) continue;
                    r = L->cur;
                    L->cur = L->sp ? L->st[--L->sp] : NONE;
                } else {
                    if (L->nlq == 0) continue;
                    r = L->lq[--L->nlq];
                }
                const uint32_t c = (r & 3u) + 1u;
                st->prims += c, st->leaf_lanes++, st->leaf_groups++;
                if (c > maxc) maxc = c;
            }
            st->leaf_steps++, st->leaf_prim_steps += maxc;
        } else {
            for (int l = 0; l < 64; l++) {
                lane_t* L = &lanes[l];
                if (!L->active) continue;
                if (pol == 0) {
                    if (!is_int(L->cur)) continue;
                } else {
                    if (L->cur == NONE || L->nlq + 4 > (int)P->queue_cap) continue;
                }
                interior_step(L, pol, st);
                st->int_lanes++;
            }
            st->int_steps++;
        }
    }
}

/* rays: [n_rays][7] (o, d, loop iteration); a path starts where the iteration is 0; item i = paths 4i .. 4i+3 */
int sim_run(const double* box, const uint32_t* ref, uint32_t root_rec, const double root_node_box[6], double t0, double t1,
            const double* rays, uint64_t n_rays, const sim_params* P, sim_stats* out) {
    g_box = box, g_ref = ref, g_t0 = t0, g_t1 = t1;
    memset(out, 0, sizeof(*out));
    /* path starts */
    uint64_t n_paths = 0;
    for (uint64_t i = 0; i < n_rays; i++) n_paths += rays[7 * i + 6] == 0.0;
    uint64_t* pstart = (uint64_t*)malloc((n_paths + 1) * sizeof(uint64_t));
    uint64_t k = 0;
    for (uint64_t i = 0; i < n_rays; i++)
        if (rays[7 * i + 6] == 0.0) pstart[k++] = i;
    pstart[n_paths] = n_rays;
    const uint64_t n_items = n_paths / 4;
    const uint32_t S = P->n_slots;
    /* slot state: item, path within the item, ray within the path */
    uint64_t* s_item = (uint64_t*)malloc(S * sizeof(uint64_t));
    uint32_t* s_path = (uint32_t*)calloc(S, sizeof(uint32_t));
    uint64_t* s_ray = (uint64_t*)malloc(S * sizeof(uint64_t));
    uint64_t next_item = 0;
    for (uint32_t s = 0; s < S; s++) {
        s_item[s] = next_item < n_items ? next_item++ : (uint64_t)-1;
        s_ray[s] = s_item[s] != (uint64_t)-1 ? pstart[4 * s_item[s]] : 0;
    }
    ready_t* ready = (ready_t*)malloc((size_t)S * sizeof(ready_t));
    ready_t* sorted = (ready_t*)malloc((size_t)S * sizeof(ready_t));
    const uint32_t n_windows = S / 512;
    const uint32_t wpw = P->windows_per_wave ? P->windows_per_wave : 43;
    const uint32_t n_waves = (n_windows + wpw - 1) / wpw;
    for (;;) {
        if (next_item >= n_items) break; /* the frame's end (a thinning pool) is not simulated */
        out->rounds++;
        /* per window: the READY list, by first slot entered */
        size_t* wbeg = (size_t*)malloc((n_windows + 1) * sizeof(size_t));
        size_t nr = 0;
        for (uint32_t w = 0; w < n_windows; w++) {
            wbeg[w] = nr;
            size_t nw = 0;
            for (uint32_t j = 0; j < 512; j++) {
                const uint32_t s = w * 512 + j;
                if (s_item[s] == (uint64_t)-1) continue;
                const double* r = rays + 7 * s_ray[s];
                out->rays++;
                double inv[3] = {1.0 / r[3], 1.0 / r[4], 1.0 / r[5]};
                double rb[6] = {root_node_box[0], root_node_box[1], root_node_box[2], root_node_box[3], root_node_box[4],
                                root_node_box[5]};
                if (!slab(rb, r, inv)) continue;
                const uint32_t m = record_mask(root_rec, r, inv);
                out->records++; /* the first record, tested by the kernel that made the ray */
                if (!m) continue;
                ready[nw].ray = r, ready[nw].mask = m, nw++;
            }
            for (uint32_t key = 0; key < 4; key++)
                for (size_t i = 0; i < nw; i++)
                    if ((ready[i].mask & (0u - ready[i].mask)) == (1u << key)) sorted[nr++] = ready[i];
        }
        wbeg[n_windows] = nr;
        /* waves: wave g takes windows g, g + n_waves, ... */
        ready_t* mine = (ready_t*)malloc((nr + 1) * sizeof(ready_t));
        for (uint32_t g = 0; g < n_waves; g++) {
            size_t n = 0;
            for (uint32_t w = g; w < n_windows; w += n_waves) {
                memcpy(mine + n, sorted + wbeg[w], (wbeg[w + 1] - wbeg[w]) * sizeof(ready_t));
                n += wbeg[w + 1] - wbeg[w];
            }
            run_wave(mine, n, root_rec, P, out);
        }
        free(mine);
        free(wbeg);
        /* advance the slots */
        for (uint32_t s = 0; s < S; s++) {
            if (s_item[s] == (uint64_t)-1) continue;
            const uint64_t p = 4 * s_item[s] + s_path[s];
            if (s_ray[s] + 1 < pstart[p + 1]) {
                s_ray[s]++;
            } else if (s_path[s] < 3) {
                s_path[s]++;
                s_ray[s] = pstart[p + 1];
            } else {
                s_path[s] = 0;
                s_item[s] = next_item < n_items ? next_item++ : (uint64_t)-1;
                if (s_item[s] != (uint64_t)-1) s_ray[s] = pstart[4 * s_item[s]];
            }
        }
    }
    free(pstart), free(s_item), free(s_path), free(s_ray), free(ready), free(sorted);
    return 0;
}
