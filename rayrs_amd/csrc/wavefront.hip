// wavefront.hip -- the radiance integrator as a stream of paths through three
// gfx950 kernels per round:
//
//   trav_kernel   persistent waves, one BVH query per lane, lanes refilled from the
//                 pool as queries finish; per-lane stack (and, the default walk, a
//                 queue of leaf groups set aside) in LDS          (bvh.rs:391-415)
//   hit_kernel    Material::evaluate + emission + Russian roulette (lib.rs:528-547)
//   miss_kernel   Scene::background                               (lib.rs:555)
// When a path ends in the hit or miss kernel the same lane adds the sample to the
// item's sum and starts the item's next sample (or takes the next item from the
// device-wide counter) and emits its primary ray (main.rs:67-79, lib.rs:202-210);
// gen_kernel does that once for the initial fill of the pool.
//
// A slot is one (pixel, sample-chunk) item with at most one path in flight, so the
// samples of an item are summed in order, exactly as the reference's per-pixel loop
// does.  Every slot carries a one-byte state; a kernel takes a window of 512
// consecutive slots, compacts the slots that are in its state into a list in LDS
// (__ballot + popcount rank) and works through the list 64 at a time, so its waves
// run with all lanes on the same code.  There are no global queues: a single-word
// atomic counter saturates near 90 updates/us on this chip, and the first version of
// this file, which pushed every slot through global queues, spent most of its time
// there.  What is left of atomics is one per 512-slot window for half of the
// traversal kernel's windows (load balance) and one per 256 items for new work.
// Kernel boundaries on one stream order the state changes; no in-kernel cross-CU
// hand-off is needed.
#include <hip/hip_runtime.h>

#include "device_path.h"
#include "kernels.h"
#include "lab_ticks.h"
#include "wavefront.h"

namespace rayrs {

constexpr uint32_t SPL = 8;             // state bytes per lane
constexpr uint32_t WINDOW = 64 * SPL;  // slots per window
constexpr uint32_t HIT_SURFACES_LDS = 32;

// Builds, in LDS, the list of slots of window `win` whose state is `want`.
// Returns the list length (wave-uniform).  list entries are offsets inside the window.
static_assert(SPL == 8, "StateWords/BatchFeed assume two words per lane");
struct StateWords {
    uint32_t w[SPL / 4];  // this lane's SPL state bytes of a window
};

RR_DEV StateWords load_state_words(const WfDev& wf, uint32_t win) {
    // np is a multiple of 1024 (the host rounds the pool up), so the load is in range
    const uint32_t* src = reinterpret_cast<const uint32_t*>(wf.state + win * WINDOW) + (threadIdx.x & 63u) * (SPL / 4);
    StateWords r;
#pragma unroll
    for (uint32_t k = 0; k < SPL / 4; k++) r.w[k] = src[k];
    return r;
}

RR_DEV uint32_t compact_words(const StateWords& sw, uint8_t want, uint16_t* list) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lanemask_lt = (1ull << lane) - 1ull;  // (v_mbcnt here costs the traversal kernel its last registers)
    const uint32_t* words = sw.w;
    uint32_t count = 0;
#pragma unroll
    for (int j = 0; j < (int)SPL; j++) {
        const uint32_t s = (words[j >> 2] >> ((j & 3) * 8)) & 0xffu;
        const bool m = s == (uint32_t)want;
        const unsigned long long mask = __ballot(m);
        if (m) list[count + (uint32_t)__popcll(mask & lanemask_lt)] = (uint16_t)(lane * SPL + (uint32_t)j);
        count += (uint32_t)__popcll(mask);
    }
    return count;
}

RR_DEV uint32_t compact_window(const WfDev& wf, uint32_t win, uint8_t want, uint16_t* list) {
    return compact_words(load_state_words(wf, win), want, list);
}

// A READY state carries the octant of its ray's direction in the high nibble, and the traversal kernel builds its list
// octant by octant: the rays a wave takes together look the same way, so they meet the tree's records in much the same
// order (same-box A/B: traversal -1.7 % on the headline frame, -2.9 % on config 3; a fourth key bit -- steeper than
// 45 degrees or not -- doubles the passes and gains less).
RR_DEV uint8_t ready_state(V3 d) {
    const uint32_t key = (d.x < 0.0 ? 1u : 0u) | (d.y < 0.0 ? 2u : 0u) | (d.z < 0.0 ? 4u : 0u);
    return (uint8_t)(WF_READY | (key << 4));
}
RR_DEV uint32_t compact_window_ready(const WfDev& wf, uint32_t win, uint16_t* list) {
    const StateWords sw = load_state_words(wf, win);
    uint32_t count = 0;
#pragma nounroll  // (unrolled, the 64 ballots' masks cost the kernel registers it does not have)
    for (uint32_t key = 0; key < 8u; key++) count += compact_words(sw, (uint8_t)(WF_READY | (key << 4)), list + count);
    return count;
}

// A PRE-TESTED ray's READY state (finish_rays) carries, instead of the octant, WHICH slots of the walk tree's first record it
// enters (bits 3..6; at least one): the traversal kernel starts the walk below those slots -- the record has been tested, by the
// kernel that made the ray, with the arithmetic and the values trav_interior_step would test it with -- and builds its
// list slot by slot of the FIRST slot entered: the rays a wave takes together start in the same quarter of the scene.
// A list entry carries the mask above the slot's offset in its window (9 bits).
RR_DEV uint8_t ready_state_pre(uint32_t mask) { return (uint8_t)(WF_READY | (mask << 3)); }
RR_DEV uint32_t compact_window_ready_pre(const WfDev& wf, uint32_t win, uint16_t* list) {
    const StateWords sw = load_state_words(wf, win);
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lanemask_lt = (1ull << lane) - 1ull;
    uint32_t count = 0;
#pragma nounroll
    for (uint32_t key = 0; key < 4u; key++) {
#pragma unroll
        for (int j = 0; j < (int)SPL; j++) {
            const uint32_t s = (sw.w[j >> 2] >> ((j & 3) * 8)) & 0xffu;
            const uint32_t m = (s >> 3) & 15u;
            const bool hit = (s & 7u) == (uint32_t)WF_READY && (m & (0u - m)) == (1u << key);
            const unsigned long long mask = __ballot(hit);
            if (hit) list[count + (uint32_t)__popcll(mask & lanemask_lt)] = (uint16_t)((lane * SPL + (uint32_t)j) | (m << 9));
            count += (uint32_t)__popcll(mask);
        }
    }
    return count;
}

// The hit and miss kernels' walk over their slots: wave g of n_waves takes windows g,
// g + n_waves, ... and within a window the slots of state `want`, 64 at a time.  The state
// bytes of the following window are loaded while the current one is being worked on, and the
// kernels fetch batch b + 1's slot records before they compute batch b, so that a wave waits
// for memory once per batch (the record that depends on the slot's contents) instead of three
// times.  Fewer than 64 slots left over from a window are carried into the next one's list (the
// list holds pool-wide slot indices), so every batch but a wave's last is full: at the headline
// frame's 58 % / 34 % of a window that is 4.6 instead of 5 and 2.7 instead of 3 batches per window.
constexpr uint32_t FEED_LIST = WINDOW + 64;  // entries per wave

struct BatchFeed {
    uint32_t next_win, n_waves, n_windows, count, k, total;  // wave-uniform
    StateWords ahead;                                     // state bytes of window next_win
    uint8_t want;
    uint32_t* list;
};

RR_DEV uint32_t compact_words_abs(const StateWords& sw, uint8_t want, uint32_t base, uint32_t* list) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t* words = sw.w;
    uint32_t count = 0;
#pragma unroll
    for (int j = 0; j < (int)SPL; j++) {
        const uint32_t s = (words[j >> 2] >> ((j & 3) * 8)) & 0xffu;
        const bool m = s == (uint32_t)want;
        const unsigned long long mask = __ballot(m);
        if (m) list[count + lanes_below(mask)] = base + lane * SPL + (uint32_t)j;
        count += (uint32_t)__popcll(mask);
    }
    return count;
}

RR_DEV void feed_init(BatchFeed& f, const WfDev& wf, uint32_t wave, uint32_t n_waves, uint8_t want, uint32_t* list) {
    f.next_win = wave, f.n_waves = n_waves, f.n_windows = wf.np / WINDOW;
    f.count = 0, f.k = 0, f.total = 0, f.want = want, f.list = list;
    f.ahead.w[0] = f.ahead.w[1] = 0;
    if (f.next_win < f.n_windows) f.ahead = load_state_words(wf, f.next_win);
}

// Next batch: false when the wave's windows are exhausted.
RR_DEV bool feed_next(BatchFeed& f, const WfDev& wf, uint32_t& slot, bool& valid) {
    const uint32_t lane = threadIdx.x & 63u;
    while (f.count - f.k < 64u && f.next_win < f.n_windows) {
        const uint32_t left = f.count - f.k;  // < 64: one entry per lane, moved to the front
        const uint32_t carry = lane < left ? f.list[f.k + lane] : 0u;
        if (lane < left) f.list[lane] = carry;
        const uint32_t fresh = compact_words_abs(f.ahead, f.want, f.next_win * WINDOW, f.list + left);
        f.total += fresh;
        f.count = left + fresh;
        f.k = 0;
        f.next_win += f.n_waves;
        if (f.next_win < f.n_windows) f.ahead = load_state_words(wf, f.next_win);
    }
    if (f.k >= f.count) return false;
    valid = f.k + lane < f.count;
    slot = valid ? f.list[f.k + lane] : 0u;
    f.k += 64u;
    return true;
}

// Sum over the 64 lanes (every lane must call it); the result is valid in lane 0 (and all lanes).
RR_DEV unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}

// One atomic per wave for a per-lane counter.
RR_DEV void wave_atomic_add(unsigned long long* dst, unsigned long long v) {
    const unsigned long long s = wave_sum(v);
    if ((threadIdx.x & 63u) == 0 && s) atomicAdd(dst, s);
}

// ---- kernel arguments, re-read where they are used ----
// The gen, hit and miss kernels take (SceneDev, CameraDev, RenderDev, WfDev) by value: 488 bytes of scalars, loop invariants
// all.  Left to itself the compiler loads them at the kernel's entry, runs out of scalar registers in the hit kernel's loop,
// parks them in lanes of a vector register and fetches each back with a v_readlane -- a VECTOR instruction per dword -- where
// it is used: 330 of them in next_sample and finish_rays.  The arguments sit in memory already (the kernarg segment, served
// by the scalar cache): karg<T, OFF>() hands out a view of one of them behind a pointer the compiler cannot see through, so
// the fields are s_load-ed where the view is made and live only as long as that region needs them.
constexpr uint32_t ka_up(uint32_t off, uint32_t a) { return (off + a - 1u) / a * a; }
constexpr uint32_t KA_SC = 0;
constexpr uint32_t KA_CAM = ka_up(KA_SC + (uint32_t)sizeof(SceneDev), (uint32_t)alignof(CameraDev));
constexpr uint32_t KA_RP = ka_up(KA_CAM + (uint32_t)sizeof(CameraDev), (uint32_t)alignof(RenderDev));
constexpr uint32_t KA_WF = ka_up(KA_RP + (uint32_t)sizeof(RenderDev), (uint32_t)alignof(WfDev));
template <class T, uint32_t OFF>
RR_DEV const T& karg() {
    typedef const __attribute__((address_space(4))) char* KPtr;
    KPtr p = (KPtr)__builtin_amdgcn_kernarg_segment_ptr() + OFF;
    asm volatile("" : "+s"(p));
    return *(const T*)p;
}

RR_DEV RaySlot* ray_slot(const WfDev& wf, uint32_t slot) { return &wf.slots[slot].ray; }
RR_DEV TailSlot* tail_slot(const WfDev& wf, uint32_t slot) { return &wf.slots[slot].tail; }
RR_DEV double* light_slot(const WfDev& wf, uint32_t slot) { return wf.light + (size_t)slot * 4u; }

// ------------------------------------------------------------------- init

__global__ void __launch_bounds__(256) wf_init_kernel(WfDev wf, uint32_t live) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        WfCtl* c = wf.ctl;
        c->next_window = 0;
        c->live_slots = live;
    }
    if (i < 2u * wf.n_flat_waves) wf.wave_items[i] = 0ull;
    if (i >= wf.np) return;
    // (the slot records are not touched: an IDLE slot has no item by definition -- wf_gen_kernel reads nothing of it --
    // and nothing ever reads a DEAD one; round 3 wrote every slot's cursor word here, 3.9 GB per headline frame)
    wf.state[i] = i < live ? WF_IDLE : WF_DEAD;
}

// -------------------------------------------------------------------- gen

// For every lane with `want`: the slot's path has ended (or it never had one).
// Write out the item if its samples are all done, take the next sample -- or the
// next item from the device-wide counter -- and emit its primary ray; the slot
// becomes READY, or DEAD when the counter has run out.  Called by all 64 lanes
// (it uses wave ballots); lanes without `want` only take part in those.
// The item record is passed in registers (ItemRegs): its loads are issued by the caller
// together with the slot's other records, so that they are not one more memory round trip
// at the end of the lane's work.
struct ItemRegs {
    double acc[3];
    uint32_t item, s_end, pix;
    // TailSlot::s_cur AS LOADED: sample cursor | SLOT_LIGHT_BIT | SLOT_ITEM_BIT.  The fields are taken apart where they are
    // used (cursor / has_item / has_light below), not where the word is loaded: load_item runs for batch b + 1 while batch b
    // is still to be computed, and three bit operations on the loaded word there made the wave wait for the requests it
    // had just issued -- a full memory latency at the top of every batch, the look-ahead undone (hit kernel: s_waitcnt
    // vmcnt(4) behind its eight slot requests; found in the ISA, round 6).
    uint32_t word;
    RR_DEV uint32_t cursor() const { return word & SLOT_SAMPLE_MASK; }
    RR_DEV bool has_item() const { return (word >> 31) != 0u; }
    RR_DEV bool has_light() const { return ((word >> 30) & 1u) != 0u; }  // the path's light is in the side array (else it is +0)
};

RR_DEV ItemRegs load_item(const WfDev& wf, uint32_t slot) {
    const TailSlot* t = tail_slot(wf, slot);
    ItemRegs r;
    r.acc[0] = t->acc[0], r.acc[1] = t->acc[1], r.acc[2] = t->acc[2];
    r.word = t->s_cur;
    r.item = t->item, r.s_end = t->s_end, r.pix = t->pix;
    return r;
}

// The RNG key of the sample a slot has in flight: the item's pixel and the sample before its cursor.
RR_DEV uint64_t sample_key(const RenderDev& rp, const CameraDev& cam, const ItemRegs& ir) {
    return rr_path_key(rp.seed, (uint64_t)(ir.pix >> 16) * cam.W + (ir.pix & 0xffffu), (uint64_t)(ir.cursor() - 1u));
}

// A wave's private range of reserved item ids [next, end).  Items are taken from the
// device-wide counter ITEM_RESERVE at a time (one atomic), not one batch at a time: with
// ~10^5 batches per round finishing items, per-batch atomics on the single counter word
// (which saturates near 90 updates/us on this chip) would cost more than the shading itself.
// The range lives in registers during a launch and in WfDev::wave_items between launches.
constexpr uint32_t ITEM_RESERVE = 256;
constexpr unsigned long long ITEMS_GONE = ~0ull;  // ItemRange::end of a wave that found the counter exhausted

struct ItemRange {
    unsigned long long next, end;
};

RR_DEV unsigned long long wave_uniform64(unsigned long long v) {  // a value every lane holds alike, into scalar registers
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// SCALAR (the hit kernel): the range in scalar registers.  Left in the vector registers the loads return it in, it makes
// next_sample's item loop wait for every request in flight at each turn -- the hit kernel's look-ahead loads among them.
// The miss kernel, whose look-ahead is requested at its loop's top and waited for at Scene::background's lookup anyway,
// measured 13 ms SLOWER with the scalar range (230.8 against 243.9 ms a frame, profiles/r06_leaf_queue.txt (6)): it keeps the
// vector one.
template <bool SCALAR = false>
RR_DEV ItemRange load_item_range(const WfDev& wf, uint32_t wave) {
    ItemRange r;
    r.next = wf.wave_items[2 * (size_t)wave];
    r.end = wf.wave_items[2 * (size_t)wave + 1];
    if (SCALAR) r.next = wave_uniform64(r.next), r.end = wave_uniform64(r.end);
    return r;
}
RR_DEV void store_item_range(const WfDev& wf, uint32_t wave, const ItemRange& r) {
    if ((threadIdx.x & 63u) == 0) {
        wf.wave_items[2 * (size_t)wave] = r.next;
        wf.wave_items[2 * (size_t)wave + 1] = r.end;
    }
}

// Refills an empty range from the device-wide counter.  False when the counter has run out -- which the wave then
// remembers (range.end = ITEMS_GONE, kept in WfDev::wave_items between launches): at the end of a frame every batch of
// every wave would otherwise ask the one counter word again, which serves ~90 atomics per microsecond.  Wave-uniform.
RR_DEV bool refill_item_range(const RenderDev& rp, ItemRange& range) {
    unsigned long long first = ITEMS_GONE;
    if (range.end != ITEMS_GONE) {
        if ((threadIdx.x & 63u) == 0) first = atomicAdd(rp.next_item, (unsigned long long)ITEM_RESERVE);
        const uint32_t flo = __builtin_amdgcn_readfirstlane((uint32_t)first);
        const uint32_t fhi = __builtin_amdgcn_readfirstlane((uint32_t)(first >> 32));
        first = ((unsigned long long)fhi << 32) | flo;
    }
    if (first >= rp.total_items) {
        range.next = range.end = ITEMS_GONE;
        return false;
    }
    range.next = first;
    range.end = first + ITEM_RESERVE < rp.total_items ? first + ITEM_RESERVE : rp.total_items;
    return true;
}

// What the kernels that start samples count per lane.
struct SampleCount {
    unsigned long long paths;     // samples started
    unsigned long long direct;    // of them: primary rays that missed the root box (answered here, below)
    unsigned long long deferred;  // of them: primary rays that missed the root box, left to the miss kernel (DEFER)
    uint32_t retired;             // slots that found no further item
    // wave-uniform (scalar) tallies of finish_rays, the pre-test of the rays these kernels make:
    HotTally hot;                 // the hot group's gate and primitive tests
    uint32_t pre_miss;            // bounced rays that missed the root box: a Miss (bvh.rs:394), left to the miss kernel
    uint32_t pre_done;            // rays that entered the root box and none of the four slots of the walk tree's first record:
                                  // their query ends with the hot group's answer, here
    uint32_t pre_root;            // rays put to the first record (all that entered the root box): one record visit each
};

// ---- the pre-test of a new ray by the kernel that made it (scenes with a hot group: layout.h HotGroupDev) ----
// BvhTree::intersect begins every query with the root Node's box (bvh.rs:394).  On a scene with a hot group the default
// walk then owes the ray (i) the hot group -- its gating box, its primitives -- and (ii) the walk of the tree without it,
// which begins with that tree's first record.  (i) and the first record of (ii) are wave-uniform data, and the kernels
// that MAKE rays hold them in full waves (compacted batches), so they do both here, for every ray of a batch at once:
//   * a ray that misses the root box is a Miss: state MISS;
//   * the closest hit so far -- the hot group's -- is written to the slot (RaySlot::t / prim: t1 and "none" without one);
//   * a ray that enters none of the first record's four slots has nothing left to visit: its query is answered, state HIT
//     or MISS, and it never travels through the traversal kernel (six rays in ten on the headline frame);
//   * the others become READY, with the slots they enter in their state byte (ready_state_pre): the traversal kernel takes
//     the closest hit so far from the slot and starts the walk below those slots.
// Every test is the one BvhTree::intersect makes, on the same values; the closest hit is the smallest accepted t, the
// first primitive in depth-first order on exact ties (bvh.rs:62), in whatever order and by whichever kernel the
// primitives are tested.  `got`: the lane holds a ray (o, d) for `slot`; `enters`: it is known to enter the root box.
RR_DEV void finish_rays(bool got, bool enters, bool primary, uint32_t slot, V3 o, V3 d, V3 inv, uint32_t bd, SampleCount& sn) {
    const SceneDev& sc = karg<SceneDev, KA_SC>();
    const WfDev& wf = karg<WfDev, KA_WF>();
    Trav tv;
    tv.inv = inv, tv.best_t = sc.t1, tv.best_prim = 0xffffffffu, tv.cur = TRAV_DONE, tv.sp = 0;
    WorkCount wc{0, 0, 0, 0, 0};
    const bool live = got && enters;
    hot_group_step<false>(sc, o, d, live, tv, wc, sn.hot);
    const uint32_t slots = live ? hot_root_record_entered(sc, o, inv) : 0u;
    const bool walk = slots != 0u;
    sn.pre_root += (uint32_t)__popcll(__ballot(live));
    sn.pre_done += (uint32_t)__popcll(__ballot(live && !walk));
    sn.pre_miss += (uint32_t)__popcll(__ballot(got && !enters && !primary));
    if (got) {
        RaySlot* rs = ray_slot(wf, slot);
        rs->o[0] = o.x, rs->o[1] = o.y, rs->o[2] = o.z;
        rs->d[0] = d.x, rs->d[1] = d.y, rs->d[2] = d.z;
        rs->bd = bd;
        const bool hit = tv.best_prim != 0xffffffffu;
        if (walk || hit) {  // (a MISS says it all: nothing reads t or prim of such a slot)
            rs->t = tv.best_t;
            rs->prim = tv.best_prim;
        }
        wf.state[slot] = walk ? ready_state_pre(slots) : (hit ? WF_HIT : WF_MISS);
    }
}

// A primary ray that misses the box of the BVH's root Node is a Miss before anything else is looked at
// (bvh.rs:394), so its sample is the background (lib.rs:555) -- finished here, in registers, instead of
// sending the ray through the traversal and miss kernels for the same answer: the lane goes on to the
// item's next sample, and to the next item, until it holds a ray that enters the root box.  From the
// reference's obj_scene camera that is every seventh primary ray (the sky above the floor's far edge).
// DEFER (the hit kernel): such a ray is written to its slot like any other and the slot left in state MISS -- the miss
// kernel runs behind the hit kernel in the same round and finishes the sample there.  Answering it here means a texel
// fetch that the lane waits for with everything the wave has stored so far still in flight (a wave's memory operations
// complete in order), in nearly every batch (a seventh of the new samples, ~25 of them per batch), and another turn of
// the loop below: the hit kernel has bandwidth to spare and no latency to spare; the miss kernel does this work anyway.
// CARRY (the hit kernel): lanes whose path goes on bring their bounced ray (co, cd, cbd) along; it is written out at the
// end together with the new samples' primary rays, so that the pre-test above runs once, on a full wave.
// The rays a batch ends up with: next_sample makes them, emit_rays pre-tests and writes them.  The two are separate calls so
// that a kernel can issue its look-ahead loads between them: emit_rays is several thousand cycles of arithmetic on scalar
// operands without a call or a wait in it -- the one stretch of these kernels a request can be in flight behind.
struct NewRays {
    bool got, enters, carry;  // the lane holds a ray for its slot; it is known to enter the root box; it is a bounced ray
    V3 o, d, inv;
    uint32_t bd;
};

template <bool COMPACT, bool DEFER = false>
RR_DEV NewRays next_sample(bool want, uint32_t slot, const ItemRegs& ir, bool acc_dirty, ItemRange& range,
                           SampleCount& sn, bool carry = false, V3 co = V3{0.0, 0.0, 0.0}, V3 cd = V3{0.0, 0.0, 1.0},
                           uint32_t cbd = 0u) {
    // (the kernel's arguments: views made where a region needs them -- karg above)
    const bool pre = karg<SceneDev, KA_SC>().hot != nullptr;  // wave-uniform: the rays made here are pre-tested (finish_rays)
    bool todo = want;  // lanes still without a ray for their slot
    bool has_item = want && ir.has_item();
    uint32_t item = ir.item, s_cur = ir.cursor(), s_end = ir.s_end;
    uint32_t row = ir.pix >> 16, col = ir.pix & 0xffffu;
    double acc0 = ir.acc[0], acc1 = ir.acc[1], acc2 = ir.acc[2];
    bool fresh = false;            // the slot's item record has to be written in full
    bool acc_write = acc_dirty;    // the item goes on with a sum the slot does not hold yet
    // the ray the lane ends up with, written out behind the loop
    bool got = carry, enters = true;
    V3 o = co, d = cd, inv = mk(0.0, 0.0, 0.0);
    uint32_t bd = cbd;
    for (;;) {
        const RenderDev& rp = karg<RenderDev, KA_RP>();
        const WfDev& wf = karg<WfDev, KA_WF>();
        const uint32_t cam_W = karg<CameraDev, KA_CAM>().W, cam_H = karg<CameraDev, KA_CAM>().H;
        // an item whose samples are all done is written out (its sum goes to the resolve kernel)
        if (todo && has_item && s_cur >= s_end) {
            double* dst = rp.partial + (size_t)item * 3;
            dst[0] = acc0;
            dst[1] = acc1;
            dst[2] = acc2;
            has_item = false;
        }
        // slots without an item take the next ones from the wave's reserved range (ballot + rank),
        // which is refilled from the device-wide counter
        bool need = todo && !has_item;
        bool dead = false;
        unsigned long long need_mask = __ballot(need);
        while (need_mask != 0ull) {
            if (range.next >= range.end && !refill_item_range(rp, range)) {  // wave-uniform
                if (need) dead = true;  // the counter has run out: these slots are done
                break;
            }
            const uint32_t avail = (uint32_t)(range.end - range.next);
            const uint32_t rank = lanes_below(need_mask);
            if (need && rank < avail) {
                item = (uint32_t)(range.next + rank);
                uint32_t s_begin;
                item_geometry(rp, item, row, col, s_begin, s_end);
                s_cur = s_begin;
                if (row >= cam_H || col >= cam_W || rp.max_bounces == 0u) {
                    // padding pixel of an edge tile (never read) or radiance() with an empty loop: zeros
                    double* dst = rp.partial + (size_t)item * 3;
                    dst[0] = dst[1] = dst[2] = 0.0;
                    if (row < cam_H && col < cam_W) sn.paths += s_end - s_begin;
                } else {
                    has_item = true;
                    fresh = true;
                    acc0 = acc1 = acc2 = 0.0;
                    need = false;
                }
            }
            const uint32_t wanted = (uint32_t)__popcll(need_mask);
            range.next += wanted < avail ? wanted : avail;
            need_mask = __ballot(need);
        }
        if (todo && dead) {
            tail_slot(wf, slot)->s_cur = 0;
            wf.state[slot] = WF_DEAD;
            sn.retired++;
            todo = false;
        }
        if (todo && has_item) {
            // start the slot's next sample (main.rs:68-76)
            const CameraDev& cam = karg<CameraDev, KA_CAM>();
            const SceneDev& sc = karg<SceneDev, KA_SC>();
            Rng rng;
            rng.key = rr_path_key(rp.seed, (uint64_t)row * cam.W + col, (uint64_t)s_cur);
            rng.draw = 0;
            // image origin is upper left, camera origin lower right (main.rs:74-75)
            primary_ray(cam, cam.H - row, cam.W - col, rng, o, d);
            sn.paths++;
            s_cur++;
            // (DEFER with the pre-test: 1 / d and the root box wait for the end, where the bounced rays need them too)
            if (!(DEFER && pre)) {
                inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
                enters = root_box_hit(sc, o, inv);
            }
            if (!enters && !DEFER) {
                // radiance() with the first query a Miss: light 0 + throughput 1 * background (lib.rs:522-523, :555)
                const V3 result = v_add(mk(0.0, 0.0, 0.0), v_mul(mk(1.0, 1.0, 1.0), background(sc, d)));
                acc0 += result.x;
                acc1 += result.y;
                acc2 += result.z;
                acc_write = true;
                sn.direct++;
                enters = true;
            } else {
                bd = 1u | (rng.draw << 16);  // first query; throughput 1 and light 0 are implied
                TailSlot* t = tail_slot(wf, slot);
                t->s_cur = s_cur | SLOT_ITEM_BIT;  // a new path: no light yet
                if (fresh) {
                    t->item = item;
                    t->s_end = s_end;
                    t->pix = row << 16 | col;  // both below 2^16 (checked at launch)
                }
                if (fresh || acc_write) t->acc[0] = acc0, t->acc[1] = acc1, t->acc[2] = acc2;
                got = true;
                todo = false;
            }
        }
        if (__ballot(todo) == 0ull) break;
    }
    return NewRays{got, enters, carry, o, d, inv, bd};
}

// ---- the rays: bounced ones brought along (carry) and the new samples' primary rays
template <bool DEFER>
RR_DEV void emit_rays(uint32_t slot, const NewRays& nr, SampleCount& sn) {
    const bool got = nr.got, carry = nr.carry;
    bool enters = nr.enters;
    const V3 o = nr.o, d = nr.d;
    V3 inv = nr.inv;
    const uint32_t bd = nr.bd;
    const bool pre = karg<SceneDev, KA_SC>().hot != nullptr;  // wave-uniform: the rays are pre-tested (finish_rays)
    if (__ballot(got) == 0ull) return;
    const bool primary = got && !carry;
    if (pre) {
        if (DEFER) {  // every ray of the batch: 1 / d and the root Node's box (bvh.rs:394)
            inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
            enters = root_box_hit(karg<SceneDev, KA_SC>(), o, inv);
        }
        if (primary && !enters) sn.deferred++;  // (only with DEFER: elsewhere such a sample was finished above)
        finish_rays(got, enters, primary, slot, o, d, inv, bd, sn);
    } else if (got) {
        const WfDev& wf = karg<WfDev, KA_WF>();
        RaySlot* rs = ray_slot(wf, slot);
        rs->o[0] = o.x, rs->o[1] = o.y, rs->o[2] = o.z;
        rs->d[0] = d.x, rs->d[1] = d.y, rs->d[2] = d.z;
        rs->bd = bd;
        if (primary && !enters) {  // DEFER
            wf.state[slot] = WF_MISS;  // the root box test was this ray's query (bvh.rs:394): the miss kernel's
            sn.deferred++;
        } else {
            wf.state[slot] = ready_state(d);  // (a bounced ray's root box is the traversal kernel's to test)
        }
    }
}

// rays and samples the sample-starting kernels account for, one atomic each per wave
RR_DEV void store_sample_count(const SceneDev& sc, const RenderDev& rp, const WfDev& wf, const SampleCount& sn) {
    wave_atomic_add(&rp.counters->paths, sn.paths);
    const unsigned long long direct = wave_sum(sn.direct), deferred = wave_sum(sn.deferred);
    const unsigned long long pre = (unsigned long long)sn.pre_miss + sn.pre_done;  // queries answered by finish_rays
    if ((threadIdx.x & 63u) == 0 && (direct | deferred | pre)) {
        atomicAdd(&rp.counters->rays, direct + deferred + pre);
        if (direct) atomicAdd(&rp.counters->escaped_paths, direct);  // (the miss kernel counts the deferred ones' escape)
        if (direct | deferred) atomicAdd(&rp.counters->direct_rays, direct + deferred);
    }
    if (rp.count_work && sc.hot != nullptr && (threadIdx.x & 63u) == 0) {  // what the pre-test did, for the work counters
        Counters* c = rp.counters;
        const HotPtr h = hot_ptr(sc);
        const unsigned long long e = sn.hot.entered;
        atomicAdd(&c->pre_rays, pre);
        atomicAdd(&c->interior_visits, (unsigned long long)sn.pre_root);  // the first record of the walk tree: tested here for every ray, never again
        atomicAdd(&c->pre_root_records, (unsigned long long)sn.pre_root);
        atomicAdd(&c->hot_lane, (unsigned long long)sn.hot.owed);
        atomicAdd(&c->hot_prim_tests, e * h->count);
        atomicAdd(&c->hot_tri_divided, (unsigned long long)sn.hot.divided);
        if (h->n_tri) atomicAdd(&c->tri_tests, e * h->n_tri);
        if (h->n_sphere) atomicAdd(&c->sphere_tests, e * h->n_sphere);
        if (h->n_plane) atomicAdd(&c->plane_tests, e * h->n_plane);
    }
    const uint32_t r = (uint32_t)wave_sum(sn.retired);
    if ((threadIdx.x & 63u) == 0 && r) atomicSub(&wf.ctl->live_slots, r);
}

// Initial fill of the pool (every live slot starts IDLE).
template <bool COMPACT>
__global__ void __launch_bounds__(256) wf_gen_kernel(SceneDev sc, CameraDev cam, RenderDev rp, WfDev wf) {
    __shared__ uint16_t lists[4][WINDOW];
    const uint32_t lane = threadIdx.x & 63u;
    uint16_t* list = lists[threadIdx.x >> 6];
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n_windows = wf.np / WINDOW;
    SampleCount sn{0, 0, 0, 0, HotTally{0, 0, 0}, 0, 0, 0};
    ItemRange range = load_item_range(wf, wave);
    for (uint32_t win = wave; win < n_windows; win += n_waves) {
        const uint32_t count = compact_window(wf, win, WF_IDLE, list);
        for (uint32_t k = 0; k < count; k += 64u) {
            const bool valid = k + lane < count;
            const uint32_t slot = win * WINDOW + (valid ? (uint32_t)list[k + lane] : 0u);
            // an IDLE slot has no item BY DEFINITION: wf_init_kernel writes the state bytes only, and the slot's line still
            // holds whatever the previous frame left there -- so nothing of it may be read (tests/test_gpu_render.py
            // ::test_two_different_frames_back_to_back_on_one_scene); the item starts from zeros made here
            ItemRegs ir;
            ir.acc[0] = ir.acc[1] = ir.acc[2] = 0.0;
            ir.item = ir.s_end = ir.pix = ir.word = 0u;
            const NewRays nr = next_sample<COMPACT>(valid, slot, ir, false, range, sn);
            emit_rays<false>(slot, nr, sn);
        }
    }
    store_item_range(wf, wave, range);
    store_sample_count(sc, rp, wf, sn);
}

// ------------------------------------------------------------------- trav

// Scheduling thresholds (RenderDev::refill_min / leaf_min): with fewer traversing
// lanes than refill_min the wave retires its finished queries and takes new rays
// from its window list; a leaf phase runs once leaf_min lanes stand on a leaf (or
// none is on an interior record).
//
// EXACT (the default walk: nothing is culled, so the order in which a ray's leaf groups are tested changes nothing): a lane
// does not stand on a leaf -- it sets the group aside on a queue of its own in LDS and walks on (device_path.h
// trav_interior_step_defer), so it takes part in interior phases while it has a record to visit AND in leaf phases while a
// group waits.  A leaf phase runs once leaf_min lanes have a group waiting, once leaf_wait lanes can do nothing else, or
// when no lane has a record to visit.  Interior phases at 0.77 of the lanes instead of 0.59, a quarter fewer of them
// (profiles/r06_leaf_queue.txt; priced beforehand with scripts/sim/walk_sched_sim.py).

// PRE (the default walk on a scene with a hot group, layout.h HotGroupDev): the rays come pre-tested by the kernel that
// made them (finish_rays above) -- they are known to enter the root box and at least one slot of the tree's first record,
// and the slot holds the closest hit so far (the hot group's): the walk starts from that, at the first record.
template <bool COMPACT, bool COUNT, bool EXACT, bool PRE>
__global__ void __launch_bounds__(256, 5) wf_trav_kernel(SceneDev sc, RenderDev rp, WfDev wf) {
    static_assert(EXACT || !PRE, "pre-tested rays are the default walk's");
    extern __shared__ uint32_t lds_dyn[];
    WfCtl* ctl = wf.ctl;
    if (ctl->live_slots == 0u) return;  // a round enqueued behind the frame's last one (abi.cpp look-behind)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    // dynamic LDS: 4 stacks of (stack_lds + 1 spare + leafq queued leaf groups) x 64 words, 4 window lists of WINDOW uint16,
    // hot_records wide records
    const uint32_t stack_words = (sc.stack_lds + 1u + sc.leafq) * 64u;
    const LaneStack stack{lds_dyn + (size_t)wave * stack_words + lane,
                          wf.stack_spill + ((size_t)blockIdx.x * 256u + threadIdx.x), sc.stack_lds, wf.trav_threads};
    uint16_t* list = reinterpret_cast<uint16_t*>(lds_dyn + 4u * (size_t)stack_words) + wave * WINDOW;
    uint4* hot_lds = reinterpret_cast<uint4*>(lds_dyn + 4u * (size_t)stack_words + 4u * WINDOW / 2u);
    {
        constexpr uint32_t G = COMPACT ? 8u : 16u;  // 16-byte granules per record
        const uint4* src = reinterpret_cast<const uint4*>(sc.nodes);
        for (uint32_t i = threadIdx.x; i < sc.hot_records * G; i += 256u)
            hot_lds[(i / G) * HotNodes::stride<COMPACT>() + i % G] = src[i];
        __syncthreads();
    }
    const HotNodes hot{hot_lds, sc.hot_records};
    const unsigned long long lanemask_lt = (1ull << lane) - 1ull;
    // PRE: the references of the first record's four slots (wave-uniform: scalar registers)
    uint32_t root_ref0 = 0, root_ref1 = 0, root_ref2 = 0, root_ref3 = 0;
    if (PRE) {
        const HotPtr h = hot_ptr(sc);
        root_ref0 = h->root_ref[0], root_ref1 = h->root_ref[1], root_ref2 = h->root_ref[2], root_ref3 = h->root_ref[3];
    }
    // which of them are leaf groups (an unused slot is never entered)
    const uint32_t root_leaves = (root_ref0 >= REF_LEAF_BASE ? 1u : 0u) | (root_ref1 >= REF_LEAF_BASE ? 2u : 0u) |
                                 (root_ref2 >= REF_LEAF_BASE ? 4u : 0u) | (root_ref3 >= REF_LEAF_BASE ? 8u : 0u);

    const uint32_t n_windows = wf.np / WINDOW;
    // Windows of the pool are handed out in two ways.  The first rp.static_windows windows are
    // dealt round robin: wave g takes g, g + n_waves, ... and asks nobody.  The rest go to whoever
    // runs dry first, through one atomic cursor (WfCtl::next_window), which evens out the
    // waves' finishing times.  (Round 1: all of the pool through the cursor cost a fifth of the waves'
    // time; with persistent waves taking ~43 windows each per launch the cursor serves 25 atomics per
    // microsecond of the ~90 it can, and a dealt share of 5 % is as good as 50 %: profiles/r04_xcd_streams.txt.)
    // Once most slots have run out of work (the tail of a frame, long on a small tile share) the
    // cursor's atomics are all a launch would wait for, and balance no longer matters: deal everything.
    const uint32_t n_waves = gridDim.x * 4u;
    const uint32_t static_windows = ctl->live_slots < wf.np / 4u ? n_windows : rp.static_windows;
    uint32_t static_next = blockIdx.x * 4u + wave;       // wave-uniform
    uint32_t list_pos = 0, list_len = 0, list_base = 0;  // wave-uniform
    bool no_more = false;                                // wave-uniform: window cursor ran off the end

    bool active = false, pending = false;
    uint32_t slot = 0;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    Trav tv;
    tv.inv = mk(0, 0, 0), tv.best_t = 0, tv.best_prim = 0xffffffffu, tv.cur = TRAV_DONE, tv.sp = 0;
    WorkCount wc{0, 0, 0, 0, 0};
    unsigned long long n_rays = 0;
    unsigned long long u_int_wave = 0, u_int_lane = 0, u_leaf_wave = 0, u_leaf_lane = 0;
    unsigned long long tk_int = 0, tk_leaf = 0, tk_refill = 0, tk_last = COUNT ? clock64() : 0ull;

    for (;;) {
        // EXACT (the default walk): a lane's leaf groups wait in its queue while it walks on (device_path.h
        // trav_interior_step_defer) -- it is at_int whenever it has a record to visit (and room for what the record may
        // queue), at_leaf whenever a group waits; both at once is the rule.  Else: the lane stands on ONE reference.
        const bool at_int = EXACT ? (active && tv.cur != TRAV_DONE && defer_has_room(tv)) : (active && trav_at_interior(tv));
        const bool at_leaf = EXACT ? (active && defer_has_leaf(tv)) : (active && !trav_at_interior(tv));
        const int n_int = __popcll(__ballot(at_int));
        const int n_leaf = __popcll(__ballot(at_leaf));
        const int n_work = EXACT ? __popcll(__ballot(active)) : n_int + n_leaf;
        if ((n_work < (int)rp.refill_min && !no_more) || n_work == 0) {
            // ---- retire finished queries: result and new state to the slot
            if (pending) {
                const bool hit = tv.best_prim != 0xffffffffu;
                if (hit) {  // (a MISS says it all: nothing reads t or prim of such a slot, and its line stays clean)
                    RaySlot* rs = ray_slot(wf, slot);
                    rs->t = tv.best_t;
                    rs->prim = tv.best_prim;
                }
                wf.state[slot] = hit ? WF_HIT : WF_MISS;
                pending = false;
            }
            if (no_more) break;  // only reached with no query in flight
            // ---- idle lanes take rays from the wave's window list
            bool need = !active;
            unsigned long long need_mask = __ballot(need);
            while (need_mask != 0ull) {
                if (list_pos >= list_len) {
                    uint32_t w = static_next;
                    if (w < static_windows) {
                        static_next += n_waves;
                    } else {
                        if (lane == 0) w = static_windows + atomicAdd(&ctl->next_window, 1u);
                        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
                    }
                    if (w >= n_windows) {
                        no_more = true;
                        break;
                    }
                    list_base = w * WINDOW;
                    list_len = PRE ? compact_window_ready_pre(wf, w, list) : compact_window_ready(wf, w, list);
                    list_pos = 0;
                    continue;
                }
                const uint32_t avail = list_len - list_pos;
                const uint32_t rank = (uint32_t)__popcll(need_mask & lanemask_lt);
                if (need && rank < avail) {
                    const uint32_t entry = (uint32_t)list[list_pos + rank];
                    slot = list_base + (PRE ? (entry & 511u) : entry);
                    const RaySlot* rs = ray_slot(wf, slot);
                    o = mk(rs->o[0], rs->o[1], rs->o[2]);
                    d = mk(rs->d[0], rs->d[1], rs->d[2]);
                    n_rays++;
                    if (PRE) {
                        tv.inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
                        tv.best_t = rs->t;
                        tv.best_prim = rs->prim;
                        // the slots of the first record this ray enters (finish_rays tested it): the first becomes the lane's
                        // reference, the others wait on its stack so that they come off it in slot order, as
                        // trav_interior_step would have left them
                        const uint32_t m = entry >> 9;
                        // (the references are scalars; taken through an empty asm here so that their copies into vector
                        // registers for the stores below are made here and not kept -- spilled -- across the whole walk)
                        uint32_t r1 = root_ref1, r2 = root_ref2, r3 = root_ref3;
                        asm volatile("" : "+s"(r1), "+s"(r2), "+s"(r3));
                        // leaf slots among them go to the lane's queue; of the interior ones the first becomes the lane's
                        // record, the others wait on its stack so that they come off it in slot order
                        const uint32_t mi = m & ~root_leaves, ml = m & root_leaves;  // (root_leaves: scalar)
                        const int n = (int)__popc(mi);
                        const uint32_t low = mi & (0u - mi);
                        tv.cur = mi == 0u ? TRAV_DONE : (low & 1u) ? root_ref0 : (low & 2u) ? r1 : (low & 4u) ? r2 : r3;
                        tv.sp = n > 0 ? n - 1 : 0;
                        if ((mi & 2u) && low != 2u) stack.put(n - 1 - (int)__popc(mi & 1u), r1);  // (put: in LDS, or in the lane's HBM strip
                        if ((mi & 4u) && low != 4u) stack.put(n - 1 - (int)__popc(mi & 3u), r2);  //  when fewer than three entries are kept in
                        if ((mi & 8u) && low != 8u) stack.put(n - 1 - (int)__popc(mi & 7u), r3);  //  LDS; a branch-free variant measured the same)
                        if (ml != 0u) {  // (not on the obj scenes: their first record's slots are the mesh's quarters)
                            if (ml & 1u) defer_take_ref(stack, tv, root_ref0);
                            if (ml & 2u) defer_take_ref(stack, tv, r1);
                            if (ml & 4u) defer_take_ref(stack, tv, r2);
                            if (ml & 8u) defer_take_ref(stack, tv, r3);
                        }
                        active = true;
                    } else {
                        trav_init(sc, o, d, tv);
                        if (tv.cur == TRAV_DONE) {
                            pending = true;  // missed the root box: retired at the next refill
                        } else {
                            active = true;
                            if (EXACT && tv.cur >= REF_LEAF_BASE) {  // a tree that is one leaf group
                                const uint32_t ref = tv.cur;
                                tv.cur = TRAV_DONE;
                                defer_take_ref(stack, tv, ref);
                            }
                        }
                    }
                    need = false;
                }
                const uint32_t wanted = (uint32_t)__popcll(need_mask);
                list_pos += wanted < avail ? wanted : avail;
                need_mask = __ballot(need);
            }
            if (COUNT) {
                const unsigned long long now = clock64();
                tk_refill += now - tk_last, tk_last = now;
            }
            if (__ballot(active || pending) == 0ull && no_more) break;
            continue;
        }
        // EXACT: a leaf phase also once leaf_wait lanes can do nothing else (they have groups waiting and no record to
        // visit, or no room): those lanes idle through interior phases, and a lane is retired only when its queue is empty
        const bool leaf_phase = n_leaf >= (int)rp.leaf_min || n_int == 0 ||
                                (EXACT && __popcll(__ballot(at_leaf && !at_int)) >= (int)rp.leaf_wait);
        if (leaf_phase) {
            // ---- leaf phase: every lane standing on a leaf (EXACT: with a group waiting) tests its primitives
            if (COUNT) u_leaf_wave += 1, u_leaf_lane += at_leaf ? 1 : 0;
            if (at_leaf) {
                if (EXACT) {
                    trav_leaf_step_defer<COMPACT, COUNT>(sc, o, d, stack, tv, wc);
                    if (defer_finished(tv)) active = false, pending = true;
                } else {
                    trav_leaf_step<COMPACT, COUNT>(sc, o, d, stack, tv, wc);
                    if (tv.cur == TRAV_DONE) active = false, pending = true;
                }
            }
            if (COUNT) {
                const unsigned long long now = clock64();
                tk_leaf += now - tk_last, tk_last = now;
            }
        } else {
            // ---- interior phase: one record for every lane standing on one
            if (COUNT) u_int_wave += 1, u_int_lane += at_int ? 1 : 0;
            if (at_int) {
                if (EXACT) {
                    trav_interior_step_defer<COMPACT, COUNT>(sc, o, stack, hot, tv, wc);
                    if (defer_finished(tv)) active = false, pending = true;
                } else {
                    trav_interior_step<COMPACT, COUNT, false>(sc, o, stack, hot, tv, wc);
                    if (tv.cur == TRAV_DONE) active = false, pending = true;
                }
            }
            if (COUNT) {
                const unsigned long long now = clock64();
                tk_int += now - tk_last, tk_last = now;
            }
        }
    }

    Counters* c = rp.counters;
    wave_atomic_add(&c->rays, n_rays);
    if (COUNT) {
        wave_atomic_add(&c->interior_visits, wc.interior);
        wave_atomic_add(&c->tri_tests, wc.tri);
        wave_atomic_add(&c->sphere_tests, wc.sphere);
        wave_atomic_add(&c->plane_tests, wc.plane);
        wave_atomic_add(&c->step_wave, u_int_wave), wave_atomic_add(&c->step_lane, u_int_lane);
        wave_atomic_add(&c->inner_wave, u_leaf_lane), wave_atomic_add(&c->leaf_wave, u_leaf_wave);
        if (lane == 0) {
            atomicAdd(&c->interior_ticks, tk_int), atomicAdd(&c->leaf_ticks, tk_leaf);
            atomicAdd(&c->refill_ticks, tk_refill);
        }
    }
}

// -------------------------------------------------------------------- hit

// What a batch reads from its slots before anything can be computed.
struct HitIn {
    uint32_t slot;
    bool valid;
    V3 o, d, thr, light;  // light: the side array's entry, requested with the slot only when EAGER (else fetched on demand)
    double t;
    uint32_t prim, bd;
    ItemRegs ir;
};

// A path's light: +0 unless the slot says it is in the side array (wavefront.h PathSlot).
RR_DEV V3 load_light(const WfDev& wf, uint32_t slot) {
    const double* l = light_slot(wf, slot);
    return mk(l[0], l[1], l[2]);
}
RR_DEV bool light_is_plus_zero(V3 light) {  // bitwise: -0 and NaN are not
    return (rr_f64_bits(light.x) | rr_f64_bits(light.y) | rr_f64_bits(light.z)) == 0ull;
}

template <bool EAGER>
RR_DEV void load_hit_in(const WfDev& wf, HitIn& h) {  // idle lanes read slot 0: harmless
    const RaySlot* rs = ray_slot(wf, h.slot);
    h.o = mk(rs->o[0], rs->o[1], rs->o[2]);
    h.d = mk(rs->d[0], rs->d[1], rs->d[2]);
    h.t = rs->t;
    h.prim = rs->prim;
    h.bd = rs->bd;
    const TailSlot* t = tail_slot(wf, h.slot);
    h.thr = mk(t->thr[0], t->thr[1], t->thr[2]);
    h.light = EAGER ? load_light(wf, h.slot) : mk(0.0, 0.0, 0.0);
    h.ir = load_item(wf, h.slot);
}

// Built for two workgroups per CU: 256 registers, batch b + 1 requested while batch b is computed (below).
template <bool COMPACT, bool EAGER>
__global__ void __launch_bounds__(256, 3) wf_hit_kernel(SceneDev sc, CameraDev cam, RenderDev rp, WfDev wf) {
    __shared__ uint32_t lists[4][FEED_LIST];
    // The surface row is the third dependent fetch of a hit (slot -> primitive -> surface);
    // scenes have a handful of rows, so the first HIT_SURFACES_LDS of them wait in LDS.
    __shared__ SurfaceDev s_surf[HIT_SURFACES_LDS];
    if (wf.ctl->live_slots == 0u) return;
    const uint32_t n_surf_lds = sc.n_surfaces < HIT_SURFACES_LDS ? sc.n_surfaces : HIT_SURFACES_LDS;
    for (uint32_t i = threadIdx.x; i < n_surf_lds * (uint32_t)(sizeof(SurfaceDev) / 4); i += 256u)
        reinterpret_cast<uint32_t*>(s_surf)[i] = reinterpret_cast<const uint32_t*>(sc.surfaces)[i];
    __syncthreads();
    uint32_t* list = lists[threadIdx.x >> 6];
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    SampleCount sn{0, 0, 0, 0, HotTally{0, 0, 0}, 0, 0, 0};
    if (blockIdx.x == 0 && threadIdx.x == 0) wf.ctl->next_window = 0;  // the traversal kernel's window cursor
    ItemRange range = load_item_range<true>(wf, wave);
    BatchFeed feed;
    feed_init(feed, wf, wave, n_waves, WF_HIT, list);
    // Three fetches per hit depend on each other: slot -> primitive record -> surface row.  The slot records of
    // batch b + 1 are requested before batch b is computed; the primitive records of batch b + 1 as soon as batch
    // b's arithmetic is done -- before batch b's stores, so that the wait for them is not behind the stores
    // (memory operations of a wave complete in order); the surface rows wait in LDS.
    HitIn cur;
    bool have = feed_next(feed, wf, cur.slot, cur.valid);
    PrimRec<COMPACT> rec_cur;
    if (have) {
        load_hit_in<EAGER>(wf, cur);
        rec_cur = load_prim<COMPACT>(sc.prims, cur.valid ? cur.prim : 0u);
    }
    RR_TICKS_BEGIN(7);
    while (have) {
        HitIn nxt;
        bool have_next = false;
        PrimRec<COMPACT> rec_nxt;
        {
            const uint32_t slot = cur.slot;
            const bool valid = cur.valid;
            bool ended = false, goes_on = false, acc_changed = false;
            uint32_t hit_sid = 8u;
            ItemRegs ir = cur.ir;
            V3 thr = mk(1.0, 1.0, 1.0), light = mk(0.0, 0.0, 0.0), position = mk(0.0, 0.0, 0.0), dir = mk(0.0, 0.0, 1.0);
            uint32_t bd_next = 0;
            if (valid) {
                RR_TICK_LOADS_ARRIVED(6)
                const SceneDev& sc = karg<SceneDev, KA_SC>();
                const RenderDev& rp = karg<RenderDev, KA_RP>();
                const WfDev& wf = karg<WfDev, KA_WF>();
                const V3 o = cur.o;
                const V3 d = cur.d;
                const double t = cur.t;
                const uint32_t bounce = cur.bd & 0xffffu;
                // loaded unconditionally, beside the ray, and ignored while bounce == 1 (lib.rs:522-523)
                thr = bounce > 1u ? cur.thr : mk(1.0, 1.0, 1.0);
                light = mk(0.0, 0.0, 0.0);
                if (bounce > 1u && ir.has_light()) light = EAGER ? cur.light : load_light(wf, slot);
                Rng rng{sample_key(rp, karg<CameraDev, KA_CAM>(), ir), cur.bd >> 16};
                // lib.rs:528-551
                const PrimRec<COMPACT>& rec = rec_cur;
                position = v_add(o, v_scale(d, t));
                const V3 normal = prim_normal<COMPACT>(rec, position);
                const V3 view = v_unit(v_scale(d, -1.0));
                const uint32_t sid = rec.tag() >> 8;
                hit_sid = sid < 7u ? sid : 7u;
                const SurfaceDev* surf = sid < n_surf_lds ? &s_surf[sid] : sc.surfaces + sid;
                const Scatter ev = material_evaluate(surf, normal, view, rng);
                if (ev.scatter) {
                    light = v_add(light, v_mul(thr, mk(surf->emit[0], surf->emit[1], surf->emit[2])));
                    thr = v_mul(thr, ev.color);
                    const double p = rr_max(rr_max(thr.x, thr.y), thr.z);
                    if (rng.next() > p) {
                        ended = true;
                    } else if (bounce >= rp.max_bounces) {  // loop bound of lib.rs:525; lib.rs:559
                        ended = true;
                    } else {
                        thr = mk(thr.x / p, thr.y / p, thr.z / p);  // DivAssign, vecmath.rs:708-714
                        dir = ev.dir;
                        bd_next = (bounce + 1u) | (rng.draw << 16);
                        goes_on = true;
                    }
                } else {
                    ended = true;  // lib.rs:550
                }
                if (ended) {  // radiance() returns `light`; main.rs:69 adds it to the pixel
                    ir.acc[0] += light.x;
                    ir.acc[1] += light.y;
                    ir.acc[2] += light.z;
                    // a sum is never -0 (it starts at +0), so adding +0 leaves its bits alone, and the slot's copy stands
                    acc_changed = !light_is_plus_zero(light);
                }
            }
            RR_TICK(1)
            if (goes_on) {
                const WfDev& wf = karg<WfDev, KA_WF>();
                // (the bounced ray itself -- origin, direction, bounce | draw, state -- is written by next_sample below, together
                // with the new samples' primary rays: one pre-test for the whole batch, finish_rays)
                TailSlot* lt = tail_slot(wf, slot);
                lt->thr[0] = thr.x, lt->thr[1] = thr.y, lt->thr[2] = thr.z;
                const bool keep_light = !light_is_plus_zero(light);
                if (keep_light) {
                    double* l = light_slot(wf, slot);
                    l[0] = light.x, l[1] = light.y, l[2] = light.z;
                }
                // (the cursor word changes only when the path gets or loses its light: mostly it is left alone, and with
                // it the 32-byte sector it lies in -- stores cost these kernels more than anything they compute)
                if (keep_light != ir.has_light()) lt->s_cur = ir.cursor() | (keep_light ? SLOT_LIGHT_BIT : 0u) | SLOT_ITEM_BIT;
            }
            if (karg<RenderDev, KA_RP>().count_work) {  // wave-uniform; what the queries found, per surface row (bench.py: ray shares)
                const RenderDev& rp = karg<RenderDev, KA_RP>();
#pragma unroll
                for (uint32_t k = 0; k < 8u; k++) {
                    const uint32_t c = (uint32_t)__popcll(__ballot(hit_sid == k));
                    if ((threadIdx.x & 63u) == 0 && c) atomicAdd(&rp.counters->surface_hits[k], (unsigned long long)c);
                }
            }
            RR_TICK(3)
            // ---- batch b + 1's slot records are requested HERE, behind Material::evaluate: every call of an elementary function
            // (and the dispatch on the material's kind) waits for all the wave's requests in flight, so a request made ahead of
            // them -- where rounds 1-5 made it -- was waited for a few hundred instructions later: a memory latency at the top
            // of every batch, the look-ahead undone.  What follows -- next_sample, emit_rays -- is ten thousand cycles without a
            // call in it.  (Behind this batch's own stores rather than ahead of them: hit -1.3 %, measured both ways round.)
            {
                const WfDev& wfv = karg<WfDev, KA_WF>();  // (the kernel's arguments are re-read where they are used: karg)
                have_next = feed_next(feed, wfv, nxt.slot, nxt.valid);
                if (!have_next) nxt.slot = 0u, nxt.valid = false;
                RR_TICK(5)
                // (unconditional -- behind a wave's last batch every lane reads slot 0 once: under `if (have_next)` the loaded
                // registers are copied into the loop's own at the end of the conditional block, i.e. waited for at once)
                load_hit_in<EAGER>(wfv, nxt);
            }
            RR_TICK(0)
            const NewRays nr = next_sample<COMPACT, true>(ended, slot, ir, acc_changed, range, sn, goes_on, position, dir, bd_next);
            RR_TICK(4)
            // batch b + 1's slot records have had next_sample's time to arrive: its primitive records (the second of the three
            // dependent fetches of a hit), which have emit_rays' time
            rec_nxt = load_prim<COMPACT>(karg<SceneDev, KA_SC>().prims, nxt.valid ? nxt.prim : 0u);
            RR_TICK(2)
            emit_rays<true>(slot, nr, sn);
            RR_TICK(4)
        }
        cur = nxt;
        rec_cur = rec_nxt;
        have = have_next;
        RR_TICKS_BATCH();
    }
    RR_TICKS_END_HIT(rp)
    store_item_range(wf, wave, range);
    store_sample_count(sc, rp, wf, sn);
}

// ------------------------------------------------------------------- miss

struct MissIn {
    uint32_t slot;
    bool valid;
    V3 d, thr, light;
    uint32_t bd;
    ItemRegs ir;
};

template <bool EAGER>
RR_DEV void load_miss_in(const WfDev& wf, MissIn& m) {  // idle lanes read slot 0: harmless
    const RaySlot* rs = ray_slot(wf, m.slot);
    m.d = mk(rs->d[0], rs->d[1], rs->d[2]);
    m.bd = rs->bd;
    m.ir = load_item(wf, m.slot);
    const TailSlot* t = tail_slot(wf, m.slot);
    m.thr = mk(t->thr[0], t->thr[1], t->thr[2]);
    m.light = EAGER ? load_light(wf, m.slot) : mk(0.0, 0.0, 0.0);
}

template <bool COMPACT, bool EAGER>
__global__ void __launch_bounds__(256, 3) wf_miss_kernel(SceneDev sc, CameraDev cam, RenderDev rp, WfDev wf) {
    __shared__ uint32_t lists[4][FEED_LIST];
    if (wf.ctl->live_slots == 0u) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* list = lists[threadIdx.x >> 6];
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    unsigned long long n_escaped = 0;
    SampleCount sn{0, 0, 0, 0, HotTally{0, 0, 0}, 0, 0, 0};
    ItemRange range = load_item_range(wf, wave);
    BatchFeed feed;
    feed_init(feed, wf, wave, n_waves, WF_MISS, list);
    MissIn cur;
    bool have = feed_next(feed, wf, cur.slot, cur.valid);
    if (have) load_miss_in<EAGER>(wf, cur);
    RR_TICKS_BEGIN(3);
    while (have) {
        // The next batch's slot records are requested at the loop's top.  (Behind next_sample, where only emit_rays' call-free
        // stretch follows -- the hit kernel's order -- this kernel lost 5 %: it is bound by its 1.0 TB of random lines, and
        // what it needs is requests in flight for as long as possible, whoever waits for them.)
        MissIn nxt;
        bool have_next;
        {
            const WfDev& wfv = karg<WfDev, KA_WF>();  // (the kernel's arguments are re-read where they are used: karg)
            have_next = feed_next(feed, wfv, nxt.slot, nxt.valid);
            if (!have_next) nxt.slot = 0u, nxt.valid = false;
            load_miss_in<EAGER>(wfv, nxt);  // (unconditional: see the hit kernel)
        }
        RR_TICK(0)
        ItemRegs ir = cur.ir;
        if (cur.valid) {
            // throughput and light are loaded unconditionally and ignored while bounce == 1 (lib.rs:522-523)
            const bool first = (cur.bd & 0xffffu) <= 1u;
            const V3 thr = first ? mk(1.0, 1.0, 1.0) : cur.thr;
            V3 light = mk(0.0, 0.0, 0.0);
            if (!first && ir.has_light()) light = EAGER ? cur.light : load_light(karg<WfDev, KA_WF>(), cur.slot);
            const V3 result = v_add(light, v_mul(thr, background(karg<SceneDev, KA_SC>(), cur.d)));  // lib.rs:555
            ir.acc[0] += result.x;
            ir.acc[1] += result.y;
            ir.acc[2] += result.z;
        }
        RR_TICK(1)
        const NewRays nr = next_sample<COMPACT>(cur.valid, cur.slot, ir, true, range, sn);
        emit_rays<false>(cur.slot, nr, sn);
        RR_TICK(2)
        cur = nxt;
        have = have_next;
        RR_TICKS_BATCH();
    }
    RR_TICKS_END_MISS(rp)
    n_escaped = feed.total;
    store_item_range(wf, wave, range);
    if (lane == 0 && n_escaped) atomicAdd(&rp.counters->escaped_paths, n_escaped);
    store_sample_count(sc, rp, wf, sn);
}

// ----------------------------------------------------------- launch glue

static inline uint32_t trav_lds_bytes(bool compact, uint32_t stack_lds, uint32_t leafq, uint32_t hot_records) {
    return 4u * 64u * (stack_lds + 1u + leafq) * 4u + 4u * WINDOW * 2u + hot_records * (compact ? 144u : 272u);
}

uint32_t wf_window_slots() { return WINDOW; }

hipError_t wf_launch_init(const WfDev& wf, uint32_t live, hipStream_t stream) {
    hipLaunchKernelGGL(wf_init_kernel, dim3((wf.np + 255u) / 256u), dim3(256), 0, stream, wf, live);
    return hipGetLastError();
}

hipError_t wf_launch_gen(bool compact, const SceneDev& sc, const CameraDev& cam, const RenderDev& rp, const WfDev& wf,
                         uint32_t blocks, hipStream_t stream) {
    if (compact) hipLaunchKernelGGL(wf_gen_kernel<true>, dim3(blocks), dim3(256), 0, stream, sc, cam, rp, wf);
    else hipLaunchKernelGGL(wf_gen_kernel<false>, dim3(blocks), dim3(256), 0, stream, sc, cam, rp, wf);
    return hipGetLastError();
}

template <bool COMPACT, bool COUNT>
static hipError_t launch_trav_t(const SceneDev& sc, const RenderDev& rp, const WfDev& wf, uint32_t blocks,
                                hipStream_t stream) {
    const uint32_t lds = trav_lds_bytes(COMPACT, sc.stack_lds, sc.leafq, sc.hot_records);
    if (sc.exact && sc.hot) hipLaunchKernelGGL((wf_trav_kernel<COMPACT, COUNT, true, true>), dim3(blocks), dim3(256), lds, stream, sc, rp, wf);  // PRE
    else if (sc.exact) hipLaunchKernelGGL((wf_trav_kernel<COMPACT, COUNT, true, false>), dim3(blocks), dim3(256), lds, stream, sc, rp, wf);
    else hipLaunchKernelGGL((wf_trav_kernel<COMPACT, COUNT, false, false>), dim3(blocks), dim3(256), lds, stream, sc, rp, wf);
    return hipGetLastError();
}

hipError_t wf_launch_trav(bool compact, bool count, const SceneDev& sc, const RenderDev& rp, const WfDev& wf,
                          uint32_t blocks, hipStream_t stream) {
    if (compact)
        return count ? launch_trav_t<true, true>(sc, rp, wf, blocks, stream)
                     : launch_trav_t<true, false>(sc, rp, wf, blocks, stream);
    return count ? launch_trav_t<false, true>(sc, rp, wf, blocks, stream)
                 : launch_trav_t<false, false>(sc, rp, wf, blocks, stream);
}

template <bool COMPACT, bool COUNT>
static hipError_t trav_set_lds(uint32_t lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wf_trav_kernel<COMPACT, COUNT, false, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wf_trav_kernel<COMPACT, COUNT, true, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&wf_trav_kernel<COMPACT, COUNT, true, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

hipError_t wf_trav_occupancy(bool compact, uint32_t stack_lds, uint32_t leafq, uint32_t hot_records, int* blocks_per_cu) {
    hipError_t e = hipSuccess;
    const uint32_t lds = trav_lds_bytes(compact, stack_lds, leafq, hot_records);
    if (compact) {
        if ((e = trav_set_lds<true, false>(lds)) != hipSuccess) return e;
        if ((e = trav_set_lds<true, true>(lds)) != hipSuccess) return e;
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, wf_trav_kernel<true, false, false, false>, 256, lds);
    }
    if ((e = trav_set_lds<false, false>(lds)) != hipSuccess) return e;
    if ((e = trav_set_lds<false, true>(lds)) != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, wf_trav_kernel<false, false, false, false>, 256, lds);
}

hipError_t wf_launch_hit(bool compact, bool eager_light, const SceneDev& sc, const CameraDev& cam, const RenderDev& rp,
                         const WfDev& wf, uint32_t blocks, hipStream_t stream) {
    if (compact && eager_light)
        hipLaunchKernelGGL((wf_hit_kernel<true, true>), dim3(blocks), dim3(256), 0, stream, sc, cam, rp, wf);
    else if (compact)
        hipLaunchKernelGGL((wf_hit_kernel<true, false>), dim3(blocks), dim3(256), 0, stream, sc, cam, rp, wf);
    else if (eager_light)
        hipLaunchKernelGGL((wf_hit_kernel<false, true>), dim3(blocks), dim3(256), 0, stream, sc, cam, rp, wf);
    else
        hipLaunchKernelGGL((wf_hit_kernel<false, false>), dim3(blocks), dim3(256), 0, stream, sc, cam, rp, wf);
    return hipGetLastError();
}

hipError_t wf_launch_miss(bool compact, bool eager_light, const SceneDev& sc, const CameraDev& cam, const RenderDev& rp,
                          const WfDev& wf, uint32_t blocks, hipStream_t stream) {
    if (compact && eager_light)
        hipLaunchKernelGGL((wf_miss_kernel<true, true>), dim3(blocks), dim3(256), 0, stream, sc, cam, rp, wf);
    else if (compact)
        hipLaunchKernelGGL((wf_miss_kernel<true, false>), dim3(blocks), dim3(256), 0, stream, sc, cam, rp, wf);
    else if (eager_light)
        hipLaunchKernelGGL((wf_miss_kernel<false, true>), dim3(blocks), dim3(256), 0, stream, sc, cam, rp, wf);
    else
        hipLaunchKernelGGL((wf_miss_kernel<false, false>), dim3(blocks), dim3(256), 0, stream, sc, cam, rp, wf);
    return hipGetLastError();
}

}  // namespace rayrs
