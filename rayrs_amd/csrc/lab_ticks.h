// lab_ticks.h -- development instrumentation of the hit and miss kernels' loops (make -C rayrs_amd/csrc LAB=1 only):
// shader-clock shares of their stages, read by scripts/ubench/shade_ticks.py through rayrs_lab_ticks.  The timed build
// compiles every macro here to nothing; wavefront.hip holds only their one-word call sites.
#pragma once

#ifdef RAYRS_LAB_TICKS
// N accumulators, the batch count and the clock at the last mark
#define RR_TICKS_BEGIN(N) unsigned long long tk[N] = {}, tk_n = 0, tk_last = clock64()
// everything since the last mark is stage i's
#define RR_TICK(i)                                \
    {                                             \
        const unsigned long long now_ = clock64(); \
        tk[i] += now_ - tk_last, tk_last = now_;   \
    }
// ... after waiting for every load in flight (the wait for a batch's records, apart from the arithmetic behind it)
#define RR_TICK_LOADS_ARRIVED(i)                              \
    {                                                         \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      \
        RR_TICK(i)                                            \
    }
#define RR_TICKS_BATCH() tk_n++
// Counters::lab_ticks: hit kernel [0..4] stages 0..4, [5] batches, [6] stage 5, [7] stage 6; miss kernel [8..10] stages, [11] batches
#define RR_TICKS_END_HIT(rp)                                                                                         \
    if ((threadIdx.x & 63u) == 0) {                                                                                  \
        for (int i_ = 0; i_ < 5; i_++) atomicAdd(&(rp).counters->lab_ticks[i_], tk[i_]);                             \
        atomicAdd(&(rp).counters->lab_ticks[5], tk_n);                                                               \
        atomicAdd(&(rp).counters->lab_ticks[6], tk[5]), atomicAdd(&(rp).counters->lab_ticks[7], tk[6]);              \
    }
#define RR_TICKS_END_MISS(rp)                                                                   \
    if ((threadIdx.x & 63u) == 0) {                                                             \
        for (int i_ = 0; i_ < 3; i_++) atomicAdd(&(rp).counters->lab_ticks[8 + i_], tk[i_]);    \
        atomicAdd(&(rp).counters->lab_ticks[11], tk_n);                                         \
    }
#else
#define RR_TICKS_BEGIN(N)
#define RR_TICK(i)
#define RR_TICK_LOADS_ARRIVED(i)
#define RR_TICKS_BATCH()
#define RR_TICKS_END_HIT(rp)
#define RR_TICKS_END_MISS(rp)
#endif
