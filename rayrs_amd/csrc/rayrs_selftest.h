/*
 * rayrs_selftest.h -- PRIVATE device self-test hooks of librayrs_hip.so, for tests/ and scripts/ only: single
 * functions of the hot path run on the GPU so that they can be compared with the CPU checker one at a time.
 * Not part of the boundary a rayrs-lib maintainer binds (include/rayrs_hip.h): the reference has no counterpart,
 * and any of it may change.
 */
#ifndef RAYRS_SELFTEST_H
#define RAYRS_SELFTEST_H

#include <stdint.h>

#include "../../include/rayrs_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* fn: 0 sin 1 cos 2 tan 3 log 4 exp 5 acos 6 atan2(x[i], y[i]) 7 sqrt
 *     8 x/y 9 rng bits (x,y reinterpreted: unused) */
int rayrs_test_math(int device, int fn, const double* x, const double* y, uint64_t n, double* out);
int rayrs_test_rng(int device, uint64_t seed, const uint64_t* pixel, const uint64_t* sample, const uint32_t* draw,
                   uint64_t n, uint64_t* out_bits);
/* Bvh::intersect for n rays (o,d = n*3): t[i] and the object index (-1 miss).  exact: 1 = the default walk, 0 = the fast one */
int rayrs_test_intersect(rayrs_scene* scene, const double* o, const double* d, uint64_t n, int exact, double* t,
                         int64_t* object);
/* n samples, each exactly as a render starts and runs it -- the path key of (seed, pixel, sample), the primary ray, the
 * loop of lib.rs:521-560 -- in one lane each, with their traces: for sample i, n_queries[i] loop iterations, and for
 * the first min(n_queries[i], cap) of them object[i*cap + b] (-1: the query found nothing), t (0 then), throughput
 * (3 doubles) and the RNG draw index on leaving iteration b; rgb[i*3..]: radiance()'s value.  pixel[i] = row << 16 |
 * col in image coordinates.  exact: as rayrs_test_intersect. */
int rayrs_test_path_trace(rayrs_scene* scene, const rayrs_camera* camera, uint64_t seed, uint32_t max_bounces,
                          const uint32_t* pixel, const uint32_t* sample, uint64_t n, int exact, uint32_t cap,
                          uint32_t* n_queries, int64_t* object, double* t, double* throughput, uint32_t* draw, double* rgb);
/* Material::evaluate for n (normal, view, key) tuples with one material:
 * scattered[i] 0/1, color/dir = n*3, draws[i] = number of draws consumed. */
int rayrs_test_material(int device, const rayrs_material* mat, const double* normal, const double* view,
                        const uint64_t* key, uint64_t n, int32_t* scattered, double* color, double* dir,
                        uint32_t* draws);
/* Scene::background for n directions. */
int rayrs_test_background(rayrs_scene* scene, const double* dir, uint64_t n, double* rgb);

#ifdef __cplusplus
}
#endif
#endif
