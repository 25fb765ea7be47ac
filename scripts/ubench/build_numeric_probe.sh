#!/bin/bash
# An UPPER BOUND for what any rewrite of include/rayrs_numeric.h could buy the shading kernels: a build of
# the library whose elementary functions (device side only) are single f32 hardware instructions -- v_sin_f32, v_log_f32,
# v_exp_f32 ... through the compiler's builtins -- instead of the header's f64 polynomials.  Its frames are NOT the product's
# (6 decimal digits instead of 16), only its kernel times mean something: no polynomial arrangement (Estrin, paired
# evaluation, specialised entry points) can be cheaper than one instruction.  Built outside the tree into
# scripts/ubench/alt/numeric_probe.so; compare on one box with scripts/ubench/abc_libs.sh (RAYRS_HIP_LIB).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=/tmp/rayrs_numeric_probe
rm -rf $B && mkdir -p $B/include $B/rayrs_amd
cp $ROOT/include/*.h $B/include/
cp -r $ROOT/rayrs_amd/csrc $B/rayrs_amd/csrc
rm -f $B/rayrs_amd/csrc/*.o
python3 - "$B/include/rayrs_numeric.h" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
probe = r'''
/* ---- PROBE BUILD ONLY (scripts/ubench/build_numeric_probe.sh): f32 hardware stand-ins on the device ---- */
#if defined(__HIP_DEVICE_COMPILE__)
#define rr_sin rr_sin_f64
#define rr_cos rr_cos_f64
#define rr_sincos rr_sincos_f64
#define rr_tan rr_tan_f64
#define rr_log rr_log_f64
#define rr_exp rr_exp_f64
#define rr_acos rr_acos_f64
#define rr_atan rr_atan_f64
#define rr_atan2 rr_atan2_f64
#endif
'''
tail = r'''
#if defined(__HIP_DEVICE_COMPILE__)
#undef rr_sin
#undef rr_cos
#undef rr_sincos
#undef rr_tan
#undef rr_log
#undef rr_exp
#undef rr_acos
#undef rr_atan
#undef rr_atan2
RR_FN double rr_sin(double x) { return (double)__sinf((float)x); }
RR_FN double rr_cos(double x) { return (double)__cosf((float)x); }
RR_FN rr_sincos_t rr_sincos(double x) { rr_sincos_t o; o.s = (double)__sinf((float)x); o.c = (double)__cosf((float)x); return o; }
RR_FN double rr_tan(double x) { return (double)(__sinf((float)x) / __cosf((float)x)); }
RR_FN double rr_log(double x) { return (double)__logf((float)x); }
RR_FN double rr_exp(double x) { return (double)__expf((float)x); }
RR_FN double rr_acos(double x) { return (double)acosf((float)x); }
RR_FN double rr_atan(double x) { return (double)atanf((float)x); }
RR_FN double rr_atan2(double y, double x) { return (double)atan2f((float)y, (float)x); }
#endif
'''
mark = "/* -------------------------------------------------------- sin / cos / tan */"
assert mark in s
s = s.replace(mark, probe + mark, 1)
s = s.replace("#endif /* RAYRS_NUMERIC_H */", tail + "#endif /* RAYRS_NUMERIC_H */")
open(p, "w").write(s)
PY
make -C $B/rayrs_amd/csrc ../librayrs_hip.so > $B/build.log 2>&1 || { tail -20 $B/build.log; exit 1; }
mkdir -p $ROOT/scripts/ubench/alt
cp $B/rayrs_amd/librayrs_hip.so $ROOT/scripts/ubench/alt/numeric_probe.so
ls -la $ROOT/scripts/ubench/alt/numeric_probe.so
