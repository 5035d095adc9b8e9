/*
 * atc_oracle.c — builds liboracle_atc.so: the CPU parity oracle (float64 + float32 instantiations).
 * TEST INFRASTRUCTURE ONLY — see atc_oracle_impl.h for scope, pinning and citations.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fno-fast-math)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../include/atc_step.h"

#define REAL double
#define SUFFIX _f64
#define ORC_FIXED_POS 0
#define R_FMOD fmod
#define R_FMA fma
#define R_SIN sin
#define R_COS cos
#define R_ACOS acos
#define R_SQRT sqrt
#define R_TANH tanh
#define R_POW pow
#define R_ABS fabs
#define R_HYPOT hypot
#define R_ATAN2 atan2
#define R_MIN(a, b) ((a) < (b) ? (a) : (b)) /* python min(a,b): b if b < a else a — identical for non-NaN */
#define R_MAX(a, b) ((a) > (b) ? (a) : (b))
#include "atc_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef ORC_FIXED_POS
#undef R_FMOD
#undef R_FMA
#undef R_SIN
#undef R_COS
#undef R_ACOS
#undef R_SQRT
#undef R_TANH
#undef R_POW
#undef R_ABS
#undef R_HYPOT
#undef R_ATAN2

#define REAL float
#define SUFFIX _f32
#define ORC_FIXED_POS 1
#define R_FMOD fmodf
#define R_FMA fmaf
#define R_SIN sinf
#define R_COS cosf
#define R_ACOS acosf
#define R_SQRT sqrtf
#define R_TANH tanhf
#define R_POW powf
#define R_ABS fabsf
#define R_HYPOT hypotf
#define R_ATAN2 atan2f
#include "atc_oracle_impl.h"

int atc_oracle_abi_version(void) { return ATC_ABI_VERSION; }

#ifdef _OPENMP
#include <omp.h>
void atc_oracle_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int atc_oracle_max_threads(void) { return omp_get_max_threads(); }
#else
void atc_oracle_set_threads(int n) { (void)n; }
int atc_oracle_max_threads(void) { return 1; }
#endif
