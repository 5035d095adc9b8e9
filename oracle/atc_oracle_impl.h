/*
 * atc_oracle_impl.h — CPU restatement of the reference's AtcGym.step() hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle (checker) for the HIP product path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  It is never imported by the product package.
 *
 * Included twice by atc_oracle.c:
 *   REAL = double (suffix _f64): the reference as it is — Python floats, libm; pinned against the golden vectors captured
 *     from the imported reference (integer outputs exact, state within 1e-11);
 *   REAL = float (suffix _f32): the fp32 restatement — the same operation order in float with libm's float functions,
 *     EXCEPT the choices include/atc_step.h fixes for every fp32 implementation ("Aircraft positions", "Speed and heading":
 *     the fixed-point position grid; since ABI 18 the fixed-point speed / heading state with its integer rate limits, the
 *     truncating target conversion, the float64 heading kinematics and the dithered rounding; since ABI 20 the ALTITUDE as the
 *     reference's float64, operation for operation, and the timestep as a float64).  With those shared, the HIP kernels' aircraft state and every
 *     integer output match this instantiation bit for bit; it is itself pinned against the same golden vectors (integer
 *     outputs exact, fp32 values within 1e-5 — everywhere: the near-FAF exception of rounds 1-3 is retired).
 * Scalar, compiled with -ffp-contract=off.
 *
 * Pinned by: tests/golden/{g1..g12,model_test_known_answers} (see tests/test_oracle_golden.py): 48 080 + 650 963 + 93 879
 * reference steps at timesteps 1 / 2 / 5 s, the g12 episodes at 0.05 .. 3.7 s, lattices, tie-break points, shaping grids, the
 * reference's own 8 unit-test answers.
 * The multi-aircraft separation scan and noise-abatement areas have NO reference implementation
 * (README.md:51,60,62 are prose only): for those paths parity is UNPINNED and this file is the definition.
 *
 * Citations are path:line in the reference tree (fvalka/atc-reinforcement-learning).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* ---- elementary helpers --------------------------------------------------------------------------------------- */

/* Aircraft positions (model.py:33-34).
 *   float64 instantiation: Python floats, x += d  — the reference as it is.
 *   float32 instantiation: the fp32 spec of include/atc_step.h ("Aircraft positions"): 32-bit fixed point on the sector's
 *   position grid, saturating; the fp32 value the reference's formulas see is (float)(origin + fix * 2^-k); the vector
 *   to the FAF is an exact integer difference.  The HIP kernels implement the same spec, so their positions are
 *   bit-identical to this instantiation's. */
#if ORC_FIXED_POS
typedef int32_t FN(pos_t);
static int32_t FN(sat32)(int64_t v) { return v > INT32_MAX ? INT32_MAX : (v < INT32_MIN ? INT32_MIN : (int32_t)v); }
static int32_t FN(pos_spawn)(const REAL* S, int axis, REAL v) { /* entry point -> grid, in fp32 like the device */
    REAL c = (v - S[ATC_C_POS_X0 + axis]) * S[ATC_C_POS_SCALE];
    c = c > (REAL)2147483520.0 ? (REAL)2147483520.0 : (c < (REAL)-2147483648.0 ? (REAL)-2147483648.0 : c); /* int32 range */
    return (int32_t)rintf(c);
}
static REAL FN(pos_to_real)(const REAL* S, int axis, int32_t p) {
    return (REAL)((double)p * (double)S[ATC_C_POS_INV] + (double)S[ATC_C_POS_X0 + axis]);
}
static int32_t FN(pos_advance)(const REAL* S, int32_t p, REAL d) {
    REAL c = d * S[ATC_C_POS_SCALE];
    c = c > (REAL)1073741824.0 ? (REAL)1073741824.0 : (c < (REAL)-1073741824.0 ? (REAL)-1073741824.0 : c); /* +-2^30 */
    return FN(sat32)((int64_t)p + (int64_t)rintf(c));
}
static REAL FN(pos_to_faf)(const REAL* S, int axis, int32_t p) {
    int64_t faf = (int64_t)S[ATC_C_FAF_FIX + 2 * axis] * 65536 + (int64_t)S[ATC_C_FAF_FIX + 2 * axis + 1];
    return (REAL)FN(sat32)(faf - (int64_t)p) * S[ATC_C_POS_INV];
}
/* Speed and heading state of the fp32 spec (include/atc_step.h, ABI 18): 32-bit fixed point,
 *   kt = v_fix 2^-23 (unsigned),  deg = 180 + phi_fix 2^-23 (signed);  targets, rate limits and the action discriminator are
 *   integer arithmetic, the displacement is float64 from the fixed-point state. */
typedef int32_t FN(fix_t); /* (the speed's counts are unsigned: stored in the same 32 bits, used through uint32_t) */
#define ORC_QV 8388608.0 /* 2^ATC_V_FIX_SHIFT */
#define ORC_QP 8388608.0 /* 2^ATC_PHI_FIX_SHIFT */
static int32_t FN(clampi)(int32_t d, int32_t lo, int32_t hi) { return d < lo ? lo : (d > hi ? hi : d); }
/* float64 -> integer as the spec defines it: truncation toward zero, saturation at both ends, NaN -> 0 */
static int32_t FN(trunc_i32)(double x) {
    if (!(x == x)) return 0;
    if (x >= 2147483647.0) return INT32_MAX;
    if (x <= -2147483648.0) return INT32_MIN;
    return (int32_t)x;
}
static uint32_t FN(trunc_u32)(double x) {
    if (!(x == x) || x <= 0.0) return 0u;
    if (x >= 4294967295.0) return UINT32_MAX;
    return (uint32_t)x;
}
/* Heading counts are 64-bit (include/atc_step.h, ABI 19): target = trunc(a m + c) clamped to +-2^52, NaN -> 0 */
#define ORC_PHI_LIMIT 4503599627370496.0 /* 2^52 */
static int64_t FN(trunc_i64_phi)(double x, int* clamped) {
    *clamped = 0;
    if (!(x == x)) return 0;
    x = trunc(x);
    if (x > ORC_PHI_LIMIT) { *clamped = 1; x = ORC_PHI_LIMIT; }
    if (x < -ORC_PHI_LIMIT) { *clamped = 1; x = -ORC_PHI_LIMIT; }
    return (int64_t)x;
}
static int FN(is_wide)(int32_t f) { return f == INT32_MAX || f == INT32_MIN; }
/* the exact counts of a stored (32-bit field, 64-bit side word) pair, and back */
static int64_t FN(phi_load)(int32_t f, int64_t wide) { return FN(is_wide)(f) ? wide : (int64_t)f; }
static void FN(phi_put)(int64_t P, int32_t* f, int64_t* wide) {
    *f = FN(sat32)(P);
    if (FN(is_wide)(*f)) *wide = P;
}
/* a heading wrapped to within half a turn of the format's origin (include/atc_step.h): Pw = fma(rint(P RN(1/W)), -W, P) in float64 —
 * exact, and the very operations of the device */
static int32_t FN(phi_wrap)(int64_t P) {
    const double pd = (double)P;
    return (int32_t)fma(rint(pd * ATC_PHI_INV_TURN), -ATC_PHI_TURN, pd);
}
/* the heading as kinematics / angles see it, and observation word 3 (atc_gym.py:269) */
static int32_t FN(phi_eff)(int32_t f, int64_t wide) { return FN(is_wide)(f) ? FN(phi_wrap)(wide) : f; }
typedef struct FN(decode) { double m, c; } FN(decode_t);
/* target in counts = a * m + c (one fma in float64, m and c integers) */
static FN(decode_t) FN(decode_consts)(double fac, double add, double offset, double q) {
    FN(decode_t) d;
    d.m = fac * q;
    d.c = (add - offset) * q;
    return d;
}
static int32_t FN(rate_fix)(double rate, double dt, double q) { /* rint(|rate| dt 2^s), saturating */
    double r = rint(fabs(rate) * dt * q);
    return r >= 2147483647.0 ? INT32_MAX : (int32_t)r;
}
static REAL FN(v_real)(int32_t f) { return (float)(uint32_t)f * (float)(1.0 / ORC_QV); }
static REAL FN(phi_real)(int32_t f) { return fmaf((float)f, (float)(1.0 / ORC_QP), ATC_PHI_FIX_OFFSET); }
static REAL FN(phi_obs)(int32_t f, int64_t wide) {
    return FN(is_wide)(f) ? (float)((double)ATC_PHI_FIX_OFFSET + (double)wide * (1.0 / ORC_QP)) : FN(phi_real)(f);
}
/* state placed from outside (spawn, fixtures): nearest count */
static int32_t FN(v_store)(REAL v) { return (int32_t)FN(trunc_u32)(rint((double)v * ORC_QV)); }
static int32_t FN(phi_store)(REAL p) { return FN(trunc_i32)(rint(((double)p - (double)ATC_PHI_FIX_OFFSET) * ORC_QP)); }
/* float64 heading kinematics of include/atc_step.h: advances (x, y) by one step's displacement */
static void FN(advance)(const REAL* S, double dist_a, int32_t phi_fix, int32_t v_fix, int32_t t_step, int32_t* x, int32_t* y) {
    (void)S;
    const double pd = (double)phi_fix;
    const double k = rint(pd * ATC_KIN_INV180);
    const double t = fma(k, -ATC_KIN_HALF_TURN, pd);
    const double u = t * t;
    const double sp = fma(fma(fma(fma(fma(ATC_KIN_S5, u, ATC_KIN_S4), u, ATC_KIN_S3), u, ATC_KIN_S2), u, ATC_KIN_S1), u, ATC_KIN_S0);
    const double sn = sp * t;
    const double cs = fma(fma(fma(fma(fma(ATC_KIN_C5, u, ATC_KIN_C4), u, ATC_KIN_C3), u, ATC_KIN_C2), u, ATC_KIN_C1), u, 1.0);
    double dist = (double)(uint32_t)v_fix * dist_a; /* the step's distance in position-grid counts */
    if (!(((int32_t)k) & 1)) dist = -dist;         /* phi = 180 (1 + k) + t: an even k is an odd number of half turns */
    /* Dithered rounding (include/atc_step.h): counts += floor(displacement + u), u = the low 11 bits of the env's time step
     * bit-reversed, as a fraction (van der Corput sequence).  The rounding errors of a constant displacement — a straight leg
     * at constant speed — then cancel to O(log n) counts over n steps instead of adding up to n / 4.  One fma against
     * 1.5 2^41 + u (its bit pattern is assembled from integers), the integer part read from bits 11..42. */
    uint32_t rb = 0;
    for (int b = 0; b < 11; ++b) rb |= (((uint32_t)t_step >> b) & 1u) << (10 - b);
    const uint64_t mb = ((uint64_t)ATC_DITHER_MAGIC_HI << 32) | (uint64_t)rb;
    double magic_u;
    memcpy(&magic_u, &mb, sizeof magic_u);
    double rx = fma(sn, dist, magic_u), ry = fma(cs, dist, magic_u);
    uint64_t bx, by;
    memcpy(&bx, &rx, sizeof bx);
    memcpy(&by, &ry, sizeof by);
    *x = FN(sat32)((int64_t)*x + (int64_t)(int32_t)(uint32_t)(bx >> 11));
    *y = FN(sat32)((int64_t)*y + (int64_t)(int32_t)(uint32_t)(by >> 11));
}
#else
typedef double FN(pos_t);
static double FN(pos_spawn)(const REAL* S, int axis, REAL v) { (void)S; (void)axis; return v; }
static REAL FN(pos_to_real)(const REAL* S, int axis, double p) { (void)S; (void)axis; return (REAL)p; }
static double FN(pos_advance)(const REAL* S, double p, REAL d) { (void)S; return p + (double)d; }
static REAL FN(pos_to_faf)(const REAL* S, int axis, double p) { return S[ATC_C_FAF_X + axis] - (REAL)p; }
typedef REAL FN(fix_t);
static REAL FN(v_real)(REAL f) { return f; }
static REAL FN(phi_real)(REAL f) { return f; }
static REAL FN(v_store)(REAL v) { return v; }
static REAL FN(phi_store)(REAL p) { return p; }
static REAL FN(phi_eff)(REAL f, int64_t wide) { (void)wide; return f; }
static REAL FN(phi_obs)(REAL f, int64_t wide) { (void)wide; return f; }
/* model.py:345-348 rot_matrix: sin / cos of math.radians(phi) */
static void FN(sincos_heading)(REAL phi, REAL* sn, REAL* cs) {
    REAL pr = phi * (REAL)(3.14159265358979323846 / 180.0);
    *sn = R_SIN(pr);
    *cs = R_COS(pr);
}
#endif

/* Python float modulo: result takes the sign of the divisor (used by relative_angle, model.py:340-342). */
static REAL FN(py_mod)(REAL a, REAL b) {
    REAL r = R_FMOD(a, b);
    if (r != (REAL)0 && ((r < (REAL)0) != (b < (REAL)0))) r += b;
    return r;
}

/* model.py:340-342  relative_angle(angle1, angle2) = (angle2 - angle1 + 180) % 360 - 180 */
static REAL FN(relative_angle)(REAL a1, REAL a2) {
    return FN(py_mod)(a2 - a1 + (REAL)180, (REAL)360) - (REAL)180;
}

/* np.degrees: x * (180 / pi) */
static REAL FN(degrees)(REAL x) { return x * (REAL)(180.0 / 3.14159265358979323846); }

/* model.py:318-337  ray_tracing(x, y, poly): crossing number over n+1 edges with the reference's inequality set.
 * ring = x,y interleaved, n vertices (closed ring: first == last). */
static int FN(ray_tracing)(REAL x, REAL y, const REAL* ring, int n) {
    int inside = 0;
    REAL xints = (REAL)0;
    REAL p1x = ring[0], p1y = ring[1];
    for (int i = 0; i < n + 1; ++i) {
        int k = i % n;
        REAL p2x = ring[2 * k], p2y = ring[2 * k + 1];
        if (y > R_MIN(p1y, p2y)) {
            if (y <= R_MAX(p1y, p2y)) {
                if (x <= R_MAX(p1x, p2x)) {
                    if (p1y != p2y) xints = (y - p1y) * (p2x - p1x) / (p2y - p1y) + p1x;
                    if (p1x == p2x || x <= xints) inside = !inside;
                }
            }
        }
        p1x = p2x;
        p1y = p2y;
    }
    return inside;
}

/* model.py:282-292  Airspace.find_mva / get_mva_height: first polygon in list order whose (inclusive) bounds contain
 * the point and whose ray_tracing is true.  Returns the polygon index or -1 ("Outside of airspace"). */
static int FN(find_mva)(const REAL* S, REAL x, REAL y) {
    int n_mva = (int)S[ATC_H_N_MVA];
    int off = (int)S[ATC_H_OFF_POLY];
    for (int p = 0; p < n_mva; ++p) {
        const REAL* rec = S + off + p * ATC_P_WORDS;
        if (rec[ATC_P_MINX] <= x && x <= rec[ATC_P_MAXX] && rec[ATC_P_MINY] <= y && y <= rec[ATC_P_MAXY]) {
            if (FN(ray_tracing)(x, y, S + (int)rec[ATC_P_VOFF], (int)rec[ATC_P_NVERT])) return p;
        }
    }
    return -1;
}

/* model.py:212-231  Corridor._inside_corridor_angle.  NOTE: min_angle is in RADIANS (arccos) while relative_angle is in
 * DEGREES — reproduced as is (survey quirk Q5).
 * The arccos argument is evaluated in float64 in BOTH instantiations, as fma(c1, c2, s1 * s2): that is what the
 * reference's np.dot (BLAS, FMA) computes, and it decides the knife-edge "heading exactly equals the runway heading"
 * (dot == 1.0 -> arccos == 0 -> 0 <= 0 passes; measured: plain mul/add gives 0.9999999999999999 for 340 vs 700 deg where
 * np.dot gives 1.0).  Keeping this one expression in float64 makes the fp32 instantiation agree with the reference for
 * exactly aligned (e.g. integer, discrete-action) headings. */
#if ORC_FIXED_POS
/* fp32 spec (include/atc_step.h, round 5): the window's relative angle is evaluated EXACTLY from the heading's counts — its sign
 * decides a flag, and an fp32 heading (ulp 3e-5 deg at 340) cannot tell 340 -/+ 1e-5 deg from 340; the window itself in its
 * exact-arithmetic form 0 <= rel <= 45 (rel == 0: the reference's own rounding luck, ATC_C_ALIGNED_OK), like the kernels.
 * pc = heading counts (any value congruent to the heading modulo a turn). */
static double FN(rel_counts)(double d) { return fma(rint(d * ATC_PHI_INV_TURN), -ATC_PHI_TURN, d); }
static int FN(angle_window)(const REAL* S, double rel) {
    return (rel > 0.0 && rel <= (double)S[ATC_C_FAF_ANGLE] * ORC_QP) || (rel == 0.0 && S[ATC_C_ALIGNED_OK] != (REAL)0);
}
static int FN(inside_corridor_angle)(const REAL* S, REAL x, REAL y, double pc) {
    const double q = rint(((double)S[ATC_C_PHI_TO_RWY] - (double)ATC_PHI_FIX_OFFSET) * ORC_QP);
    if (FN(ray_tracing)(x, y, S + ATC_C_TRI_1, 4) && FN(angle_window)(S, FN(rel_counts)(pc - q))) return 1;
    if (FN(ray_tracing)(x, y, S + ATC_C_TRI_2, 4) && FN(angle_window)(S, FN(rel_counts)(q - pc))) return 1;
    return 0;
}
static double FN(heading_counts_of)(REAL phi) { return rint(((double)phi - (double)ATC_PHI_FIX_OFFSET) * ORC_QP); }
#else
static double FN(heading_counts_of)(REAL phi) { return (double)phi; } /* the reference takes degrees */
static int FN(inside_corridor_angle)(const REAL* S, REAL x, REAL y, double phi_in) {
    const REAL phi = (REAL)phi_in;
    REAL to_runway = S[ATC_C_PHI_TO_RWY];
    REAL faf_angle = S[ATC_C_FAF_ANGLE];
    /* rot_matrix(a) . [0,1] = (sin(rad a), cos(rad a)) */
    double tr = (double)to_runway * (3.14159265358979323846 / 180.0);
    double pr = (double)phi * (3.14159265358979323846 / 180.0);
    double dot = fma(cos(tr), cos(pr), sin(tr) * sin(pr));
    REAL beta = faf_angle - (REAL)acos(dot);
    REAL min_angle = faf_angle - beta;
    /* model.py:224-229: `if tri1 and window: True / elif tri2 and window: True / False` */
    if (FN(ray_tracing)(x, y, S + ATC_C_TRI_1, 4)) {
        REAL ra = FN(relative_angle)(to_runway, phi);
        if (min_angle <= ra && ra <= faf_angle) return 1;
    }
    if (FN(ray_tracing)(x, y, S + ATC_C_TRI_2, 4)) {
        REAL ra = FN(relative_angle)(phi, to_runway);
        if (min_angle <= ra && ra <= faf_angle) return 1;
    }
    return 0;
}
#endif

/* model.py:188-210  Corridor.inside_corridor */
static int FN(inside_corridor)(const REAL* S, REAL x, REAL y, REAL h, double phi) { /* phi: degrees (f64) | heading counts (fp32 spec) */
    if (!FN(ray_tracing)(x, y, S + ATC_C_TRI_H, 4)) return 0;
    REAL fx = S[ATC_C_FAF_X], fy = S[ATC_C_FAF_Y], nx = S[ATC_C_NRM_X], ny = S[ATC_C_NRM_Y];
    REAL t = (x - fx) * nx + (y - fy) * ny;            /* np.dot(p - faf^T, normal) */
    REAL px = fx + t * nx, py = fy + t * ny;           /* faf + t * normal */
    REAL dx = px - S[ATC_C_RWY_X], dy = py - S[ATC_C_RWY_Y];
    REAL nrm = R_SQRT(dx * dx + dy * dy);              /* np.linalg.norm */
    REAL h_max = nrm * S[ATC_C_GS_TAN] * S[ATC_C_NM_TO_FT] + S[ATC_C_RWY_H];
    if (!(h <= h_max)) return 0;
    return FN(inside_corridor_angle)(S, x, y, phi);
}

/* atc_gym.py:17-19 */
static REAL FN(sigmoid_distance)(REAL d, REAL d_max) {
    return ((REAL)1.0 - R_TANH((REAL)4.0 * (d / d_max) - (REAL)2.0)) / (REAL)2.0;
}

/* atc_gym.py:199-222 */
static REAL FN(reward_approach_position)(REAL d_faf, REAL phi_to_runway, REAL phi_rel_to_faf, REAL world_max_dist) {
    REAL reward_faf = FN(sigmoid_distance)(d_faf, world_max_dist);
    REAL reward_app_angle = R_POW(R_ABS(FN(relative_angle)(phi_to_runway, phi_rel_to_faf)) / (REAL)180.0, (REAL)1.5);
    return reward_faf * reward_app_angle * (REAL)0.8;
}

/* atc_gym.py:224-237 */
static REAL FN(reward_glideslope)(REAL h, REAL on_gp_altitude, REAL position_factor) {
    REAL f = FN(sigmoid_distance)(R_ABS(h - on_gp_altitude), (REAL)36000);
    return f * position_factor * (REAL)0.8;
}

/* atc_gym.py:239-260 */
static REAL FN(reward_approach_angle)(REAL phi_to_runway, REAL phi_rel_to_faf, REAL phi_plane, REAL position_factor) {
    REAL plane_to_runway = FN(relative_angle)(phi_to_runway, phi_plane);
    REAL s = FN(relative_angle)(phi_to_runway, phi_rel_to_faf);
    REAL side = (s > (REAL)0) ? (REAL)1 : ((s < (REAL)0) ? (REAL)-1 : (REAL)0); /* np.sign */
    REAL angle = side * plane_to_runway;
    REAL q = (angle - (REAL)22.5) / (REAL)202.0;
    REAL model = R_POW(-R_POW(q, (REAL)2.0) + (REAL)1.0, (REAL)32.0);
    return model * position_factor * (REAL)1.2;
}

/* ---- counter-based RNG for entry-point draws (build-defined; integer-only so every implementation agrees) -------- */
static uint64_t FN(mix64)(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t FN(draw)(uint64_t seed, uint32_t env, uint32_t episode, uint32_t slot) {
    uint64_t z = FN(mix64)(seed ^ ((uint64_t)env << 32 | (uint64_t)episode));
    return FN(mix64)(z ^ (uint64_t)slot);
}

/* ---- structures (host pointers; same field meaning as include/atc_step.h) ----------------------------------------- */
typedef struct FN(orc_state) {
    FN(pos_t) *x, *y;          /* [B*N] positions: float64 (reference) | 32-bit fixed point (fp32 spec), see pos_t above */
    double* h;                 /* [B*N] altitude: float64 in BOTH instantiations (include/atc_step.h, ABI 20) */
    FN(fix_t) *phi, *v;        /* [B*N] heading, speed: REAL (reference) | 32-bit fixed point (fp32 spec), see fix_t above */
    FN(fix_t)* last_act;       /* last accepted v / h / phi targets.  Reference instantiation: [3][B*N] float64 (v, h, phi).
                                  fp32 spec: [B*N][4] 32-bit words in the DEVICE's record layout (atc_state_t.last_act):
                                  v_fix, phi_fix, the altitude target's float64 in words 2..3 */
    int32_t* timesteps;
    int32_t* actions_taken;
    REAL* total_reward;
    uint64_t* active_mask;
    uint32_t* win_bits;
    int32_t* episodes;
    REAL* ep_return;
    int32_t* ep_length;
    int32_t* ep_actions;       /* actions_taken of the last finished episode */
    int64_t* phi_wide;         /* [B*N][2] fp32 spec only: exact heading counts / last heading target of WIDE aircraft
                                  (include/atc_step.h, atc_state_t.phi_wide) */
} FN(orc_state_t);

typedef struct FN(orc_out) {
    float* obs;     /* [B*N*10] float32 like the reference (atc_gym.py:276) */
    float* raw_obs; /* nullable */
    REAL* reward;   /* [B] */
    REAL* ac_reward;/* nullable [B*N] */
    uint8_t* done;  /* [B] */
    uint16_t* flags;/* [B*N] */
    REAL* min_sep;  /* nullable [B] */
    float* term_obs;/* nullable */
    int32_t* mva;   /* nullable [B*N] MVA height used for the observation (-1 outside) — oracle-only diagnostic */
} FN(orc_out_t);

/* atc_gym.py:262-277 _get_state -> float32[10]; also returns the full-precision d_faf / phi_rel_faf / on_gp used by the
 * shaping rewards (atc_gym.py:179-185 read self._d_faf etc., which are not rounded to float32). */
static void FN(get_state)(const REAL* S, FN(pos_t) px, FN(pos_t) py, double h, REAL phi, REAL phi_obs, REAL v, REAL mva, float* obs10,
                          REAL* d_faf, REAL* phi_rel_faf, REAL* on_gp) {
    /* h: the float64 altitude (both instantiations, include/atc_step.h ABI 20): words 2 and 5 are np.float32 of the reference's
     * float64 h and h - mva (atc_gym.py:266-267,276) */
    /* phi: the heading the relative angle sees; phi_obs: observation word 3 — the same number in the reference, the wrapped /
     * unwrapped pair of a WIDE heading in the fp32 spec (include/atc_step.h) */
    REAL x = FN(pos_to_real)(S, 0, px), y = FN(pos_to_real)(S, 1, py);
    REAL to_faf_x = FN(pos_to_faf)(S, 0, px);  /* faf - position, atc_gym.py:289-297 */
    REAL to_faf_y = FN(pos_to_faf)(S, 1, py);
    REAL phi_rel_runway = FN(relative_angle)(S[ATC_C_PHI_TO_RWY], phi); /* atc_gym.py:284-287 */
    *d_faf = R_HYPOT(to_faf_x, to_faf_y);                               /* atc_gym.py:294-297 */
    *phi_rel_faf = FN(degrees)(R_ATAN2(to_faf_y, to_faf_x));            /* atc_gym.py:289-292 */
    *on_gp = (REAL)318.4 * (*d_faf) + S[ATC_C_FAF_MVA] - (REAL)200;     /* atc_gym.py:279-282 */
    obs10[0] = (float)x;
    obs10[1] = (float)y;
    obs10[2] = (float)h;
    obs10[3] = (float)phi_obs;
    obs10[4] = (float)v;
    obs10[5] = (float)(h - (double)mva);
    obs10[6] = (float)*on_gp;
    obs10[7] = (float)*d_faf;
    obs10[8] = (float)*phi_rel_faf;
    obs10[9] = (float)phi_rel_runway;
}

/* atc_gym.py:187-189: float32 arithmetic on float32 vectors: (state - min - 0.5*max) / (0.5*max) */
static void FN(normalize)(const REAL* S, const float* raw, float* out) {
    for (int i = 0; i < 10; ++i) {
        float mn = (float)S[ATC_C_NORM_MIN + i];
        float half = 0.5f * (float)S[ATC_C_NORM_MAX + i];
        float t = raw[i] - mn;
        t = t - half;
        out[i] = t / half;
    }
}

/* atc_gym.py:346-348 + model.py:13-52: place aircraft `k` of env `e` at an entry point.
 * Lattice mode: slot k -> entry k mod E, level (k div E) mod n_levels (precomputed per slot in the blob).
 * Random mode: a 64-bit draw keyed by (seed, env, episode, slot); entry = (lo32 * E) >> 32, level = (hi32 * n_levels) >> 32
 * (multiply-shift range reduction: integer-only, so every implementation agrees). */
static void FN(spawn)(const REAL* S, const atc_params_t* p, int e, int k, int episode, REAL* x, REAL* y, double* h,
                      REAL* phi, REAL* v) {
    *v = S[ATC_C_V_INIT];
    if (!(p->mode & ATC_M_RANDOM_ENTRY)) {
        const REAL* rec = S + (int)S[ATC_H_OFF_SLOT] + 4 * k;
        *x = rec[0];
        *y = rec[1];
        *phi = rec[2];
        *h = rec[3];
        return;
    }
    const uint32_t n_entry = (uint32_t)(int)S[ATC_H_N_ENTRY];
    const uint64_t u = FN(draw)(p->seed, (uint32_t)e, (uint32_t)episode, (uint32_t)k);
    const int ei = (int)(((u & 0xffffffffull) * n_entry) >> 32);
    const REAL* rec = S + (int)S[ATC_H_OFF_ENTRY] + ei * ATC_E_WORDS;
    const int li = (int)(((u >> 32) * (uint32_t)(int)rec[ATC_E_NLEV]) >> 32);
    *x = rec[ATC_E_X];
    *y = rec[ATC_E_Y];
    *phi = rec[ATC_E_PHI];
    *h = (double)(rec[ATC_E_LEV0 + li] * (REAL)100);
}

/* AtcGym.reset (atc_gym.py:337-365) for env e; writes RAW obs computed with mva = 0 (atc_gym.py:351,365).
 * last_action is NOT touched here: the reference sets it once in __init__ (atc_gym.py:86, quirk Q7). */
static void FN(reset_env)(const REAL* S, int N, const FN(orc_state_t) * st, const atc_params_t* p, int e, float* obs,
                          int first) {
    if (first) {
        st->win_bits[e] = 0;
        st->episodes[e] = 0;
        st->ep_return[e] = 0;
        st->ep_length[e] = 0;
        st->ep_actions[e] = 0;
    }
    int episode = st->episodes[e];
    for (int k = 0; k < N; ++k) {
        int i = e * N + k;
        REAL sx, sy, sphi, sv;
        FN(spawn)(S, p, e, k, episode, &sx, &sy, &st->h[i], &sphi, &sv);
        st->x[i] = FN(pos_spawn)(S, 0, sx);
        st->y[i] = FN(pos_spawn)(S, 1, sy);
        st->phi[i] = FN(phi_store)(sphi);
        st->v[i] = FN(v_store)(sv);
        if (obs) {
            REAL d, pr, gp;
            FN(get_state)(S, st->x[i], st->y[i], st->h[i], FN(phi_real)(st->phi[i]), FN(phi_real)(st->phi[i]), FN(v_real)(st->v[i]),
                          (REAL)0, obs + (size_t)i * 10, &d, &pr, &gp);
        }
    }
    st->total_reward[e] = 0;
    st->actions_taken[e] = 0;
    st->timesteps[e] = 0;
    st->episodes[e] = episode + 1;
    st->active_mask[e] = (N >= 64) ? ~0ull : ((1ull << N) - 1ull);
}

int FN(atc_oracle_reset)(const REAL* S, int B, int N, const FN(orc_state_t) * st, const uint8_t* mask, float* obs,
                         const atc_params_t* p, int first) {
    if (!S || !st || !p || B < 0 || N < 1 || N > ATC_MAX_AIRCRAFT) return -1;
    size_t BN = (size_t)B * N;
    for (int e = 0; e < B; ++e) {
        if (mask && !mask[e]) continue;
        if (first) /* atc_gym.py:86: last_action = [0, 0, 0] — in the state's formats (fp32 spec: the counts of 0 kt / 0 deg, 0.0 ft) */
            for (int k = 0; k < N; ++k) {
#if ORC_FIXED_POS
                int32_t* la = &st->last_act[4 * ((size_t)e * N + k)];
                la[0] = FN(v_store)((REAL)0);
                la[1] = FN(phi_store)((REAL)0);
                la[2] = la[3] = 0; /* the float64 0.0 */
                (void)BN;
#else
                st->last_act[(size_t)0 * BN + (size_t)e * N + k] = 0;
                st->last_act[(size_t)1 * BN + (size_t)e * N + k] = 0;
                st->last_act[(size_t)2 * BN + (size_t)e * N + k] = 0;
#endif
            }
        FN(reset_env)(S, N, st, p, e, obs, first);
    }
    return 0;
}

/* AtcGym.step (atc_gym.py:128-192) for B envs x N aircraft. */
int FN(atc_oracle_step)(const REAL* S, int B, int N, const FN(orc_state_t) * st, const REAL* actions,
                        const FN(orc_out_t) * out, const atc_params_t* p) {
    if (!S || !st || !actions || !out || !p || B < 0 || N < 1 || N > ATC_MAX_AIRCRAFT) return -1;
    const size_t BN = (size_t)B * N;
    (void)BN;
    const double dtd = p->dt; /* SimParameters.timestep: a float64 like the reference's Python float (ABI 20) */
    const int discrete = (p->mode & ATC_M_DISCRETE) != 0;
    const REAL v_min = S[ATC_C_V_MIN], v_max = S[ATC_C_V_MAX], h_min = S[ATC_C_H_MIN], h_max = S[ATC_C_H_MAX];
    /* atc_gym.py:64-78: offset = (v_min, 0, 0); factor = (10,100,1) discrete | (v_max - v_min, h_max, 360) continuous */
    const REAL off[3] = {v_min, 0, 0};
    const REAL fac[3] = {discrete ? (REAL)10 : v_max - v_min, discrete ? (REAL)100 : h_max, discrete ? (REAL)1 : (REAL)360};
    const int n_mva = (int)S[ATC_H_N_MVA];
    const int n_noise = (int)S[ATC_H_N_NOISE];
    const int off_poly = (int)S[ATC_H_OFF_POLY];
#if ORC_FIXED_POS
    /* uniform terms of the fixed-point spec (include/atc_step.h, ABI 18), evaluated in float64 from the blob's fp32 constants */
    const FN(decode_t) dec_v = discrete ? FN(decode_consts)((double)fac[0], (double)off[0], 0.0, ORC_QV)
                                        : FN(decode_consts)((double)fac[0] / 2.0, (double)fac[0] / 2.0 + (double)off[0], 0.0, ORC_QV);
    const FN(decode_t) dec_p = discrete ? FN(decode_consts)((double)fac[2], (double)off[2], (double)ATC_PHI_FIX_OFFSET, ORC_QP)
                                        : FN(decode_consts)((double)fac[2] / 2.0, (double)fac[2] / 2.0 + (double)off[2],
                                                            (double)ATC_PHI_FIX_OFFSET, ORC_QP);
    const int32_t v_min_fix = FN(v_store)(v_min), v_max_fix = FN(v_store)(v_max);
    const int32_t rate_v = FN(rate_fix)((double)S[ATC_C_A_MAX], dtd, ORC_QV);      /* model.py:47-48: symmetric */
    const int32_t rate_p = FN(rate_fix)((double)S[ATC_C_PHIDOT_MAX], dtd, ORC_QP); /* model.py:49-50: symmetric */
    const int32_t discr_v = (int32_t)((double)S[ATC_C_ACT_DISCR] * ORC_QV), discr_p = (int32_t)((double)S[ATC_C_ACT_DISCR + 2] * ORC_QP);
    const double qd = dtd / 3600.0;
    const double dist_a = ldexp(qd, (int)lrint(log2((double)S[ATC_C_POS_SCALE])) - ATC_V_FIX_SHIFT); /* counts per speed count */
    if (!(0.1423 * dtd * (double)S[ATC_C_POS_SCALE] < 1073741824.0)) return -1; /* 512 kt: the displacement must fit 2^30 counts */
#endif

    /* envs are independent: the all-cores CPU baseline (bench.py) runs this loop under OpenMP; results do not depend on
     * the thread count (no cross-env state) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (B >= 64)
#endif
    for (int e = 0; e < B; ++e) {
        st->timesteps[e] += 1; /* atc_gym.py:135 */
        const int t = st->timesteps[e];
        const uint64_t act0 = st->active_mask[e];
        REAL r[ATC_MAX_AIRCRAFT];
        uint16_t fl[ATC_MAX_AIRCRAFT];
        REAL mva_h[ATC_MAX_AIRCRAFT];
        int mva_i[ATC_MAX_AIRCRAFT];

        /* pass 1: actions + kinematics for every active aircraft */
        for (int k = 0; k < N; ++k) {
            const size_t i = (size_t)e * N + k;
            fl[k] = 0;
            r[k] = 0;
            if (!((act0 >> k) & 1ull)) {
                fl[k] = ATC_F_INACTIVE;
                continue;
            }
            REAL reward = (REAL)(-0.05 * dtd); /* atc_gym.py:137 */
#if ORC_FIXED_POS
            /* fp32 spec (include/atc_step.h): speed / heading targets, rate limits and the discriminator in 32-bit fixed point
             * (ABI 18); the altitude in float64, the reference's own operations (ABI 20) */
            {
                const float av = actions[i * 3 + 0], ah = actions[i * 3 + 1], ap = actions[i * 3 + 2];
                { /* speed: model.py:60-80 */
                    const uint32_t tgt = FN(trunc_u32)(fma((double)av, dec_v.m, dec_v.c));
                    if (tgt < (uint32_t)v_min_fix || tgt > (uint32_t)v_max_fix) {
                        reward -= (REAL)1.0; /* atc_gym.py:312-315 */
                        fl[k] |= ATC_F_INVALID_V;
                    } else {
                        /* (differences of valid speeds and of the initial last_action 0 fit 32 bits: wrapping arithmetic) */
                        st->v[i] = (int32_t)((uint32_t)st->v[i] + (uint32_t)FN(clampi)((int32_t)(tgt - (uint32_t)st->v[i]), -rate_v, rate_v));
                        int32_t* la = &st->last_act[4 * i + 0];
                        const int32_t dd = (int32_t)(tgt - (uint32_t)*la);
                        if (!(dd > -discr_v && dd < discr_v)) st->actions_taken[e] += 1; /* atc_gym.py:305-306 */
                        *la = (int32_t)tgt;                                               /* atc_gym.py:311 */
                    }
                }
                { /* altitude: model.py:82-102 in float64, operation for operation (include/atc_step.h, ABI 20): the target is the
                   * reference's a * f / 2 + f / 2 + off (ONE rounding: the product of an fp32 action with f is exact), the refusals,
                   * the rate limits h_dot * timestep and the sum are its float64 operations */
                    const double td = discrete ? fma((double)ah, (double)fac[1], (double)off[1])
                                               : fma((double)ah, (double)fac[1] / 2.0, (double)fac[1] / 2.0 + (double)off[1]);
                    if (td < (double)h_min || td > (double)h_max) {
                        reward -= (REAL)1.0;
                        fl[k] |= ATC_F_INVALID_H;
                    } else {
                        double d = td - st->h[i];
                        d = fmin(d, (double)S[ATC_C_HDOT_MAX] * dtd);
                        d = fmax(d, (double)S[ATC_C_HDOT_MIN] * dtd);
                        st->h[i] = st->h[i] + d;
                        int32_t* la = &st->last_act[4 * i + 2];
                        double last;
                        memcpy(&last, la, sizeof last);
                        if (!(fabs(td - last) < (double)S[ATC_C_ACT_DISCR + 1])) st->actions_taken[e] += 1;
                        memcpy(la, &td, sizeof td);
                    }
                }
                { /* heading: model.py:104-120 — no validation, no wrap: 64-bit counts (include/atc_step.h, ABI 19) */
                    int clamped;
                    const int64_t tgt = FN(trunc_i64_phi)(fma((double)ap, dec_p.m, dec_p.c), &clamped);
                    if (clamped) fl[k] |= ATC_F_PHI_LIMIT;
                    int64_t* wide = st->phi_wide + 2 * i;
                    const int64_t P = FN(phi_load)(st->phi[i], wide[0]);
                    int64_t d = tgt - P;
                    d = d > (int64_t)rate_p ? (int64_t)rate_p : (d < -(int64_t)rate_p ? -(int64_t)rate_p : d);
                    FN(phi_put)(P + d, &st->phi[i], &wide[0]);
                    int32_t* la = &st->last_act[4 * i + 1];
                    const int64_t dd = tgt - FN(phi_load)(*la, wide[1]);
                    if (!(dd > -(int64_t)discr_p && dd < (int64_t)discr_p)) st->actions_taken[e] += 1;
                    FN(phi_put)(tgt, la, &wide[1]);
                }
            }
            /* model.py:122-129 Airplane.step, float64 from the fixed-point state */
            FN(advance)(S, dist_a, FN(is_wide)(st->phi[i]) ? FN(phi_wrap)(st->phi_wide[2 * i]) : st->phi[i], st->v[i], t, &st->x[i], &st->y[i]);
#else
            for (int c = 0; c < 3; ++c) {   /* atc_gym.py:139-141 -> _action_with_reward :299-316 */
                REAL a = actions[i * 3 + c];
                REAL tgt;
                if (discrete)
                    tgt = a * fac[c] + off[c]; /* atc_gym.py:329-330 */
                else
                    tgt = a * fac[c] / (REAL)2 + fac[c] / (REAL)2 + off[c]; /* atc_gym.py:333-335 */
                int valid = 1;
                if (c == 0) { /* model.py:60-80 */
                    if (tgt < v_min || tgt > v_max) valid = 0;
                    else {
                        REAL d = tgt - st->v[i];
                        d = R_MIN(d, S[ATC_C_A_MAX] * dtd);
                        d = R_MAX(d, S[ATC_C_A_MIN] * dtd);
                        st->v[i] = st->v[i] + d;
                    }
                } else if (c == 1) { /* model.py:82-102 */
                    if (tgt < h_min || tgt > h_max) valid = 0;
                    else {
                        REAL d = tgt - st->h[i];
                        d = R_MIN(d, S[ATC_C_HDOT_MAX] * dtd);
                        d = R_MAX(d, S[ATC_C_HDOT_MIN] * dtd);
                        st->h[i] = st->h[i] + d;
                    }
                } else { /* model.py:104-120: no validation, no wrap */
                    REAL d = tgt - st->phi[i];
                    d = R_MIN(d, S[ATC_C_PHIDOT_MAX] * dtd);
                    d = R_MAX(d, S[ATC_C_PHIDOT_MIN] * dtd);
                    st->phi[i] = st->phi[i] + d;
                }
                if (valid) {
                    REAL* la = &st->last_act[(size_t)c * BN + i];
                    if (!(R_ABS(tgt - *la) < S[ATC_C_ACT_DISCR + c])) st->actions_taken[e] += 1; /* atc_gym.py:305-306 */
                    *la = tgt;                                                                   /* atc_gym.py:311 */
                } else {
                    reward -= (REAL)1.0; /* atc_gym.py:312-315 */
                    fl[k] |= (c == 0) ? ATC_F_INVALID_V : ATC_F_INVALID_H;
                }
            }
            /* model.py:122-129 Airplane.step: rot_matrix(phi) . [0, (v/3600)*dt] */
            REAL dist = (st->v[i] / (REAL)3600) * dtd;
            REAL sn, cs;
            FN(sincos_heading)(st->phi[i], &sn, &cs);
            st->x[i] = FN(pos_advance)(S, st->x[i], sn * dist);
            st->y[i] = FN(pos_advance)(S, st->y[i], cs * dist);
#endif
            r[k] = reward;
        }

        /* pass 2: MVA floor (atc_gym.py:146-161) */
        int env_done = 0;
        for (int k = 0; k < N; ++k) {
            if (fl[k] & ATC_F_INACTIVE) continue;
            const size_t i = (size_t)e * N + k;
            int pi = FN(find_mva)(S, FN(pos_to_real)(S, 0, st->x[i]), FN(pos_to_real)(S, 1, st->y[i]));
            mva_i[k] = pi;
            if (pi >= 0) {
                REAL mva = S[off_poly + pi * ATC_P_WORDS + ATC_P_HEIGHT];
                mva_h[k] = mva;
                if (st->h[i] < (double)mva) { /* atc_gym.py:149: the float64 altitude against the integer height */
                    r[k] = (REAL)-200;
                    fl[k] |= ATC_F_BELOW_MVA;
                }
            } else {
                r[k] = (REAL)-50;
                fl[k] |= ATC_F_OUTSIDE;
                mva_h[k] = 0; /* atc_gym.py:161 */
            }
        }

        /* pass 3 (extension, no reference code): pairwise separation among aircraft active at step start */
        REAL min_sep = (REAL)1e30;
        for (int a = 0; a < N; ++a) {
            if (fl[a] & ATC_F_INACTIVE) continue;
            for (int b = a + 1; b < N; ++b) {
                if (fl[b] & ATC_F_INACTIVE) continue;
                const size_t ia = (size_t)e * N + a, ib = (size_t)e * N + b;
                REAL dx = FN(pos_to_real)(S, 0, st->x[ia]) - FN(pos_to_real)(S, 0, st->x[ib]);
                REAL dy = FN(pos_to_real)(S, 1, st->y[ia]) - FN(pos_to_real)(S, 1, st->y[ib]);
                REAL d2 = R_FMA(dx, dx, dy * dy); /* fused: the definition shared with the device kernel */
                REAL dh = R_ABS((REAL)st->h[ia] - (REAL)st->h[ib]); /* (fp32 spec: the altitudes rounded once, then the difference) */
                REAL d = R_SQRT(d2);
                if (d < min_sep) min_sep = d;
                if (d2 < (REAL)p->sep_nm * (REAL)p->sep_nm && dh < (REAL)p->sep_ft) {
                    fl[a] |= ATC_F_CONFLICT;
                    fl[b] |= ATC_F_CONFLICT;
                }
            }
        }
        if (out->min_sep) out->min_sep[e] = min_sep;

        /* pass 4: remaining override chain, observation, shaping (atc_gym.py:163-189) */
        REAL env_reward = 0;
        uint64_t act1 = act0;
        int any_won = 0;
        const int keep_active = (p->mode & ATC_M_KEEP_ACTIVE) != 0;
        for (int k = 0; k < N; ++k) {
            const size_t i = (size_t)e * N + k;
            float raw[10], nrm[10];
            if (fl[k] & ATC_F_INACTIVE) {
                for (int c = 0; c < 10; ++c) raw[c] = 0.f, nrm[c] = 0.f;
                if (out->raw_obs) memcpy(out->raw_obs + i * 10, raw, sizeof raw);
                memcpy(out->obs + i * 10, nrm, sizeof nrm);
                out->flags[i] = fl[k];
                if (out->ac_reward) out->ac_reward[i] = 0;
                if (out->mva) out->mva[i] = 0;
                continue;
            }
            if (fl[k] & ATC_F_CONFLICT) r[k] = (REAL)p->conflict_reward;
            const REAL px = FN(pos_to_real)(S, 0, st->x[i]), py = FN(pos_to_real)(S, 1, st->y[i]);
#if ORC_FIXED_POS
            const REAL phi_r = FN(phi_real)(FN(phi_eff)(st->phi[i], st->phi_wide[2 * i])), v_r = FN(v_real)(st->v[i]);
            const REAL phi_o = FN(phi_obs)(st->phi[i], st->phi_wide[2 * i]);
#else
            const REAL phi_r = st->phi[i], phi_o = st->phi[i], v_r = st->v[i];
#endif
#if ORC_FIXED_POS
            const double phi_c = (double)FN(phi_eff)(st->phi[i], st->phi_wide[2 * i]);
#else
            const double phi_c = (double)phi_r;
#endif
            if (FN(inside_corridor)(S, px, py, (REAL)st->h[i], phi_c)) { /* atc_gym.py:163-169 */
                int bonus = (p->timestep_limit - t) * 5;
                if (bonus < 0) bonus = 0;
                r[k] = (REAL)(10000 + bonus);
                fl[k] |= ATC_F_WON;
                any_won = 1;
                if (!keep_active) act1 &= ~(1ull << k); /* extension: handed over.  KEEP_ACTIVE: the reference's rule */
            }
            if (t > p->timestep_limit) { /* atc_gym.py:171-173 */
                r[k] = (REAL)-200;
                fl[k] |= ATC_F_TIMEOUT;
            }
            REAL d_faf, phi_rel_faf, on_gp;
            FN(get_state)(S, st->x[i], st->y[i], st->h[i], phi_r, phi_o, v_r, mva_h[k], raw, &d_faf, &phi_rel_faf, &on_gp);
            if (p->mode & ATC_M_REWARD_SHAPING) { /* atc_gym.py:179-185 */
                REAL pos = FN(reward_approach_position)(d_faf, S[ATC_C_PHI_TO_RWY], phi_rel_faf, S[ATC_C_WORLD_DIAG]);
                r[k] += pos;
                r[k] += FN(reward_approach_angle)(S[ATC_C_PHI_TO_RWY], phi_rel_faf, phi_r, pos);
                r[k] += FN(reward_glideslope)((REAL)st->h[i], on_gp, pos);
            }
            /* extension: noise-abatement areas (no reference code): inside polygon and below its ceiling */
            for (int q = 0; q < n_noise; ++q) {
                const REAL* rec = S + off_poly + (n_mva + q) * ATC_P_WORDS;
                if (rec[ATC_P_MINX] <= px && px <= rec[ATC_P_MAXX] && rec[ATC_P_MINY] <= py && py <= rec[ATC_P_MAXY] &&
                    st->h[i] < (double)rec[ATC_P_HEIGHT] &&
                    FN(ray_tracing)(px, py, S + (int)rec[ATC_P_VOFF], (int)rec[ATC_P_NVERT])) {
                    r[k] -= rec[ATC_P_PENALTY];
                    fl[k] |= ATC_F_NOISE;
                }
            }
            if (p->mode & ATC_M_NORMALIZE) FN(normalize)(S, raw, nrm);
            else memcpy(nrm, raw, sizeof raw);
            if (out->raw_obs) memcpy(out->raw_obs + i * 10, raw, sizeof raw);
            memcpy(out->obs + i * 10, nrm, sizeof nrm);
            out->flags[i] = fl[k];
            if (out->ac_reward) out->ac_reward[i] = r[k];
            if (out->mva) out->mva[i] = (mva_i[k] >= 0) ? (int32_t)mva_h[k] : -1;
            env_reward += r[k];
            if (fl[k] & (ATC_F_BELOW_MVA | ATC_F_OUTSIDE | ATC_F_CONFLICT | ATC_F_TIMEOUT)) env_done = 1;
        }
        /* every aircraft handed over (N = 1: the reference's win, atc_gym.py:169); KEEP_ACTIVE: any win ends the episode */
        const int env_won = keep_active ? any_won : (act1 == 0);
        if (env_won) env_done = 1;
        st->active_mask[e] = act1;
        st->total_reward[e] += env_reward; /* atc_gym.py:194-197 */
        out->reward[e] = env_reward;
        out->done[e] = (uint8_t)env_done;

        if (env_done && (p->mode & ATC_M_AUTO_RESET)) {
            st->ep_return[e] = st->total_reward[e];
            st->ep_length[e] = t;
            st->ep_actions[e] = st->actions_taken[e];
            /* every aircraft reached the corridor (N = 1: win_buffer.append(1), atc_gym.py:165) */
            st->win_bits[e] = ((st->win_bits[e] << 1) | (uint32_t)env_won) & 0x3ffu;
            if (out->term_obs) memcpy(out->term_obs + (size_t)e * N * 10, out->obs + (size_t)e * N * 10, sizeof(float) * 10 * N);
            FN(reset_env)(S, N, st, p, e, out->obs, 0);
        }
    }
    return 0;
}

/* ---- batched queries (for golden lattices G3/G4/G5) ------------------------------------------------------------------ */
int FN(atc_oracle_query_mva)(const REAL* S, int n, const REAL* x, const REAL* y, int32_t* out_h) {
    int off_poly = (int)S[ATC_H_OFF_POLY];
    for (int i = 0; i < n; ++i) {
        int pi = FN(find_mva)(S, x[i], y[i]);
        out_h[i] = pi >= 0 ? (int32_t)S[off_poly + pi * ATC_P_WORDS + ATC_P_HEIGHT] : -1;
    }
    return 0;
}
int FN(atc_oracle_query_corridor)(const REAL* S, int n, const REAL* x, const REAL* y, const REAL* h, const REAL* phi,
                                  int angle_only, uint8_t* out) {
    for (int i = 0; i < n; ++i)
        out[i] = (uint8_t)(angle_only ? FN(inside_corridor_angle)(S, x[i], y[i], FN(heading_counts_of)(phi[i]))
                                      : FN(inside_corridor)(S, x[i], y[i], h[i], FN(heading_counts_of)(phi[i])));
    return 0;
}
int FN(atc_oracle_query_shaping)(const REAL* S, int n, const REAL* d_faf, const REAL* phi_rel_faf, const REAL* phi_plane,
                                 const REAL* h, const REAL* on_gp, REAL* out3) {
    for (int i = 0; i < n; ++i) {
        REAL pos = FN(reward_approach_position)(d_faf[i], S[ATC_C_PHI_TO_RWY], phi_rel_faf[i], S[ATC_C_WORLD_DIAG]);
        out3[3 * i + 0] = pos;
        out3[3 * i + 1] = FN(reward_approach_angle)(S[ATC_C_PHI_TO_RWY], phi_rel_faf[i], phi_plane[i], pos);
        out3[3 * i + 2] = FN(reward_glideslope)(h[i], on_gp[i], pos);
    }
    return 0;
}
REAL FN(atc_oracle_relative_angle)(REAL a1, REAL a2) { return FN(relative_angle)(a1, a2); }
REAL FN(atc_oracle_sigmoid)(REAL d, REAL dmax) { return FN(sigmoid_distance)(d, dmax); }

#undef CAT_
#undef CAT
#undef FN
