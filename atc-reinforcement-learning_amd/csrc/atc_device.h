// atc_device.h — gfx950 device functions of the batched AtcGym.step() path (fp32 arithmetic, fixed-point position state).
//
// Every function cites the reference lines it implements (path:line in fvalka/atc-reinforcement-learning).
// Integer outputs (done / flag words / counters / MVA index) are required to match the fp32 CPU oracle bit-for-bit,
// so comparisons that decide them use the same operation order as the reference and the translation unit is compiled
// with -ffp-contract=off (explicit fmaf only where exactness is argued).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/atc_step.h"

#define ATC_MVA_BATCH 2  // edge records fetched per L2 round trip in dirty lookup cells (8 VGPRs each)

namespace atc {

// Aircraft performance limits and the action discriminator are hard-coded in the reference (Airplane.__init__ defaults,
// model.py:13,45-50, never overridden by atc_gym.py:347; atc_gym.py:84).  They are compile-time constants here too — the
// blob still carries them and atc_scenario_create() refuses a blob whose values differ.
constexpr float kVMin = 100.0f, kVMax = 300.0f, kHMin = 0.0f, kHMax = 38000.0f;
constexpr float kAMin = -5.0f, kAMax = 5.0f, kHDotMin = -41.0f, kHDotMax = 15.0f, kPhiDotMin = -3.0f, kPhiDotMax = 3.0f;
constexpr float kDiscrV = 5.0f, kDiscrH = 50.0f, kDiscrPhi = 0.5f, kVInit = 250.0f;

constexpr float kPi = 3.14159265358979323846f;
constexpr float kDegToRad = (float)(3.14159265358979323846 / 180.0);
constexpr float kRadToDeg = (float)(180.0 / 3.14159265358979323846);

// ---- value-only fast math (results feed observations / shaping rewards, tolerance 1e-5; never a flag) ---------------
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }    // v_rcp_f32, 1 ulp
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }  // v_sqrt_f32, 1 ulp
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// ---- speed and heading: 32-bit fixed point (include/atc_step.h, ABI 18) -------------------------------------------------
//   kt = v_fix 2^-23 (unsigned counts),  deg = 180 + phi_fix 2^-23 (signed counts).  Targets, rate limits and the action
//   discriminator are integer arithmetic; the step's displacement is float64 from the fixed-point state (advance() below).
constexpr float kFixInv = 1.0f / 8388608.0f;                      // 2^-23 for both
constexpr uint32_t kVMinFix = 838860800u, kVMaxFix = 2516582400u;  // 100 / 300 kt (model.py:13) in counts
constexpr uint32_t kVInitFix = 2097152000u;                        // 250 kt (atc_gym.py:348)
constexpr int kDiscrVFix = 41943040, kDiscrPhiFix = 4194304;       // 5 kt, 0.5 deg (atc_gym.py:84) in counts
static_assert(ATC_V_FIX_SHIFT == 23 && ATC_PHI_FIX_SHIFT == 23, "fixed-point formats of include/atc_step.h");
// float64 -> integer exactly as the spec defines it = as the hardware does it: truncation toward zero, saturation at both
// ends, NaN -> 0 (a C cast is undefined outside the integer's range, so the instructions are named)
__device__ __forceinline__ int cvt_i32_f64(double x) {
    int i;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(i) : "v"(x));
    return i;
}
__device__ __forceinline__ uint32_t cvt_u32_f64(double x) {
    uint32_t i;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(i) : "v"(x));
    return i;
}
// the fp32 speed / heading every other formula of the reference sees (observation, relative angles, corridor window)
__device__ __forceinline__ float v_real(uint32_t f) { return (float)f * kFixInv; }
__device__ __forceinline__ float phi_real(int f) { return fmaf((float)f, kFixInv, ATC_PHI_FIX_OFFSET); }
// ---- the unbounded heading (include/atc_step.h, ABI 19): exact counts behind a saturating 32-bit field ------------------
// A heading (or last heading target) whose 32-bit field is INT32_MIN / INT32_MAX is WIDE: its exact counts are the integer-valued
// float64 side word atc_state_t.phi_wide[i][0] ([1] for the last target).  Everything below runs only in wavefronts that hold such
// an aircraft or receive a heading target beyond the 32-bit range — behind wave-uniform tests, off the step's straight line.
// All of it is float64 arithmetic on integers below 2^53: exact, and one instruction per operation on gfx950.
__device__ __forceinline__ bool is_wide(int f) { return f == INT32_MAX || f == INT32_MIN; }
// heading target in counts: trunc(a m + c) clamped to +-2^52, NaN -> 0 (the spec of include/atc_step.h)
__device__ __forceinline__ double phi_target_wide(double x, bool* clamped) {
    x = (x == x) ? __builtin_trunc(x) : 0.0;
    *clamped = __builtin_fabs(x) > ATC_PHI_LIMIT;
    return __builtin_fmin(__builtin_fmax(x, -ATC_PHI_LIMIT), ATC_PHI_LIMIT);
}
// a WIDE heading wrapped to within half a turn of zero: what the kinematics and every relative angle use (they are periodic)
__device__ __forceinline__ int phi_wrap(double P) {
    const double k = __builtin_rint(P * ATC_PHI_INV_TURN);
    return cvt_i32_f64(__builtin_fma(k, -ATC_PHI_TURN, P));
}
// observation word 3 (atc_gym.py:269) of a WIDE heading: exact in float64, ONE rounding
__device__ __forceinline__ float phi_obs_wide(double P) { return (float)((double)ATC_PHI_FIX_OFFSET + P * (1.0 / 8388608.0)); }
// state placed from outside (entry points): nearest count
__device__ __forceinline__ int phi_store(float p) { return cvt_i32_f64(__builtin_rint(((double)p - (double)ATC_PHI_FIX_OFFSET) * 8388608.0)); }

// Uniform float64 constants of the heading kinematics (include/atc_step.h: ATC_KIN_*) and the step's distance scale — kernel
// arguments like every other uniform term: a float64 literal cannot be an instruction operand, so written into the code each
// would be two scalar moves per use (or a register pair held across the step loop); as an argument group the fourteen arrive
// with two scalar loads.
struct alignas(16) QKin {
    double inv180, neg_half_turn;
    double s0, s1, s2, s3, s4, s5;
    double c1, c2, c3, c4, c5;
    double dist_neg;   // -(dt / 3600) 2^(k_pos - 23): position-grid counts per speed count and step, NEGATED (see advance)
    double pad[2];
};
// Airplane.step (model.py:122-129, 345-348) in float64 from the fixed-point state — the spec of include/atc_step.h, shared bit
// for bit with the fp32 instantiation of the test oracle:  k = rint(phi_fix / (180 2^23)), t = phi_fix - k 180 2^23 (exact),
// two Horner polynomials in u = t^2 scaled to counts, the distance negated when k is even (phi = 180 (1 + k) + t: an even k is
// an odd number of half turns — the host passes the NEGATED scale and the sign is flipped back when k is odd: one shift and one
// xor), and dithered rounding: counts += floor(displacement + u11), u11 = the low 11 bits of the env's time step bit-reversed,
// as a fraction (van der Corput): the round-off of a constant displacement (a straight leg) cancels to O(log n) counts over n
// steps instead of adding up.  The dither sits in the low bits of the "magic" addend 1.5 2^41 (assembled from integers); the
// integer part of the sum is bits 11..42 of the result: one fma and one v_alignbit per axis.
// a * b + c with the addend c taken from a scalar register pair (v_fma_f64, VOP3: one scalar operand).  Written out because the
// compiler otherwise picks the two-address form (v_fmac_f64) and copies every uniform coefficient into a vector register pair
// first: two v_mov per Horner step, twenty per aircraft-step, in a kernel that is bound by instruction issue.
__device__ __forceinline__ double fma_sc(double a, double b, double c_uniform) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
    return d;
}
__device__ __forceinline__ double mul_sc(double a, double b_uniform) {
    double d;
    asm("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(b_uniform));
    return d;
}
// VG: the constants of `q` live in VECTOR registers (the latency-bound instantiation for batches of at most two wavefronts per
// SIMD, csrc/atc_step.hip: LAT): plain fma / mul then take them as they are — the scalar-operand forms below exist to keep the
// compiler from copying SCALAR constants into vector register pairs.
template <bool VG = false>
__device__ __forceinline__ void advance(const QKin& q, int phi_fix, uint32_t v_fix, int t_step, int& x, int& y) {
    const double pd = (double)phi_fix;
    const double kd = __builtin_rint(VG ? pd * q.inv180 : mul_sc(pd, q.inv180));
    double t;
    if (VG) t = __builtin_fma(kd, q.neg_half_turn, pd);
    else asm("v_fma_f64 %0, %1, %2, %3" : "=v"(t) : "v"(kd), "s"(q.neg_half_turn), "v"(pd));
    const double u = t * t;
    auto fma_c = [](double a, double b, double c) { return VG ? __builtin_fma(a, b, c) : fma_sc(a, b, c); };
    // (the two Horner chains are written interleaved: each step waits for the previous one of its own chain only)
    double sp = fma_c(u, q.s5, q.s4), cp = fma_c(u, q.c5, q.c4);
    sp = fma_c(sp, u, q.s3);
    cp = fma_c(cp, u, q.c3);
    sp = fma_c(sp, u, q.s2);
    cp = fma_c(cp, u, q.c2);
    sp = fma_c(sp, u, q.s1);
    cp = fma_c(cp, u, q.c1);
    sp = fma_c(sp, u, q.s0);
    const double cs = __builtin_fma(cp, u, 1.0);
    const double sn = sp * t;
    const double dneg = VG ? (double)v_fix * q.dist_neg : mul_sc((double)v_fix, q.dist_neg);
    const uint32_t flip = (uint32_t)cvt_i32_f64(kd) << 31;
    const double dist = __hiloint2double(__double2hiint(dneg) ^ (int)flip, __double2loint(dneg));
    const double magic = __hiloint2double((int)ATC_DITHER_MAGIC_HI, (int)(__builtin_bitreverse32((uint32_t)t_step) >> 21));
    const double rx = __builtin_fma(sn, dist, magic), ry = __builtin_fma(cs, dist, magic);
    const int nx = (int)__builtin_amdgcn_alignbit((uint32_t)__double2hiint(rx), (uint32_t)__double2loint(rx), 11u);
    const int ny = (int)__builtin_amdgcn_alignbit((uint32_t)__double2hiint(ry), (uint32_t)__double2loint(ry), 11u);
    x = __builtin_elementwise_add_sat(x, nx);
    y = __builtin_elementwise_add_sat(y, ny);
}

// atan2 in DEGREES (np.degrees(np.arctan2(y, x)), atc_gym.py:289-292) — value-only: octant reduction to a = min/max in
// [0,1], a * P(a^2) with a 7-term polynomial fitted in degrees for the error RELATIVE to max(1 deg, result) — raw (reset)
// observations are compared with 1e-5 max(1, |value|): 7.8e-7 max in fp32 —, a reciprocal instead of a division, the sign of
// y copied in with one bit-field insert.
__device__ __forceinline__ float atan2_deg(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(fmaxf(ax, ay), 1e-30f), mn = fminf(ax, ay);   // (0, 0) -> a = 0 -> 0 like np.arctan2
    const float a = mn * fast_rcp(mx);
    const float s = a * a;
    float p = 0.4501694142818451f;
    p = fmaf(p, s, -2.119493007659912f);
    p = fmaf(p, s, 4.8039751052856445f);
    p = fmaf(p, s, -7.726700305938721f);
    p = fmaf(p, s, 11.3909912109375f);
    p = fmaf(p, s, -19.094654083251953f);
    p = fmaf(p, s, 57.29574203491211f);
    float r = p * a;                                   // atan(min/max) in [0, 45] degrees
    r = (ay > ax) ? (90.0f - r) : r;                   // first quadrant
    r = (x < 0.0f) ? (180.0f - r) : r;
    return __uint_as_float((__float_as_uint(r) & 0x7fffffffu) | (__float_as_uint(y) & 0x80000000u));   // r >= 0: copysign
}

// Python float modulo by 360 (sign of the divisor), as used by relative_angle (model.py:340-342):  CPython computes
// r = fmod(a, 360) exactly and, if r has the wrong sign, r += 360 (one rounding).
// Here: q = floor(a * RN(1/360)) estimates the quotient, and r = fma(-360, q, a) is the EXACT value a - 360 q rounded
// once (a and 360 q are both multiples of ulp(a) below 2^24 ulp(a)).
//   * q exact        -> r is what CPython returns, including the a = -tiny case that rounds to 360.0
//   * q one too big  -> r is exact and negative; r + 360 rounds once, like CPython's `r += b`.  Happens only just below a
//                       multiple of 360 from a = 1799.9999 upwards (first of 50 184 floats below 2^25).
//   * q too small    -> never: RN(1/360) > 1/360, so a RN(1/360) lies on the far side of the true quotient from zero, and
//                       the rounding of the product is monotonic across the (representable) integer below it.
// Both statements checked exhaustively for every float with 2^-12 <= |a| < 2^25 (tools/check_mod360.py).
// No IEEE division (bit-identical to the fmodf-based oracle, tests/test_hip_parity.py).
__device__ __forceinline__ float py_mod360(float a) {
    const float q = floorf(a * (1.0f / 360.0f));
    const float r = fmaf(-360.0f, q, a);
    return (r < 0.0f) ? r + 360.0f : r;
}

// model.py:340-342
__device__ __forceinline__ float relative_angle(float a1, float a2) { return py_mod360(a2 - a1 + 180.0f) - 180.0f; }
// relative_angle for a VALUE-ONLY argument that is itself an approximation (the bearing to the FAF out of atan2_deg in the
// shaping terms): w - 360 floor((w + 180) / 360) with w = a2 - a1 — four operations for any range, no compares.  Agrees with
// the exact form above to a few ulp(w) except within ~2e-5 deg of the wrap at +-180, where the reference's own result jumps
// by 360 — a discontinuity that the fp32 bearing crosses at slightly different positions anyway.  Not used where the
// arguments can sit ON the wrap exactly (integer headings against phi_to_runway) and not for anything that decides a flag.
__device__ __forceinline__ float relative_angle_value(float a1, float a2) {
    const float w = a2 - a1;
    const float q = floorf(fmaf(w, 1.0f / 360.0f, 0.5f));
    return fmaf(q, -360.0f, w);
}
// model.py:318-337 ray_tracing over a closed ring (x,y interleaved in LDS, n vertices, first == last).
// Same inequality set and evaluation order; the reference's n+1-th iteration re-visits ring[0] from ring[n-1]
// (identical points for closed rings -> never counted) and is reproduced for rings that are not closed.
__device__ __forceinline__ bool ray_tracing(float x, float y, const float* ring, int n) {
    bool inside = false;
    float p1x = ring[0], p1y = ring[1];
    for (int i = 1; i <= n; ++i) {
        const int k = (i == n) ? 0 : i;
        const float2 p2 = *reinterpret_cast<const float2*>(ring + 2 * k);
        const float p2x = p2.x, p2y = p2.y;
        if (y > fminf(p1y, p2y) && y <= fmaxf(p1y, p2y) && x <= fmaxf(p1x, p2x)) {
            // p1y != p2y is implied here (y > min and y <= max)
            const float xints = (y - p1y) * (p2x - p1x) / (p2y - p1y) + p1x;
            if (p1x == p2x || x <= xints) inside = !inside;
        }
        p1x = p2x;
        p1y = p2y;
    }
    return inside;
}

// model.py:282-292 Airspace.find_mva: first polygon in list order whose inclusive bounds contain the point and whose
// ray_tracing is true; returns its index (-1 = "Outside of airspace") and its height through *height.
//
// K = the sector blob in global memory.  Uniform-index reads of it become scalar loads (SGPRs); the per-lane reads
// below (polygon records, edge records) are L2/L1-resident gathers and sit on rare paths only.
//
// With the lookup grid (atc_hip/scenario.py:build_grid) a CLEAN cell answers directly (index + height in the cell) and
// a DIRTY cell lists, polygon by polygon in priority order, those edges whose crossing depends on where in the cell the
// point lies; edges that no point of the cell can cross are dropped, edges that every point of the cell crosses are
// folded into the polygon's BASE parity — walking the list with the reference's formula and xor-ing BASE yields the
// same parity as ray_tracing over the full ring.  CERTAIN edges lie > 1e-3 nm to the right of the whole cell:
// x <= xints holds whatever the rounding of xints.  A polygon ends with its LAST edge (bounds contain the whole cell)
// or with a terminator record holding its bounds (model.py:286).
__device__ __forceinline__ bool in_bounds(const float* rec, float x, float y) {
    const float4 b = *reinterpret_cast<const float4*>(rec);  // minx, miny, maxx, maxy
    return b.x <= x && x <= b.z && b.y <= y && y <= b.w;
}
// The lookup is split so that the caller can issue the cell gather early and resolve it later: the L2 round trip then
// overlaps independent work (in the step kernel: the whole separation scan) instead of stalling the wavefront.
struct MvaCell {
    float2 cell;   // (+-(code + 64 noise mask), first_record | height): > 0 dirty, code = n_records; <= 0 clean, code = polygon + 1
};
// bit q set: the aircraft has to be tested against noise-abatement area q (all of them where there is no grid to ask)
__device__ __forceinline__ uint32_t noise_candidates(const float* __restrict__ grid, const MvaCell& c) {
    if (!grid) return 0xffffu;
    return ((uint32_t)(int)fabsf(c.cell.x) >> 6) & 0xffffu;
}
// the bounds of the corridor's horizontal triangle meet the aircraft's cell (bit 22 of the cell code): only then can
// Runway.inside_corridor (model.py:198) accept the point
__device__ __forceinline__ bool corridor_candidate(const MvaCell& c) { return ((uint32_t)(int)fabsf(c.cell.x) & (1u << 22)) != 0u; }
// The grid header (origin, 1 / cell, columns, rows, record pool) — uniform.  The step kernel receives it with its arguments
// (evaluated on the host: gfx950 has no scalar float conversion); the query kernel reads it from the blob.
struct GridHdr {
    float x0, y0, inv;
    int nx, nx_last, ny_last;   // columns, columns - 1, rows - 1
    int off_pool;               // words from the grid start to the record pool
};
__host__ __device__ inline GridHdr grid_header(const float* grid) {
    GridHdr g = {0.0f, 0.0f, 0.0f, 1, 0, 0, 0};
    if (grid) {
        g.x0 = grid[ATC_G_X0]; g.y0 = grid[ATC_G_Y0]; g.inv = grid[ATC_G_INV];
        g.nx = (int)grid[ATC_G_NX]; g.nx_last = g.nx - 1; g.ny_last = (int)grid[ATC_G_NY] - 1;
        g.off_pool = (int)grid[ATC_G_OFF_POOL];
    }
    return g;
}
// float -> int32 as the hardware defines it for EVERY input (v_cvt_i32_f32: truncation, saturation at both ends, NaN -> 0).
// A C cast is undefined outside the int range, so the instruction is named explicitly where that range matters.
__device__ __forceinline__ int cvt_i32_sat(float f) {
    int i;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(i) : "v"(f));
    return i;
}
// Cell of a point.  The grid's outermost ring of cells is CLEAN and OUTSIDE the airspace with no noise-area candidate
// (atc_hip/scenario.py:build_grid pads for it, atc_scenario_create checks it), so a point beyond the grid — or NaN — simply
// takes the border cell its clamped index names: (int) of a float saturates, NaN converts to 0, and one unsigned minimum per
// axis clamps both ends (a negative index is a huge unsigned).  No range compare, no select; the answer of such a point
// is "outside the airspace" (model.py:289) as it must be.
__device__ __forceinline__ MvaCell mva_cell_load(const float* __restrict__ grid, const GridHdr& g, float x, float y) {
    MvaCell c;
    c.cell = make_float2(0.0f, 0.0f);
    if (grid) {
        const float fx = (x - g.x0) * g.inv;
        const float fy = (y - g.y0) * g.inv;
        const uint32_t ix = min((uint32_t)cvt_i32_sat(fx), (uint32_t)g.nx_last);
        const uint32_t iy = min((uint32_t)cvt_i32_sat(fy), (uint32_t)g.ny_last);
        // (24-bit multiply-add: rows and columns are far below 2^23; the 32-bit multiply is a quarter-rate instruction)
        // uniform base + 32-bit byte offset: the scalar-base addressing form, no 64-bit address arithmetic per lane
        c.cell = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(grid) +
                                                  (uint32_t)(ATC_G_HDR * 4 + 8 * __umul24(iy, (uint32_t)g.nx) + 8 * ix));
    }
    return c;
}
// A dirty cell's first record (the LINE record of a split cell), requested ahead of its use: the second dependent L2 round trip of
// the lookup then overlaps the observation arithmetic instead of stalling the wavefront.  Wave-uniform: if any lane's cell is dirty,
// EVERY lane requests a record (lanes in clean cells record 0 — one shared cache line): no per-lane branch around the loads.
struct MvaPre {
    float4 g, m;
};
__device__ __forceinline__ uint32_t first_record(const MvaCell& c) {   // byte offset in the pool (0 for clean cells)
    return (c.cell.x > 0.0f) ? 32u * (uint32_t)(int)c.cell.y : 0u;
}
__device__ __forceinline__ MvaPre mva_prefetch(const float* __restrict__ grid, const GridHdr& gh, const MvaCell& c) {
    MvaPre p;
    p.g = p.m = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (grid && __builtin_amdgcn_ballot_w64(c.cell.x > 0.0f) != 0ull) {
        const char* pool = reinterpret_cast<const char*>(grid + gh.off_pool);
        const uint32_t rec0 = first_record(c);
        p.g = *reinterpret_cast<const float4*>(pool + rec0);
        p.m = *reinterpret_cast<const float4*>(pool + rec0 + 16u);
    }
    return p;
}
// The walk of a dirty cell's ordinary records (rec0 = its first, n of them): the reference's crossing test per edge record, polygon by
// polygon in priority order (see above).  Returns the polygon index (-1: none) and its height.
template <int kBatch>
__device__ __forceinline__ int mva_walk(const char* __restrict__ pool, uint32_t rec0, int n, float x, float y, float* height) {
    *height = 0.0f;
    bool inside = false;
    // Records are fetched in batches of kBatch (both 16-byte halves of each, all loads issued before the first use):
    // one L2 round trip per batch instead of one per record.  Indices past the list are clamped (loads stay in
    // bounds) and their records ignored.
    for (int base = 0; base < n; base += kBatch) {
        float4 g[kBatch], m[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int e = min(base + u, n - 1);
            g[u] = *reinterpret_cast<const float4*>(pool + (rec0 + 32u * (uint32_t)e));
            m[u] = *reinterpret_cast<const float4*>(pool + (rec0 + 32u * (uint32_t)e + 16u));
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            if (base + u < n) {
                const int code = (int)m[u].w;
                bool decide, ok = true;
                if (code & ATC_GE_TERM) {  // g = polygon bounds (model.py:286)
                    decide = true;
                    ok = g[u].x <= x && x <= g[u].z && g[u].y <= y && y <= g[u].w;
                } else {
                    // the three cheap tests of model.py:328-330 (m.x / m.y = min / max of the edge's y, precomputed)
                    if (y > m[u].x && y <= m[u].y && x <= fmaxf(g[u].x, g[u].z)) {
                        bool cross = (code & ATC_GE_CERTAIN) != 0;
                        if (!cross) {
                            const float xints = (y - g[u].y) * (g[u].z - g[u].x) / (g[u].w - g[u].y) + g[u].x;
                            cross = (g[u].x == g[u].z) || x <= xints;
                        }
                        inside = inside != cross;
                    }
                    decide = (code & ATC_GE_LAST) != 0;
                }
                if (decide) {
                    if ((inside != ((code & ATC_GE_BASE) != 0)) && ok) {
                        *height = m[u].z;
                        return code >> 4;
                    }
                    inside = false;
                }
            }
        }
    }
    return -1;
}
template <int kBatch = ATC_MVA_BATCH>
__device__ __forceinline__ int mva_resolve(const float* __restrict__ K, const float* __restrict__ grid, const GridHdr& gh,
                                           const MvaCell& c, float x, float y, float* height, const MvaPre* pre = nullptr) {
    *height = 0.0f;
    if (grid) {
        const float2 cell = c.cell;
        const uint32_t bits = (uint32_t)(int)fabsf(cell.x);
        const int code = (int)(bits & 63u);
        const bool dirty = cell.x > 0.0f;
        int res = code - 1;               // clean cell: polygon + 1 (0 = outside the airspace), height
        float h = dirty ? 0.0f : cell.y;
        // Some lane's cell is cut by a border (wave-uniform test; nine wavefronts in ten at the headline size).  SPLIT cells
        // (atc_hip/scenario.py:_line_split — four dirty cells in five) are answered by their LINE record: the line, a margin that
        // covers every rounding of the reference's x-intersection, and the answer of either side — one 32-byte fetch, one fma, two
        // compares, selects; written for the whole wavefront without a per-lane branch (lanes in clean cells evaluate record 0 and
        // discard the result).  Only a point inside the margin band, or in a cell with a vertex or a second border, walks the
        // ordinary records behind it — behind a second wave-uniform test.
        if (__builtin_amdgcn_ballot_w64(dirty) != 0ull) {
            const char* pool = reinterpret_cast<const char*>(grid + gh.off_pool);   // uniform
            const uint32_t rec0 = first_record(c);
            // (p1x, p1y, dx/dy, margin | left polygon + 1, height, right polygon + 1, height; already requested by mva_prefetch?)
            const float4 g = pre ? pre->g : *reinterpret_cast<const float4*>(pool + rec0);
            const float4 m = pre ? pre->m : *reinterpret_cast<const float4*>(pool + rec0 + 16u);
            const float xl = fmaf(y - g.y, g.z, g.x);
            const bool line = dirty && (bits & ATC_G_CELL_LINE) != 0u;
            const bool left = x < xl - g.w, right = x > xl + g.w;
            const bool decided = line && (left || right);
            res = decided ? (int)(left ? m.x : m.z) - 1 : res;
            h = decided ? (left ? m.y : m.w) : h;
            const bool walk = dirty && !decided;
            if (__builtin_amdgcn_ballot_w64(walk) != 0ull) {
                if (walk) res = mva_walk<kBatch>(pool, rec0 + (line ? 32u : 0u), code, x, y, &h);
            }
        }
        *height = h;
        return res;
    }
    const float* tab = K + (int)K[ATC_H_OFF_POLY];
    const int n_mva = (int)K[ATC_H_N_MVA];
    for (int p = 0; p < n_mva; ++p) {
        const float* rec = tab + p * ATC_P_WORDS;
        if (in_bounds(rec, x, y) && ray_tracing(x, y, K + (int)rec[ATC_P_VOFF], (int)rec[ATC_P_NVERT])) {
            *height = rec[ATC_P_HEIGHT];
            return p;
        }
    }
    return -1;
}
__device__ __forceinline__ int find_mva(const float* __restrict__ K, const float* __restrict__ grid, float x, float y,
                                        float* height) {
    const GridHdr gh = grid_header(grid);
    const MvaCell c = mva_cell_load(grid, gh, x, y);
    return mva_resolve(K, grid, gh, c, x, y, height);
}

// ---- LDS-resident lookup table (ABI 21; include/atc_step.h: atc_scenario_attach_lds_table, built by atc_hip/scenario.py:
//      build_lds_table).  For the latency-bound multi-step launches of ONE-aircraft envs (64 envs per wavefront, one wavefront per
//      SIMD): there the lookup grid's gather is 0.8 us of exposed wait in every step and its dirty cells a second / third dependent
//      trip in 40 % of the wavefront-steps (profiles/experiments/README.md, round 6).  A workgroup stages the table once per
//      launch; a step then reads a 16-bit level-1 code (0.5 nm cells), a 16-bit sub-cell code (8 x 8 per refined cell) and — for a
//      cell ONE border line splits — that line's 32-byte record, all from LDS: no vector-memory wait on the step's chain.  A lane in
//      a RESIDUAL sub-cell (a vertex or a second border inside it: 0.2 % of the aircraft, one wavefront-step in twelve) walks that
//      sub-cell's own edge records — one trip to global memory for the lanes concerned; a point inside a line's margin band (one
//      wavefront-step in sixty) sends its WHOLE wavefront to the global grid for this step.  Same answers as the ordered polygon
//      scan either way (tests/test_lds_table.py: the numpy restatement against the fp32 oracle on the CPU, the kernels on the GPU).
struct LdsTab {             // kernel argument: where the table lives in global memory + its header terms (host-checked)
    const uint4* src;       // 16-byte pieces; nullptr: no table attached
    int n16;
    float x0, y0, inv;      // level-1 origin and 1 / cell
    int nx, nx_last, ny_last;
    int off_l1, off_sub, off_line, off_hts, off_resid;   // byte offsets from the table's start (the staged part)
    const char* pool;       // edge records of the RESIDUAL sub-cells (global memory, the lookup grid's record format)
};
struct LdsCode {
    uint32_t c1, sub;       // level-1 code, sub-cell index sy * 8 + sx
};
// first half (with the kinematics): bin the point like mva_cell_load does (clamped indices: the outermost ring is clean and outside)
__device__ __forceinline__ LdsCode lds_cell_load(const char* tab, const LdsTab& t, float x, float y) {
    const float fx = (x - t.x0) * t.inv;
    const float fy = (y - t.y0) * t.inv;
    const uint32_t ix = min((uint32_t)cvt_i32_sat(fx), (uint32_t)t.nx_last);
    const uint32_t iy = min((uint32_t)cvt_i32_sat(fy), (uint32_t)t.ny_last);
    const uint32_t sx = min((uint32_t)cvt_i32_sat((fx - (float)ix) * 8.0f), 7u);
    const uint32_t sy = min((uint32_t)cvt_i32_sat((fy - (float)iy) * 8.0f), 7u);
    LdsCode c;
    c.c1 = *reinterpret_cast<const uint16_t*>(tab + ((uint32_t)t.off_l1 + 2u * (__umul24(iy, (uint32_t)t.nx) + ix)));
    c.sub = sy * 8u + sx;
    return c;
}
// second half: polygon index (-1: outside) + height like mva_resolve.  *walk != 0: a RESIDUAL sub-cell — the lane walks that word's
// records (lds_walk); *band: inside a LINE record's margin band — no answer here (the caller asks the grid)
__device__ __forceinline__ int lds_resolve(const char* tab, const LdsTab& t, const LdsCode& lc, float x, float y, float* height,
                                           uint32_t* walk, bool* band) {
    const uint32_t c1 = lc.c1;
    const bool is_sub = ((c1 >> 13) & 3u) == (uint32_t)ATC_LDS_SUB;
    const uint32_t sidx = is_sub ? (c1 & 0x1fffu) * 64u + lc.sub : 0u;   // (lanes without a refined cell read sub-cell 0 and discard it)
    const uint32_t c2 = *reinterpret_cast<const uint16_t*>(tab + ((uint32_t)t.off_sub + 2u * sidx));
    const uint32_t c = is_sub ? c2 : c1;
    const uint32_t kind = (c >> 13) & 3u, pay = c & 0x1fffu;
    const bool clean = kind == (uint32_t)ATC_LDS_CLEAN, is_line = kind == (uint32_t)ATC_LDS_LINE;
    const uint32_t li = is_line ? pay : 0u;
    // LINE record (atc_hip/scenario.py:_line_split): p1x, p1y, dx/dy, margin | left polygon + 1, height, right polygon + 1, height
    const float4 g = *reinterpret_cast<const float4*>(tab + ((uint32_t)t.off_line + 32u * li));
    const float4 m = *reinterpret_cast<const float4*>(tab + ((uint32_t)t.off_line + 32u * li + 16u));
    const bool resid = kind == (uint32_t)ATC_LDS_RESID;
    // ONE 4-byte read for both: a clean cell's height or a residual sub-cell's walk word (the sections are neighbours)
    const uint32_t w4 = *reinterpret_cast<const uint32_t*>(tab + (resid ? (uint32_t)t.off_resid + 4u * pay : (uint32_t)t.off_hts + 4u * (clean ? pay : 0u)));
    const float hc = __uint_as_float(w4);
    const float xl = fmaf(y - g.y, g.z, g.x);
    const bool left = x < xl - g.w, right = x > xl + g.w;
    const bool decided = is_line && (left || right);
    *height = clean ? hc : (left ? m.y : m.w);
    *walk = resid ? w4 : 0u;    // (a walk word is never 0: n_records >= 1)
    *band = is_line && !decided;
    return (clean ? (int)pay : (int)(left ? m.x : m.z)) - 1;
}
// the walk of a RESIDUAL sub-cell's records (word = first record | n << 24): ONE batch of four records covers nine lists in ten
__device__ __forceinline__ int lds_walk(const LdsTab& t, uint32_t word, float x, float y, float* height) {
    return mva_walk<4>(t.pool, 32u * (word & 0xffffffu), (int)(word >> 24), x, y, height);
}
__device__ __forceinline__ bool lds_corridor_candidate(const LdsCode& lc) { return (lc.c1 & 0x8000u) != 0u; }

// model.py:212-231 Corridor._inside_corridor_angle.
// The reference compares min_angle = arccos(dir_rwy . dir_plane) [radians] with relative_angle [degrees]:
//     min_angle <= rel <= 45.
// In exact arithmetic arccos(dir_rwy . dir_plane) = |rel| * pi/180, so the window is  0 <= rel <= 45 :
//   rel > 0  : |rel| pi/180 < rel            -> lower bound holds
//   rel < 0  : lower bound fails
//   rel == 0 : holds iff the rounded dot product is exactly 1.0 — the reference's own rounding luck, evaluated once on
//              the host with the reference's expression (ATC_C_ALIGNED_OK).
// This drops sin/cos/acos from the path.  Round 5: rel is evaluated EXACTLY, from the heading's counts — the wrapped difference
// heading - to_runway in units of 2^-23 deg, float64 arithmetic on integers (the reduction of include/atc_step.h) — because its SIGN
// decides a flag: an aircraft told to fly the runway heading holds it to within the fp32 rounding of its action (340 -/+ 1e-5 deg), which
// an fp32 heading (ulp 3e-5 at 340) cannot tell from 340 — the reference's float64 can (G11: the intercepts flown at heading - 360).
// It differs from the float64 reference only for 0 < rel < ~1e-6 deg (where the reference's own arccos noise decides).
// `pc`: heading counts as a float64 (any value congruent to the heading modulo a turn: the window is periodic).
__device__ __forceinline__ double to_runway_counts(const float* __restrict__ K) {
    return __builtin_rint(((double)K[ATC_C_PHI_TO_RWY] - (double)ATC_PHI_FIX_OFFSET) * 8388608.0);
}
__device__ __forceinline__ double rel_counts(double d) {   // d wrapped to within half a turn of zero (exact)
    return __builtin_fma(__builtin_rint(d * ATC_PHI_INV_TURN), -ATC_PHI_TURN, d);
}
__device__ __forceinline__ bool angle_window(const float* __restrict__ K, double rel) {
    return (rel > 0.0 && rel <= (double)K[ATC_C_FAF_ANGLE] * 8388608.0) || (rel == 0.0 && K[ATC_C_ALIGNED_OK] != 0.0f);
}
__device__ __forceinline__ bool inside_corridor_angle(const float* __restrict__ K, float x, float y, double pc) {
    // model.py:224-229: `if tri1 and window: True / elif tri2 and window: True / False`
    const double q = to_runway_counts(K);
    if (ray_tracing(x, y, K + ATC_C_TRI_1, 4) && angle_window(K, rel_counts(pc - q))) return true;   // relative_angle(to_runway, phi)
    if (ray_tracing(x, y, K + ATC_C_TRI_2, 4) && angle_window(K, rel_counts(q - pc))) return true;   // relative_angle(phi, to_runway)
    return false;
}
// a heading given in degrees (the query entry points): its nearest count
__device__ __forceinline__ double heading_counts_of(float phi_deg) {
    return __builtin_rint(((double)phi_deg - (double)ATC_PHI_FIX_OFFSET) * 8388608.0);
}

__device__ __forceinline__ float4 tri_bbox(const float* __restrict__ K) {
    static_assert(ATC_C_TRI_BBOX % 4 == 0, "bounds must be 16-byte aligned");
    return *reinterpret_cast<const float4*>(K + ATC_C_TRI_BBOX);
}
// model.py:188-210 Corridor.inside_corridor
__device__ __forceinline__ bool inside_corridor(const float* __restrict__ K, const float4& bb, float x, float y, float h, double pc) {
    // exact early-out: a point the crossing test accepts lies within the ring's bounds bb = ATC_C_TRI_BBOX (precomputed on
    // the host; the step kernel has them among its arguments, so the test waits for no load)
    if (!((x >= bb.x) & (x <= bb.z) & (y >= bb.y) & (y <= bb.w))) return false;
    if (!ray_tracing(x, y, K + ATC_C_TRI_H, 4)) return false;
    const float fx = K[ATC_C_FAF_X], fy = K[ATC_C_FAF_Y], nx = K[ATC_C_NRM_X], ny = K[ATC_C_NRM_Y];
    const float t = (x - fx) * nx + (y - fy) * ny;
    const float px = fx + t * nx, py = fy + t * ny;
    const float dx = px - K[ATC_C_RWY_X], dy = py - K[ATC_C_RWY_Y];
    const float nrm = sqrtf(dx * dx + dy * dy);
    const float h_max = nrm * K[ATC_C_GS_TAN] * K[ATC_C_NM_TO_FT] + K[ATC_C_RWY_H];
    if (!(h <= h_max)) return false;
    return inside_corridor_angle(K, x, y, pc);
}

// Uniform constants of the observation / shaping stage.  The step kernel receives them precomputed on the host with its
// arguments (kernel arguments live in SGPRs; evaluated in the kernel, uniform float arithmetic occupies the vector unit
// in every lane: gfx950 has no scalar float ALU); reset / observe / query kernels derive them from the blob.
struct alignas(8) ObsConst {
    int faf_x, faf_y;   // the FAF on the position grid (ATC_C_FAF_FIX)
    float pos_inv;      // nm per grid count (ATC_C_POS_INV)
    float to_rwy;       // phi_to_runway (ATC_C_PHI_TO_RWY)
    float on_gp_c;      // faf_mva - 200: on_gp_altitude = 318.4 d_faf + faf_mva - 200 (atc_gym.py:279-287)
    float sig_a;        // 8 log2(e) / world_diag, see sigmoid2
};
__host__ __device__ inline ObsConst obs_const(const float* K) {
    ObsConst c;
    c.faf_x = (int)K[ATC_C_FAF_FIX] * 65536 + (int)K[ATC_C_FAF_FIX + 1];
    c.faf_y = (int)K[ATC_C_FAF_FIX + 2] * 65536 + (int)K[ATC_C_FAF_FIX + 3];
    c.pos_inv = K[ATC_C_POS_INV];
    c.to_rwy = K[ATC_C_PHI_TO_RWY];
    c.on_gp_c = K[ATC_C_FAF_MVA] - 200.0f;
    c.sig_a = (8.0f * 1.44269504088896341f) / K[ATC_C_WORLD_DIAG];
    return c;
}
constexpr float kSigB = -4.0f * 1.44269504088896341f;
constexpr float kSigGs = (8.0f * 1.44269504088896341f) / 36000.0f;   // glideslope: d_max = 36000 (atc_gym.py:236)

// atc_gym.py:17-19  (1 - tanh(4 d/dmax - 2)) / 2  ==  1 / (1 + exp(2 (4 d/dmax - 2)))  ==  1 / (1 + 2^(a d + kSigB)) with
// a = 8 log2(e) / dmax   [exact identities; value-only]
__device__ __forceinline__ float sigmoid2(float d, float a) {
    return fast_rcp(1.0f + __builtin_amdgcn_exp2f(fmaf(d, a, kSigB)));
}

struct Shaping {
    float pos, ang, gs;
};
struct ShapingCore {
    float pos;   // _reward_approach_position
    float m;     // (-(q ** 2) + 1) ** 32: _reward_approach_angle = m * pos * 1.2
    float sg;    // sigmoid of the glideslope error: _reward_glideslope = sg * pos * 0.8
};
// atc_gym.py:199-260: _reward_approach_position, _reward_approach_angle, _reward_glideslope (value-only arithmetic)
__device__ __forceinline__ ShapingCore shaping_core(const ObsConst& c, float d_faf, float phi_rel_faf, float plane_to_runway,
                                                     float h, float on_gp) {
    // plane_to_runway = relative_angle(phi_to_runway, phi_plane): the caller already has it as obs[9]
    ShapingCore r;
    const float rel_faf = relative_angle_value(c.to_rwy, phi_rel_faf);
    const float u = fabsf(rel_faf) * (1.0f / 180.0f);
    r.pos = sigmoid2(d_faf, c.sig_a) * (u * fast_sqrt(u)) * 0.8f;   // u ** 1.5
    // np.sign(rel_faf) * plane_to_runway as a sign-bit xor.  (np.sign(0) = 0 is not reproduced: there u = 0, so pos = 0 and
    // the angle term m * pos * 1.2 is 0 whatever the sign factor — m is finite.)
    const float sp = __uint_as_float(__float_as_uint(plane_to_runway) ^ (__float_as_uint(rel_faf) & 0x80000000u));
    const float q = fmaf(sp, 1.0f / 202.0f, -22.5f / 202.0f);
    float m = fmaf(-q, q, 1.0f);   // (-(q ** 2.0) + 1.0) ** 32.0 by five squarings (even power: sign-safe)
    m = m * m;
    m = m * m;
    m = m * m;
    m = m * m;
    r.m = m * m;
    r.sg = sigmoid2(fabsf(h - on_gp), kSigGs);
    return r;
}
__device__ __forceinline__ Shaping shaping_rewards(const ObsConst& c, float d_faf, float phi_rel_faf, float plane_to_runway,
                                                   float h, float on_gp) {
    const ShapingCore k = shaping_core(c, d_faf, phi_rel_faf, plane_to_runway, h, on_gp);
    Shaping r;
    r.pos = k.pos;
    r.ang = k.m * k.pos * 1.2f;
    r.gs = k.sg * k.pos * 0.8f;
    return r;
}
// pos + ang + gs as pos * (1 + 1.2 m + 0.8 sg): what the step adds to the reward (atc_gym.py:179-185)
__device__ __forceinline__ float shaping_total(const ShapingCore& k) {
    return k.pos * fmaf(k.sg, 0.8f, fmaf(k.m, 1.2f, 1.0f));
}

// counter-based RNG for entry draws: integer-only, identical to the oracle's (oracle/atc_oracle_impl.h: mix64/draw)
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t draw(uint64_t seed, uint32_t env, uint32_t episode, uint32_t slot) {
    const uint64_t z = mix64(seed ^ ((uint64_t)env << 32 | (uint64_t)episode));
    return mix64(z ^ (uint64_t)slot);
}

// ---- aircraft positions: 32-bit fixed point on the sector's position grid (include/atc_step.h "Aircraft positions") ----
__device__ __forceinline__ int sat_add(int a, int b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ int sat_sub(int a, int b) { return __builtin_elementwise_sub_sat(a, b); }
// the fp32 position every formula of the reference sees: (float)(origin + fix * 2^-k), ONE rounding (the sum is exact in
// float64: a 32-bit integer scaled by a power of two plus a small integer)
__device__ __forceinline__ float pos_to_real(const float* __restrict__ K, int axis, int p) {
    return (float)fma((double)p, (double)K[ATC_C_POS_INV], (double)K[ATC_C_POS_X0 + axis]);
}
// the same from the exponent k and the origin as kernel arguments: conversion, exponent adjustment, addition, conversion —
// each takes its uniform operand from a scalar register (the fma form needs two: one goes through vector registers first)
__device__ __forceinline__ float pos_to_real(int neg_k, double origin, int p) {
    return (float)(__builtin_ldexp((double)p, neg_k) + origin);
}
// entry point (fp32 nm) -> grid
__device__ __forceinline__ int pos_spawn(const float* __restrict__ K, int axis, float v) {
    float c = (v - K[ATC_C_POS_X0 + axis]) * K[ATC_C_POS_SCALE];
    c = fminf(fmaxf(c, -2147483648.0f), 2147483520.0f);
    return (int)rintf(c);
}
// faf - position (atc_gym.py:289-297): exact integer difference on the grid -> fp32 relative precision near the FAF
__device__ __forceinline__ float pos_to_faf(int faf, float pos_inv, int p) { return (float)sat_sub(faf, p) * pos_inv; }

struct Aircraft {
    int x, y;         // position grid counts
    double h;         // altitude [ft]: the reference's float64 (include/atc_step.h, ABI 20)
    int phi;          // heading, fixed point (deg = 180 + phi 2^-23)
    uint32_t v;       // speed, fixed point (kt = v 2^-23)
};

// atc_gym.py:346-351,365 + model.py:13-52: aircraft k of env e enters at an entry point — from the blob's SPAWN RECORDS
// (include/atc_step.h: ATC_H_OFF_SPAWN): the fixed-point state and the raw reset observation of an aircraft placed at a lattice
// slot / an entry point were evaluated once by the sector compiler, so a reset is four 16-byte gathers instead of two position
// conversions, a heading conversion and a whole _get_state (square root, atan2, modulo).  Under the measurement protocol an
// env of 16 aircraft resets every ~25 steps — a wavefront meets a reset in one step out of six.
// Lattice mode reads record k; random mode maps a 64-bit draw to (entry, level) by multiply-shift — no integer division on
// the reset path —, reads the entry's record and puts the level's altitude into the state and into observation words 2 and 5.
// obs == nullptr: state only.
__device__ __forceinline__ Aircraft spawn(const float* __restrict__ K, int off_spawn, const atc_params_t& p, int e, int k,
                                          int episode, float* obs) {
    Aircraft a;
    a.v = kVInitFix;
    uint32_t rec = (uint32_t)k;
    float h_level = 0.0f;
    const bool random = (p.mode & ATC_M_RANDOM_ENTRY) != 0;
    if (random) {
        const uint32_t n_entry = (uint32_t)(int)K[ATC_H_N_ENTRY];
        const uint64_t u = draw(p.seed, (uint32_t)e, (uint32_t)episode, (uint32_t)k);
        const int ei = (int)__umulhi((uint32_t)(u & 0xffffffffu), n_entry);
        const float* er = K + (int)K[ATC_H_OFF_ENTRY] + ei * ATC_E_WORDS;
        const int li = (int)__umulhi((uint32_t)(u >> 32), (uint32_t)(int)er[ATC_E_NLEV]);
        h_level = er[ATC_E_LEV0 + li] * 100.0f;
        rec = (uint32_t)(ATC_MAX_AIRCRAFT + ei);
    }
    const char* r = reinterpret_cast<const char*>(K + off_spawn) + rec * (ATC_SPAWN_WORDS * 4u);
    const int4 st = *reinterpret_cast<const int4*>(r);
    a.x = st.x;
    a.y = st.y;
    a.h = (double)(random ? h_level : __int_as_float(st.z));   // (flight level x 100 ft: an integer)
    a.phi = st.w;
    if (obs) {
        const float4 o0 = *reinterpret_cast<const float4*>(r + 16), o1 = *reinterpret_cast<const float4*>(r + 32);
        const float2 o2 = *reinterpret_cast<const float2*>(r + 48);
        obs[0] = o0.x; obs[1] = o0.y; obs[2] = random ? h_level : o0.z; obs[3] = o0.w;
        obs[4] = o1.x; obs[5] = random ? h_level : o1.y; obs[6] = o1.z; obs[7] = o1.w;
        obs[8] = o2.x; obs[9] = o2.y;
    }
    return a;
}

// observation word 5 (atc_gym.py:267,276): np.float32 of the float64 difference h - mva (mva: an integer height, 0 outside / at reset)
__device__ __forceinline__ float alt_above(double h, float mva) { return (float)(h - (double)mva); }

struct Obs {
    float o[ATC_OBS_DIM];
    float d_faf, phi_rel_faf, on_gp;
};
// atc_gym.py:262-297 _get_state.  (px, py) = grid position, (x, y) = its fp32 value
// phi: the heading the relative angle sees; phi_obs: observation word 3 — the same number unless the heading is WIDE
// h = (float)altitude, h_above = (float)(altitude - mva) with the difference taken in float64 like the reference's (see alt_above)
__device__ __forceinline__ Obs get_state(const ObsConst& c, int px, int py, float x, float y, float h, float phi, float phi_obs, float v,
                                         float h_above) {
    Obs r;
    const float to_faf_x = pos_to_faf(c.faf_x, c.pos_inv, px);
    const float to_faf_y = pos_to_faf(c.faf_y, c.pos_inv, py);
    r.d_faf = fast_sqrt(fmaf(to_faf_x, to_faf_x, to_faf_y * to_faf_y));   // np.hypot (value-only)
    r.phi_rel_faf = atan2_deg(to_faf_y, to_faf_x);                        // np.degrees(np.arctan2) (value-only)
    r.on_gp = fmaf(318.4f, r.d_faf, c.on_gp_c);
    r.o[0] = x;
    r.o[1] = y;
    r.o[2] = h;
    r.o[3] = phi_obs;
    r.o[4] = v;
    r.o[5] = h_above;
    r.o[6] = r.on_gp;
    r.o[7] = r.d_faf;
    r.o[8] = r.phi_rel_faf;
    r.o[9] = relative_angle(c.to_rwy, phi);   // exact form: headings and phi_to_runway are often integers, i.e. ON the wrap
    return r;
}

}  // namespace atc
