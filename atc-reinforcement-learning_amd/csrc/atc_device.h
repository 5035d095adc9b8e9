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

#ifndef ATC_MVA_BATCH
#define ATC_MVA_BATCH 2  // edge records fetched per L2 round trip in dirty lookup cells (8 VGPRs each)
#endif

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

// sin/cos of a heading in DEGREES — the fp32 heading kinematics of include/atc_step.h (shared with the fp32 instantiation
// of the test oracle, so that positions are bit-identical).  The reduction is exact: k = rint(phi/90), t = phi - 90 k is
// computed by one fma and is exactly representable (|t| <= 45 + slop, a multiple of ulp(phi)); the polynomials (fitted on
// |t| <= 46.4 deg) have 8.5e-8 / 8.3e-8 max abs error in fp32 — the accuracy class of libm's sinf/cosf, at ~1/4 of the
// instructions of a generic radian sincosf (no Payne-Hanek path, no division).  Kinematics: model.py:122-129, 345-348.
__device__ __forceinline__ void sincos_deg(float phi, float* sn, float* cs) {
    const float k = rintf(phi * (1.0f / 90.0f));
    const float t = fmaf(-90.0f, k, phi);
    const float r = t * kDegToRad;
    const float r2 = r * r;
    const float sp = fmaf(fmaf(fmaf(ATC_SIN_C3, r2, ATC_SIN_C2), r2, ATC_SIN_C1), r2, 1.0f);
    const float s = sp * r;
    const float c = fmaf(fmaf(fmaf(fmaf(ATC_COS_C4, r2, ATC_COS_C3), r2, ATC_COS_C2), r2, ATC_COS_C1), r2, 1.0f);
    const int q = (int)k & 3;
    const float s1 = (q & 1) ? c : s;
    const float c1 = (q & 1) ? s : c;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}

// atan2 in DEGREES (np.degrees(np.arctan2(y, x)), atc_gym.py:289-292) — value-only: octant reduction to a = min/max in
// [0,1], 8-term odd polynomial (1.5e-7 rad max error), reciprocal instead of a division.
__device__ __forceinline__ float atan2_deg(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = (mx > 0.0f) ? mn * fast_rcp(mx) : 0.0f;
    const float s = a * a;
    float p = -0.004054565913975239f;
    p = fmaf(p, s, 0.021862955763936043f);
    p = fmaf(p, s, -0.0559123270213604f);
    p = fmaf(p, s, 0.0964219719171524f);
    p = fmaf(p, s, -0.1390862911939621f);
    p = fmaf(p, s, 0.19946566224098206f);
    p = fmaf(p, s, -0.33329859375953674f);
    p = fmaf(p, s, 0.9999993443489075f);
    float r = p * a;                                   // atan(min/max) in [0, pi/4]
    r = (ay > ax) ? (0.5f * kPi - r) : r;              // first quadrant
    r = (x < 0.0f) ? (kPi - r) : r;
    r = (y < 0.0f) ? -r : r;
    return r * kRadToDeg;
}

// Python float modulo by 360 (sign of the divisor), as used by relative_angle (model.py:340-342):  CPython computes
// r = fmod(a, 360) exactly and, if r has the wrong sign, r += 360 (one rounding).
// Here: q = floor(a * (1/360)) estimates the quotient (off by at most one either way), and r = fma(-360, q, a) is the
// EXACT value a - 360 q rounded once (a and 360 q are both multiples of ulp(a) below 2^24 ulp(a)).
//   * q exact        -> r is what CPython returns, including the a = -tiny case that rounds to 360.0
//   * q one too big  -> r is exact and negative; r + 360 rounds once, like CPython's `r += b`
//   * q one too small-> r in [360, 720) exactly, detected by a >= 360 (q + 1) (integers below 2^24: exact); r - 360 exact.
// No IEEE division (bit-identical to the fmodf-based oracle on every input, checked in tests/test_hip_parity.py).
__device__ __forceinline__ float py_mod360(float a) {
    const float q = floorf(a * (1.0f / 360.0f));
    float r = fmaf(-360.0f, q, a);
    if (r < 0.0f) r += 360.0f;
    else if (a >= (q + 1.0f) * 360.0f) r -= 360.0f;
    return r;
}

// model.py:340-342
__device__ __forceinline__ float relative_angle(float a1, float a2) { return py_mod360(a2 - a1 + 180.0f) - 180.0f; }
// The same for angles whose difference is known to lie in [-360, 720) (a bearing from atan2 against a runway heading):
// there CPython's fmod is the identity or one subtraction and only the sign fix-up remains — same result, half the work.
__device__ __forceinline__ float relative_angle_near(float a1, float a2) {
    float r = a2 - a1 + 180.0f;
    r = (r < 0.0f) ? r + 360.0f : ((r >= 360.0f) ? r - 360.0f : r);
    return r - 180.0f;
}
// v / 3600 (model.py:124), correctly rounded without the IEEE division sequence: q0 = v * RN(1/3600), one fma gives the
// exact remainder v - 3600 q0, a second folds it back.  Bit-identical to v / 3600.0f for every float with
// 2^-4 <= |v| < 2^16 and for 0 (checked exhaustively, 1.7e8 values; aircraft speeds are 100..300 kt).
__device__ __forceinline__ float div3600(float v) {
    constexpr float r = 1.0f / 3600.0f;
    const float q0 = v * r;
    return fmaf(fmaf(-q0, 3600.0f, v), r, q0);
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
    bool in_grid;  // false: beyond the padded bbox (also NaN) -> outside the airspace
};
// bit q set: the aircraft has to be tested against noise-abatement area q (all of them where there is no grid cell to ask)
__device__ __forceinline__ uint32_t noise_candidates(const float* __restrict__ grid, const MvaCell& c) {
    if (!grid || !c.in_grid) return 0xffffu;
    return (uint32_t)(int)fabsf(c.cell.x) >> 6;
}
// the grid header (origin, 1 / cell, columns, rows): read EARLY by the caller — as part of mva_cell_load it was a scalar-load
// round trip (plus a second one behind a short-circuit test) between the new position and the cell gather
struct GridHdr {
    float x0, y0, inv, nx, ny, off_pool;
};
__device__ __forceinline__ GridHdr grid_header(const float* __restrict__ grid) {
    GridHdr g = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (grid) {
        const float4 a = *reinterpret_cast<const float4*>(grid);   // ATC_G_X0, ATC_G_Y0, ATC_G_INV, ATC_G_NX
        const float2 b = *reinterpret_cast<const float2*>(grid + ATC_G_NY);   // ATC_G_NY, ATC_G_OFF_POOL
        g.x0 = a.x; g.y0 = a.y; g.inv = a.z; g.nx = a.w;
        g.ny = b.x; g.off_pool = b.y;
    }
    return g;
}
__device__ __forceinline__ MvaCell mva_cell_load(const float* __restrict__ grid, const GridHdr& g, float x, float y) {
    MvaCell c;
    c.cell = make_float2(0.0f, 0.0f);
    c.in_grid = false;
    if (grid) {
        const float fx = (x - g.x0) * g.inv;
        const float fy = (y - g.y0) * g.inv;
        c.in_grid = (fx >= 0.0f) & (fx < g.nx) & (fy >= 0.0f) & (fy < g.ny);
        // clamped index: the load is unconditional (no branch in front of it), the result is ignored when !in_grid
        const int ix = c.in_grid ? (int)fx : 0, iy = c.in_grid ? (int)fy : 0;
        // (24-bit multiply-add: rows and columns are far below 2^23; the 32-bit multiply is a quarter-rate instruction)
        // uniform base + 32-bit byte offset: the scalar-base addressing form, no 64-bit address arithmetic per lane
        c.cell = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(grid) +
                                                  (uint32_t)(ATC_G_HDR * 4 + 8 * (__mul24(iy, (int)g.nx) + ix)));
    }
    return c;
}
__device__ __forceinline__ int mva_resolve(const float* __restrict__ K, const float* __restrict__ grid, const GridHdr& gh,
                                           const MvaCell& c, float x, float y, float* height) {
    *height = 0.0f;
    if (grid) {
        if (!c.in_grid) return -1;
        const float2 cell = c.cell;
        const int code = (int)fabsf(cell.x) & 63;
        if (!(cell.x > 0.0f)) {  // clean cell: polygon + 1 (0 = outside the airspace), height
            *height = cell.y;
            return code - 1;
        }
        const int n = code;
#ifdef ATC_ABLATE_WALK
        *height = 3000.0f;  // developer-only timing ablation: dirty cells answered without walking their edge list
        return 1;
#endif
        // Records are fetched in batches of kBatch (both 16-byte halves of each, all loads issued before the first use):
        // one L2 round trip per batch instead of one per record.  Indices past the list are clamped (loads stay in
        // bounds) and their records ignored.
        const char* pool = reinterpret_cast<const char*>(grid + (int)gh.off_pool);   // uniform
        const uint32_t rec0 = 32u * (uint32_t)(int)cell.y;                          // this lane's first record, bytes
        constexpr int kBatch = ATC_MVA_BATCH;
        bool inside = false;
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

// model.py:212-231 Corridor._inside_corridor_angle.
// The reference compares min_angle = arccos(dir_rwy . dir_plane) [radians] with relative_angle [degrees]:
//     min_angle <= rel <= 45.
// In exact arithmetic arccos(dir_rwy . dir_plane) = |rel| * pi/180, so the window is  0 <= rel <= 45 :
//   rel > 0  : |rel| pi/180 < rel            -> lower bound holds
//   rel < 0  : lower bound fails
//   rel == 0 : holds iff the rounded dot product is exactly 1.0 — the reference's own rounding luck, evaluated once on
//              the host with the reference's expression (ATC_C_ALIGNED_OK).
// This drops sin/cos/acos from the path and removes the fp32 noise band (|rel| < 3.5e-4 deg) a literal fp32
// transcription would have; it differs from the float64 reference only for 0 < rel < ~1e-6 deg.
__device__ __forceinline__ bool angle_window(const float* __restrict__ K, float rel) {
    return (rel > 0.0f && rel <= K[ATC_C_FAF_ANGLE]) || (rel == 0.0f && K[ATC_C_ALIGNED_OK] != 0.0f);
}
__device__ __forceinline__ bool inside_corridor_angle(const float* __restrict__ K, float x, float y, float phi) {
    // model.py:224-229: `if tri1 and window: True / elif tri2 and window: True / False`
    const float to_runway = K[ATC_C_PHI_TO_RWY];
    if (ray_tracing(x, y, K + ATC_C_TRI_1, 4) && angle_window(K, relative_angle(to_runway, phi))) return true;
    if (ray_tracing(x, y, K + ATC_C_TRI_2, 4) && angle_window(K, relative_angle(phi, to_runway))) return true;
    return false;
}

// model.py:188-210 Corridor.inside_corridor
__device__ __forceinline__ bool inside_corridor(const float* __restrict__ K, float x, float y, float h, float phi) {
    // exact early-out: a point the crossing test accepts lies within the ring's bounds (precomputed on the host)
    // (one 16-byte scalar load and bitwise ands: four short-circuit tests were four dependent scalar-load round trips)
    static_assert(ATC_C_TRI_BBOX % 4 == 0, "bounds must be 16-byte aligned");
    const float4 bb = *reinterpret_cast<const float4*>(K + ATC_C_TRI_BBOX);
    if (!((x >= bb.x) & (x <= bb.z) & (y >= bb.y) & (y <= bb.w))) return false;
    if (!ray_tracing(x, y, K + ATC_C_TRI_H, 4)) return false;
    const float fx = K[ATC_C_FAF_X], fy = K[ATC_C_FAF_Y], nx = K[ATC_C_NRM_X], ny = K[ATC_C_NRM_Y];
    const float t = (x - fx) * nx + (y - fy) * ny;
    const float px = fx + t * nx, py = fy + t * ny;
    const float dx = px - K[ATC_C_RWY_X], dy = py - K[ATC_C_RWY_Y];
    const float nrm = sqrtf(dx * dx + dy * dy);
    const float h_max = nrm * K[ATC_C_GS_TAN] * K[ATC_C_NM_TO_FT] + K[ATC_C_RWY_H];
    if (!(h <= h_max)) return false;
    return inside_corridor_angle(K, x, y, phi);
}

// atc_gym.py:17-19  (1 - tanh(4 d/dmax - 2)) / 2  ==  1 / (1 + exp(2 (4 d/dmax - 2)))   [exact identity]
__device__ __forceinline__ float sigmoid_distance(float d, float inv_d_max) {
    const float z2 = fmaf(8.0f * inv_d_max, d, -4.0f);  // 2 * (4 d/dmax - 2)
    return fast_rcp(1.0f + fast_exp(z2));
}

struct Shaping {
    float pos, ang, gs;
};
// atc_gym.py:199-260: _reward_approach_position, _reward_approach_angle, _reward_glideslope
__device__ __forceinline__ Shaping shaping_rewards(const float* __restrict__ K, float d_faf, float phi_rel_faf, float plane_to_runway,
                                                   float h, float on_gp) {
    // plane_to_runway = relative_angle(phi_to_runway, phi_plane): the caller already has it as obs[9]
    const float to_rwy = K[ATC_C_PHI_TO_RWY];
    Shaping r;
    const float rel_faf = relative_angle_near(to_rwy, phi_rel_faf);  // phi_rel_faf = atan2 in [-180, 180], to_rwy in [0, 360)
    const float u = fabsf(rel_faf) * (1.0f / 180.0f);
    r.pos = sigmoid_distance(d_faf, fast_rcp(K[ATC_C_WORLD_DIAG])) * (u * fast_sqrt(u)) * 0.8f;  // u ** 1.5
    const float side = (rel_faf > 0.0f) ? 1.0f : ((rel_faf < 0.0f) ? -1.0f : 0.0f);  // np.sign
    const float q = (side * plane_to_runway - 22.5f) * (1.0f / 202.0f);
    float m = -(q * q) + 1.0f;  // (-(q ** 2.0) + 1.0) ** 32.0 by five squarings (even power: sign-safe)
    m = m * m;
    m = m * m;
    m = m * m;
    m = m * m;
    m = m * m;
    r.ang = m * r.pos * 1.2f;
    r.gs = sigmoid_distance(fabsf(h - on_gp), 1.0f / 36000.0f) * r.pos * 0.8f;
    return r;
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
// the fp32 position every formula of the reference sees: (float)(origin + fix * 2^-k), ONE rounding
__device__ __forceinline__ float pos_to_real(const float* __restrict__ K, int axis, int p) {
    return (float)fma((double)p, (double)K[ATC_C_POS_INV], (double)K[ATC_C_POS_X0 + axis]);
}
// model.py:122-129: x += d with d computed in fp32; the grid advances by rint(d 2^k) counts, saturating
__device__ __forceinline__ int pos_advance(const float* __restrict__ K, int p, float d) {
    float c = d * K[ATC_C_POS_SCALE];
    c = fminf(fmaxf(c, -1073741824.0f), 1073741824.0f);
    return sat_add(p, (int)rintf(c));
}
// entry point (fp32 nm) -> grid
__device__ __forceinline__ int pos_spawn(const float* __restrict__ K, int axis, float v) {
    float c = (v - K[ATC_C_POS_X0 + axis]) * K[ATC_C_POS_SCALE];
    c = fminf(fmaxf(c, -2147483648.0f), 2147483520.0f);
    return (int)rintf(c);
}
// faf - position (atc_gym.py:289-297): exact integer difference on the grid -> fp32 relative precision near the FAF
__device__ __forceinline__ float pos_to_faf(const float* __restrict__ K, int axis, int p) {
    const int faf = (int)K[ATC_C_FAF_FIX + 2 * axis] * 65536 + (int)K[ATC_C_FAF_FIX + 2 * axis + 1];
    return (float)sat_sub(faf, p) * K[ATC_C_POS_INV];
}

struct Aircraft {
    int x, y;         // position grid counts
    float h, phi, v;
};

// atc_gym.py:346-348 + model.py:13-52: aircraft k of env e enters at an entry point.
// Lattice mode reads the per-slot record precomputed in the blob (one 16-byte load); random mode maps a 64-bit draw to
// (entry, level) by multiply-shift — no integer division on the reset path.
__device__ __forceinline__ Aircraft spawn(const float* __restrict__ K, const atc_params_t& p, int e, int k, int episode) {
    Aircraft a;
    a.v = kVInit;
    if (!(p.mode & ATC_M_RANDOM_ENTRY)) {
        const float4 rec = *reinterpret_cast<const float4*>(K + (int)K[ATC_H_OFF_SLOT] + 4 * k);
        a.x = pos_spawn(K, 0, rec.x);
        a.y = pos_spawn(K, 1, rec.y);
        a.phi = rec.z;
        a.h = rec.w;
        return a;
    }
    const uint32_t n_entry = (uint32_t)(int)K[ATC_H_N_ENTRY];
    const uint64_t u = draw(p.seed, (uint32_t)e, (uint32_t)episode, (uint32_t)k);
    const int ei = (int)__umulhi((uint32_t)(u & 0xffffffffu), n_entry);
    const float* rec = K + (int)K[ATC_H_OFF_ENTRY] + ei * ATC_E_WORDS;
    const int li = (int)__umulhi((uint32_t)(u >> 32), (uint32_t)(int)rec[ATC_E_NLEV]);
    a.x = pos_spawn(K, 0, rec[ATC_E_X]);
    a.y = pos_spawn(K, 1, rec[ATC_E_Y]);
    a.phi = rec[ATC_E_PHI];
    a.h = rec[ATC_E_LEV0 + li] * 100.0f;
    return a;
}

struct Obs {
    float o[ATC_OBS_DIM];
    float d_faf, phi_rel_faf, on_gp;
};
// atc_gym.py:262-297 _get_state.  (px, py) = grid position, (x, y) = its fp32 value
__device__ __forceinline__ Obs get_state(const float* __restrict__ K, int px, int py, float x, float y, float h, float phi,
                                         float v, float mva) {
    Obs r;
    const float to_faf_x = pos_to_faf(K, 0, px);
    const float to_faf_y = pos_to_faf(K, 1, py);
    r.d_faf = fast_sqrt(fmaf(to_faf_x, to_faf_x, to_faf_y * to_faf_y));   // np.hypot (value-only)
    r.phi_rel_faf = atan2_deg(to_faf_y, to_faf_x);                        // np.degrees(np.arctan2) (value-only)
    r.on_gp = 318.4f * r.d_faf + K[ATC_C_FAF_MVA] - 200.0f;
    r.o[0] = x;
    r.o[1] = y;
    r.o[2] = h;
    r.o[3] = phi;
    r.o[4] = v;
    r.o[5] = h - mva;
    r.o[6] = r.on_gp;
    r.o[7] = r.d_faf;
    r.o[8] = r.phi_rel_faf;
    r.o[9] = relative_angle(K[ATC_C_PHI_TO_RWY], phi);
    return r;
}

}  // namespace atc
