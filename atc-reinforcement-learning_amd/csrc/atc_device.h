// atc_device.h — gfx950 device functions of the batched AtcGym.step() path (fp32 arithmetic, fp64 position state).
//
// Every function cites the reference lines it implements (path:line in fvalka/atc-reinforcement-learning).
// Integer outputs (done / flag words / counters / MVA index) are required to match the fp32 CPU oracle bit-for-bit,
// so comparisons that decide them use the same operation order as the reference and the translation unit is compiled
// with -ffp-contract=off (explicit fmaf only where exactness is argued).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/atc_step.h"

namespace atc {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kDegToRad = (float)(3.14159265358979323846 / 180.0);
constexpr float kRadToDeg = (float)(180.0 / 3.14159265358979323846);

// Python float modulo by 360 (sign of the divisor), as used by relative_angle (model.py:340-342).
// a - 360*floor(a/360) evaluated with one fma is the exactly rounded value of (fmod(a,360) [+360]), i.e. what
// CPython computes; the quotient can only be off by +1 when a/360 rounds up to an integer, fixed by the r < 0 branch
// (then r is exact and r + 360 rounds once, like CPython's `r += b`).
__device__ __forceinline__ float py_mod360(float a) {
    float q = floorf(a / 360.0f);
    float r = fmaf(-360.0f, q, a);
    if (r < 0.0f) r += 360.0f;
    return r;
}

// model.py:340-342
__device__ __forceinline__ float relative_angle(float a1, float a2) { return py_mod360(a2 - a1 + 180.0f) - 180.0f; }

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
// ray_tracing is true; -1 = "Outside of airspace".  With a lookup grid (global memory, L2-resident) clean cells answer
// directly and dirty cells restrict the ordered scan to their candidate polygons (identical results by construction,
// see atc_hip/scenario.py:build_grid).
__device__ __forceinline__ int find_mva(const float* S, const float* __restrict__ grid, float x, float y) {
    const int n_mva = (int)S[ATC_H_N_MVA];
    uint32_t cand = (n_mva >= 32) ? 0xffffffffu : ((1u << n_mva) - 1u);
    if (grid) {
        const float fx = (x - grid[ATC_G_X0]) * grid[ATC_G_INV];
        const float fy = (y - grid[ATC_G_Y0]) * grid[ATC_G_INV];
        const float nx = grid[ATC_G_NX], ny = grid[ATC_G_NY];
        if (!(fx >= 0.0f && fx < nx && fy >= 0.0f && fy < ny)) return -1;  // beyond the padded bbox (also NaN)
        const float c = grid[ATC_G_HDR + (int)fy * (int)nx + (int)fx];
        if (c < ATC_GRID_MASK_BASE) return (int)c - 1;
        cand = (uint32_t)(c - ATC_GRID_MASK_BASE);
    }
    const float* tab = S + (int)S[ATC_H_OFF_POLY];
    while (cand) {
        const int p = __builtin_ctz(cand);
        cand &= cand - 1u;
        const float* rec = tab + p * ATC_P_WORDS;
        if (rec[ATC_P_MINX] <= x && x <= rec[ATC_P_MAXX] && rec[ATC_P_MINY] <= y && y <= rec[ATC_P_MAXY]) {
            if (ray_tracing(x, y, S + (int)rec[ATC_P_VOFF], (int)rec[ATC_P_NVERT])) return p;
        }
    }
    return -1;
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
__device__ __forceinline__ bool angle_window(const float* S, float rel) {
    return (rel > 0.0f && rel <= S[ATC_C_FAF_ANGLE]) || (rel == 0.0f && S[ATC_C_ALIGNED_OK] != 0.0f);
}
__device__ __forceinline__ bool inside_corridor_angle(const float* S, float x, float y, float phi) {
    const float to_runway = S[ATC_C_PHI_TO_RWY];
    if (ray_tracing(x, y, S + ATC_C_TRI_1, 4)) return angle_window(S, relative_angle(to_runway, phi));
    if (ray_tracing(x, y, S + ATC_C_TRI_2, 4)) return angle_window(S, relative_angle(phi, to_runway));
    return false;
}

// model.py:188-210 Corridor.inside_corridor
__device__ __forceinline__ bool inside_corridor(const float* S, float x, float y, float h, float phi) {
    if (!ray_tracing(x, y, S + ATC_C_TRI_H, 4)) return false;
    const float fx = S[ATC_C_FAF_X], fy = S[ATC_C_FAF_Y], nx = S[ATC_C_NRM_X], ny = S[ATC_C_NRM_Y];
    const float t = (x - fx) * nx + (y - fy) * ny;
    const float px = fx + t * nx, py = fy + t * ny;
    const float dx = px - S[ATC_C_RWY_X], dy = py - S[ATC_C_RWY_Y];
    const float nrm = sqrtf(dx * dx + dy * dy);
    const float h_max = nrm * S[ATC_C_GS_TAN] * S[ATC_C_NM_TO_FT] + S[ATC_C_RWY_H];
    if (!(h <= h_max)) return false;
    return inside_corridor_angle(S, x, y, phi);
}

// atc_gym.py:17-19  (1 - tanh(4 d/dmax - 2)) / 2  ==  1 / (1 + exp(2 (4 d/dmax - 2)))   [exact identity]
__device__ __forceinline__ float sigmoid_distance(float d, float d_max) {
    const float z = 4.0f * (d / d_max) - 2.0f;
    return 1.0f / (1.0f + __expf(2.0f * z));
}

struct Shaping {
    float pos, ang, gs;
};
// atc_gym.py:199-260: _reward_approach_position, _reward_approach_angle, _reward_glideslope
__device__ __forceinline__ Shaping shaping_rewards(const float* S, float d_faf, float phi_rel_faf, float phi_plane, float h,
                                                   float on_gp) {
    const float to_rwy = S[ATC_C_PHI_TO_RWY];
    Shaping r;
    const float rel_faf = relative_angle(to_rwy, phi_rel_faf);
    const float u = fabsf(rel_faf) / 180.0f;
    r.pos = sigmoid_distance(d_faf, S[ATC_C_WORLD_DIAG]) * (u * sqrtf(u)) * 0.8f;  // u ** 1.5
    const float plane_to_runway = relative_angle(to_rwy, phi_plane);
    const float side = (rel_faf > 0.0f) ? 1.0f : ((rel_faf < 0.0f) ? -1.0f : 0.0f);  // np.sign
    const float q = (side * plane_to_runway - 22.5f) / 202.0f;
    float m = -(q * q) + 1.0f;  // (-(q ** 2.0) + 1.0) ** 32.0 by five squarings (even power: sign-safe)
    m = m * m;
    m = m * m;
    m = m * m;
    m = m * m;
    m = m * m;
    r.ang = m * r.pos * 1.2f;
    r.gs = sigmoid_distance(fabsf(h - on_gp), 36000.0f) * r.pos * 0.8f;
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

struct Aircraft {
    double x, y;      // positions accumulate in fp64 (fp32 accumulation drifts 0.5 ulp/step on straight legs)
    float h, phi, v;
};

// atc_gym.py:346-348 + model.py:13-52: aircraft k of env e enters at an entry point.
__device__ __forceinline__ Aircraft spawn(const float* S, const atc_params_t& p, int e, int k, int episode) {
    const int n_entry = (int)S[ATC_H_N_ENTRY];
    const float* tab = S + (int)S[ATC_H_OFF_ENTRY];
    int ei, li;
    if (p.mode & ATC_M_RANDOM_ENTRY) {
        const uint64_t u = draw(p.seed, (uint32_t)e, (uint32_t)episode, (uint32_t)k);
        ei = (int)((uint32_t)(u & 0xffffffffu) % (uint32_t)n_entry);
        li = (int)((uint32_t)(u >> 32) % (uint32_t)(int)tab[ei * ATC_E_WORDS + ATC_E_NLEV]);
    } else {
        ei = k % n_entry;
        li = (k / n_entry) % (int)tab[ei * ATC_E_WORDS + ATC_E_NLEV];
    }
    const float* rec = tab + ei * ATC_E_WORDS;
    Aircraft a;
    a.x = (double)rec[ATC_E_X];
    a.y = (double)rec[ATC_E_Y];
    a.phi = rec[ATC_E_PHI];
    a.h = rec[ATC_E_LEV0 + li] * 100.0f;
    a.v = S[ATC_C_V_INIT];
    return a;
}

struct Obs {
    float o[ATC_OBS_DIM];
    float d_faf, phi_rel_faf, on_gp;
};
// atc_gym.py:262-297 _get_state
__device__ __forceinline__ Obs get_state(const float* S, float x, float y, float h, float phi, float v, float mva) {
    Obs r;
    const float to_faf_x = S[ATC_C_FAF_X] - x;
    const float to_faf_y = S[ATC_C_FAF_Y] - y;
    r.d_faf = sqrtf(to_faf_x * to_faf_x + to_faf_y * to_faf_y);           // np.hypot
    r.phi_rel_faf = atan2f(to_faf_y, to_faf_x) * kRadToDeg;               // np.degrees(np.arctan2)
    r.on_gp = 318.4f * r.d_faf + S[ATC_C_FAF_MVA] - 200.0f;
    r.o[0] = x;
    r.o[1] = y;
    r.o[2] = h;
    r.o[3] = phi;
    r.o[4] = v;
    r.o[5] = h - mva;
    r.o[6] = r.on_gp;
    r.o[7] = r.d_faf;
    r.o[8] = r.phi_rel_faf;
    r.o[9] = relative_angle(S[ATC_C_PHI_TO_RWY], phi);
    return r;
}

}  // namespace atc
