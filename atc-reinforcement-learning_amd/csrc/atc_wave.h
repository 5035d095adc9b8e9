// atc_wave.h — cross-lane primitives of the step kernels (gfx950, 64-lane wavefronts): reductions over groups of W consecutive lanes
// through DPP operand modifiers, and the separation scans of envs of up to 16 aircraft (one DPP row per env, or XOR partners inside
// a quad / half row).  No LDS, no waits, no branches except the rare wave-uniform "some pair is close" block.
// Uses ATC_RARE (atc_step.hip) — included from there.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---------------------------------------------------------------------------------------------------------------
// wavefront-group helpers (groups of W consecutive lanes, W a power of two <= 64)
// ---------------------------------------------------------------------------------------------------------------
// Butterfly exchange partner for reductions over groups of W lanes.  Inside a DPP row (W <= 16) the stages are
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: after each stage both halves of the growing
// block hold the same value, so the mirrored pairings are as good as xor pairings — and they are VALU operand modifiers
// (no ds_bpermute, no index registers).  Wider groups fall back to __shfl_xor for the cross-row stages.
template <int STAGE>
__device__ __forceinline__ int dpp_stage(int v) {
    constexpr int ctrl = STAGE == 1 ? 0xB1 : STAGE == 2 ? 0x4E : STAGE == 4 ? 0x141 : 0x140;
    return __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, false);
}
template <int O>
__device__ __forceinline__ float xchg(float v) {
    if (O <= 8) return __int_as_float(dpp_stage<O>(__float_as_int(v)));
    return __shfl_xor(v, O, 64);
}
template <int O>
__device__ __forceinline__ int xchg(int v) {
    if (O <= 8) return dpp_stage<O>(v);
    return __shfl_xor(v, O, 64);
}
template <int W>
__device__ __forceinline__ float group_sum(float v) {
    if (W > 1) v += xchg<1>(v);
    if (W > 2) v += xchg<2>(v);
    if (W > 4) v += xchg<4>(v);
    if (W > 8) v += xchg<8>(v);
    if (W > 16) v += xchg<16>(v);
    if (W > 32) v += xchg<32>(v);
    return v;
}
template <int W>
__device__ __forceinline__ int group_sum_i(int v) {
    if (W > 1) v += xchg<1>(v);
    if (W > 2) v += xchg<2>(v);
    if (W > 4) v += xchg<4>(v);
    if (W > 8) v += xchg<8>(v);
    if (W > 16) v += xchg<16>(v);
    if (W > 32) v += xchg<32>(v);
    return v;
}
template <int W>
__device__ __forceinline__ float group_min(float v) {
    if (W > 1) v = fminf(v, xchg<1>(v));
    if (W > 2) v = fminf(v, xchg<2>(v));
    if (W > 4) v = fminf(v, xchg<4>(v));
    if (W > 8) v = fminf(v, xchg<8>(v));
    if (W > 16) v = fminf(v, xchg<16>(v));
    if (W > 32) v = fminf(v, xchg<32>(v));
    return v;
}
// (__builtin_amdgcn_ballot_w64 takes the predicate as the lane mask it already is; HIP's __ballot(int) first materialises
// it per lane and compares again: two vector instructions per use)
template <int W>
__device__ __forceinline__ uint64_t group_ballot(bool pred, int lane) {
    const uint64_t b = __builtin_amdgcn_ballot_w64(pred);
    if (W == 64) return b;
    const int base = lane & ~(W - 1);
    return (b >> base) & ((1ull << (W & 63)) - 1ull);   // (W < 64 here; the mask keeps the W = 64 instantiation warning-free)
}

// Separation scan for N = 16: one env = one DPP row (16 lanes).  Partner state arrives by row rotation (v_*_dpp
// row_ror:D, D = 1..15 visits every other lane of the row exactly once) — no LDS, no waits, no branches.
template <int D>
__device__ __forceinline__ float row_ror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + D, 0xf, 0xf, false));
}
// margin = max(d^2 - sep^2, |dh| - sep_ft) is negative exactly when both separation minima are violated (the sign of an
// IEEE difference is exact), so "any partner in conflict" is min over partners of margin < 0 — four VALU operations per
// partner, no compares, no mask arithmetic.
// Each unordered pair is evaluated ONCE: rotation D (1..8) makes a lane evaluate the partner D lanes away, and the inverse
// rotation 16-D hands that pair's margin (and d^2) back to the partner, for which it is the same pair (d^2 and |dh| are
// symmetric, so the values are bit-identical to what the partner would have computed).  D = 8 is its own inverse.
template <int D, bool WANT_MIN>
struct PairScan16 {
    static __device__ __forceinline__ void run(float xs, float y, float h, float sep2, float sep_ft, float& min_d2,
                                               float& margin) {
        const float dx = xs - row_ror<D>(xs), dy = y - row_ror<D>(y), dh = h - row_ror<D>(h);
        const float d2 = fmaf(dx, dx, dy * dy);
        const float m = fmaxf(d2 - sep2, fabsf(dh) - sep_ft);
        margin = fminf(margin, m);
        if (D < 8) margin = fminf(margin, row_ror<16 - D>(m));
        if (WANT_MIN) {
            min_d2 = fminf(min_d2, d2);
            if (D < 8) min_d2 = fminf(min_d2, row_ror<16 - D>(d2));
        }
        PairScan16<D + 1, WANT_MIN>::run(xs, y, h, sep2, sep_ft, min_d2, margin);
    }
};
template <bool WANT_MIN>
struct PairScan16<9, WANT_MIN> {
    static __device__ __forceinline__ void run(float, float, float, float, float, float&, float&) {}
};
// The same scan for launches that do not report the minimum separation (the fast variant): the horizontal question first — two
// rotated subtracts, a multiply, an fma and one compare into a lane mask — and the altitude is fetched and the result handed back to
// the partner only behind a wave-uniform test of that mask.  Conflict = (d^2 < sep^2) & (|dh| < sep_ft), the oracle's expression.
// (Round 5 tried the three-dimensional question per rotation plus a scan horizon for this width: fewer instructions, no time —
// profiles/experiments/README.md: scan16_form1.  The horizon ships for the LDS-staged widths only.)
struct ScanLimits {
    float sep2, sep_ft;       // the separation minima (squared horizontal, vertical)
    float sep2_h, sep_ft_h;   // the same with the horizon's closing distance added (== the minima where no horizon is used)
};
template <int D>
__device__ __forceinline__ int row_ror_i(int v) {
    // (every lane of a row rotation has a source: `old` is never used — passing v itself spares the zero the compiler would
    // otherwise materialise for it)
    return __builtin_amdgcn_update_dpp(v, v, 0x120 + D, 0xf, 0xf, false);
}
template <int D>
struct NearScan16H {
    static __device__ __forceinline__ void run(float xs, float y, float h, float sep2, float sep_ft, int& conf) {
        const float dx = xs - row_ror<D>(xs), dy = y - row_ror<D>(y);
        const float d2 = fmaf(dx, dx, dy * dy);
        const bool near = d2 < sep2;
        if (ATC_RARE(__builtin_amdgcn_ballot_w64(near) != 0ull)) {
            const float dh = h - row_ror<D>(h);
            const int c = (near && fabsf(dh) < sep_ft) ? 1 : 0;
            conf |= c;
            if (D < 8) conf |= row_ror_i<16 - D>(c);   // the partner's copy of the same pair (D = 8 is its own inverse)
        }
        NearScan16H<D + 1>::run(xs, y, h, sep2, sep_ft, conf);
    }
};
template <>
struct NearScan16H<9> {
    static __device__ __forceinline__ void run(float, float, float, float, float, int&) {}
};

// Separation scan for N <= 8 (W = 2, 4, 8): the partners of lane k are the lanes k ^ m, m = 1..W-1, of its aligned group,
// all reachable with DPP operand modifiers — quad_perm for m = 1, 2, 3, row_half_mirror for m = 7 (= 7 - k within 8 lanes)
// and quad_perm applied to the half-mirrored copy for m = 4, 5, 6 (7 ^ 3, 7 ^ 2, 7 ^ 1).  No LDS, no waits.  Both lanes of
// a pair evaluate it (bit-identical: d^2 and |dh| are symmetric), so no hand-back is needed.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <bool WANT_MIN>
__device__ __forceinline__ void pair_eval(float xs, float y, float h, float px, float py, float ph, float sep2, float sep_ft,
                                          float& min_d2, float& margin) {
    const float dx = xs - px, dy = y - py;
    const float d2 = fmaf(dx, dx, dy * dy);
    margin = fminf(margin, fmaxf(d2 - sep2, fabsf(h - ph) - sep_ft));
    if (WANT_MIN) min_d2 = fminf(min_d2, d2);
}
template <int W, bool WANT_MIN>
__device__ __forceinline__ void pair_scan_xor(float xs, float y, float h, float sep2, float sep_ft, float& min_d2,
                                              float& margin) {
    constexpr int X1 = 0xB1, X2 = 0x4E, X3 = 0x1B, HALF_MIRROR = 0x141;  // quad_perm [1,0,3,2] [2,3,0,1] [3,2,1,0]
    pair_eval<WANT_MIN>(xs, y, h, dpp_f<X1>(xs), dpp_f<X1>(y), dpp_f<X1>(h), sep2, sep_ft, min_d2, margin);
    if (W >= 4) {
        pair_eval<WANT_MIN>(xs, y, h, dpp_f<X2>(xs), dpp_f<X2>(y), dpp_f<X2>(h), sep2, sep_ft, min_d2, margin);
        pair_eval<WANT_MIN>(xs, y, h, dpp_f<X3>(xs), dpp_f<X3>(y), dpp_f<X3>(h), sep2, sep_ft, min_d2, margin);
    }
    if (W >= 8) {
        const float mx = dpp_f<HALF_MIRROR>(xs), my = dpp_f<HALF_MIRROR>(y), mh = dpp_f<HALF_MIRROR>(h);
        pair_eval<WANT_MIN>(xs, y, h, mx, my, mh, sep2, sep_ft, min_d2, margin);                                      // k ^ 7
        pair_eval<WANT_MIN>(xs, y, h, dpp_f<X3>(mx), dpp_f<X3>(my), dpp_f<X3>(mh), sep2, sep_ft, min_d2, margin);   // k ^ 4
        pair_eval<WANT_MIN>(xs, y, h, dpp_f<X2>(mx), dpp_f<X2>(my), dpp_f<X2>(mh), sep2, sep_ft, min_d2, margin);   // k ^ 5
        pair_eval<WANT_MIN>(xs, y, h, dpp_f<X1>(mx), dpp_f<X1>(my), dpp_f<X1>(mh), sep2, sep_ft, min_d2, margin);   // k ^ 6
    }
}

