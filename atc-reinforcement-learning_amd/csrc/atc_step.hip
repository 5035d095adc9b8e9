// atc_step.hip — libatcstep.so: hand-written gfx950 (CDNA4, MI355X) kernels + C-ABI of the batched AtcGym.step() path.
//
// Work decomposition: lane = one aircraft slot, W = next_pow2(N) consecutive lanes = one env, 64 / W envs per wavefront; workgroup =
// 256 consecutive slots, one per tile, no block barrier, no atomics in the fast variant, no MFMA (no contraction on this path).
//   State in HBM (include/atc_step.h, ABI 20): ac = (x, y on the 32-bit position grid, heading counts, speed counts) 16 B, alt = the
//   altitude as the reference's float64 8 B, last_act 16 B (touched only by steps that may change it), a 16-byte env record shared
//   by the W lanes of an env; every array moves with ONE access per lane on consecutive addresses, 32-bit offsets from uniform bases.
//   The sector blob stays in global memory: constants through scalar loads, the MVA lookup one 8-byte gather per aircraft (+ edge
//   records in cells a border cuts).  No staging prologue: the first thing a wavefront does is issue its state loads.
//   Separation scan, every unordered pair once: N = 16 one DPP row per env (row rotation fused into the subtract, the inverse
//   rotation hands the result back); N <= 8 XOR partners through quad_perm / row_half_mirror; N > 16 LDS planes + packed fp32 +
//   ballot masks rotated on the scalar unit.  Per-env sums by DPP butterflies, done / won masks by ballot.
//   The single-step kernel has NO loop around the step body (its own instantiation, ONE); multi-step launches keep the state in
//   registers across a run-time step loop under an 80-VGPR bound; LAT is their form for at most two wavefronts per SIMD; k_serve
//   is the resident one-env step server.  How each choice was measured: DESIGN_HISTORY.md, profiles/experiments/.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared   (see build.py)
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <sched.h>
#include <emmintrin.h>   // host side of atc_step_packet: 16-byte loads of the mapped result packet
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "atc_device.h"

using namespace atc;

struct atc_scenario {
    float* d_blob;   // device copy of the whole blob (constants + polygons + entries [+ grid])
    int n_words;
    int off_grid;    // 0 = no grid
    int device;
    int n_cu;
    float consts[ATC_C_END];   // host copy of the blob's header + constants block (derive() evaluates uniform terms from it)
    float ghdr[ATC_G_HDR];     // host copy of the lookup grid's header (zeros without a grid)
    uint64_t uid;              // unique per created handle (never reused: keys the per-thread cache of derive())
    void* d_lds;               // LDS-resident lookup table (atc_scenario_attach_lds_table): device copy, or null
    LdsTab lt;                 // ... and the kernel argument that describes it (src == null: none)
    size_t lds_tab_bytes;
    int max_lds;               // the device's LDS bytes per workgroup
};

static thread_local char g_err[512] = "";

#define ATC_OK 0
#define ATC_ERR_ARG (-1)
#define ATC_ERR_HIP (-2)

static int fail_arg(const char* msg) {
    snprintf(g_err, sizeof g_err, "bad argument: %s", msg);
    return ATC_ERR_ARG;
}
static int fail_hip(hipError_t e, const char* what) {
    snprintf(g_err, sizeof g_err, "HIP error in %s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();  // reported here: do not leave it as the runtime's sticky "last error" for the caller's framework
    return ATC_ERR_HIP;
}
#define HIP_TRY(expr)                                    \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return fail_hip(_e, #expr); \
    } while (0)

// Rejected / default-off variants of rounds 1-5 (scan forms, ablation masks, store orders, ...) were moved out of this file in
// round 6: profiles/experiments/ holds the patch that restores them and one line per measurement.
// Wave-uniform conditions that almost never hold (or almost always): the hint moves the rare block out of the step's straight
// line, so that the common path FALLS THROUGH its branches instead of jumping over code (a taken branch refills the
// instruction buffer: ~20 cycles to a wavefront alone on its SIMD, tools/ubench/valu_rates.hip).
#define ATC_RARE(x) __builtin_expect(!!(x), 0)
#define ATC_USUAL(x) __builtin_expect(!!(x), 1)
constexpr int kBlock = 256;
#ifndef ATC_TRACE
#define ATC_TRACE 0  // developer-only: per-wavefront s_memtime stamps at phase boundaries (pointer smuggled in params)
#endif
#if ATC_TRACE
// one row of 8 stamps per wavefront AND step (row = wavefront * steps + step)
#ifndef ATC_TRACE_MODE
#define ATC_TRACE_MODE 0   // 0: the step's phases (tools/trace_rollout.py); 1: stamps 1..5 dissect the step's first phase instead
#endif
#define ATC_STAMP_AT(row, n) do { if ((threadIdx.x & 63) == 0 && (row)) (row)[(n)] = __builtin_amdgcn_s_memtime(); } while (0)
#define ATC_STAMP(n) do { if (ATC_TRACE_MODE == 0) ATC_STAMP_AT(trow, n); } while (0)
#define ATC_STAMP_B(n) do { if (ATC_TRACE_MODE == 0) ATC_STAMP_AT(so.trace, n); } while (0)
#define ATC_STAMP_TOP(row, n) do { if (ATC_TRACE_MODE == 1) ATC_STAMP_AT(row, n); } while (0)
#define ATC_STAMP_END(row, n) do { if (ATC_TRACE_MODE == 2) ATC_STAMP_AT(row, n); } while (0)   // mode 2: the step's LAST phase dissected
#define ATC_TRACE_PARAM , unsigned long long* trace_row
#define ATC_TRACE_PASS(x) , (x)
#else
#define ATC_STAMP(n) do {} while (0)
#define ATC_STAMP_B(n) do {} while (0)
#define ATC_STAMP_TOP(row, n) do {} while (0)
#define ATC_STAMP_END(row, n) do {} while (0)
#define ATC_TRACE_PARAM
#define ATC_TRACE_PASS(x)
#endif
// Multi-step launches of the DPP widths re-read the STATE pointers from the kernarg segment after the step loop (see StepArgs)
// instead of carrying them across it; the LDS-scan widths name them (measured per width: profiles/r02_experiments.txt).
constexpr bool loop_rereads_state(int W) { return W < 32; }
#define ATC_GRID_CAP 8        // workgroups per CU before the reset / observe / query kernels grid-stride
#define ATC_MIN_WAVES 4       // waves per SIMD the single-step kernel is register-budgeted for (<= 128 VGPRs; it needs 55-72)
#define ATC_MIN_WAVES_LOOP 6  // multi-step launches: <= 80 VGPRs.  Without the bound the allocator keeps literal constants and
                              // other loop invariants in registers across the step loop (86-99 VGPRs, 4-5 waves per SIMD)

#include "atc_wave.h"   // group reductions (DPP butterflies) and the DPP separation scans

// Per-lane addressing = uniform 64-bit base + 32-bit BYTE offset (the host guarantees B*N*40 < 4 GiB): the compiler can
// then use the scalar-base addressing form and does not keep a 64-bit address pair per array alive in VGPRs.
template <typename T>
__device__ __forceinline__ T* at(void* base, uint32_t byte_off) {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ const T* at(const void* base, uint32_t byte_off) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
// Byte offsets of 12- and 40-byte records as shift-adds: the compiler multiplies (v_mul_lo_u32 is a quarter-rate instruction,
// and lane indices do not fit the 24-bit multiplier); asm keeps the two full-rate instructions from being folded back.
__device__ __forceinline__ uint32_t times12(uint32_t i) {
    uint32_t t;
    asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(t) : "v"(i));   // 3 i
    return t << 2;
}
__device__ __forceinline__ uint32_t times40(uint32_t i) {
    uint32_t t;
    asm("v_lshl_add_u32 %0, %1, 2, %1" : "=v"(t) : "v"(i));   // 5 i
    return t << 3;
}
// write-once output streams (observation, flag words) are stored non-temporal: 30.7 vs 31.9 us at 65 536 x 16 (r01)
template <typename T>
__device__ __forceinline__ void stream_store(T* p, T v) {
    __builtin_nontemporal_store(v, p);
}
// One 16-byte store written THROUGH to system memory (sc0 sc1): a chunk of the result packet a host polls in mapped memory.  A plain
// store to host memory may stay in the device's L2 until the kernel ends — which a resident kernel (k_serve) never does.
__device__ __forceinline__ void store16_system(uint4* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_obs(float* __restrict__ dst, const float* o) {
    // 10 floats = 40 B per aircraft: 8-byte aligned -> five 8-byte stores
    float2* d = reinterpret_cast<float2*>(dst);
#pragma unroll
    for (int c = 0; c < 5; ++c) d[c] = make_float2(o[2 * c], o[2 * c + 1]);
}

// ---------------------------------------------------------------------------------------------------------------
// step / rollout kernel
//   FULL = false : the fast variant — obs, reward, done, flags only (what step() returns); the optional outputs
//                  (raw_obs, ac_reward, min_sep, term_obs) and their pointers are compiled out.
//   One workgroup handles 256 consecutive slots (no grid-stride loop: with a loop the compiler hoists the uniform loads
//   out of it and spills the scalar registers that then stay live across the whole body).
//   The body is straight-line for every lane: lanes whose aircraft is not under control compute on their frozen state and
//   the results are discarded by selects; only the rare paths (dirty grid cells, corridor interior, auto-reset) branch.
// ---------------------------------------------------------------------------------------------------------------
struct LaneIds {        // who this lane is (one aircraft slot of one env)
    int tid, lane, e, k;
    uint32_t i;         // aircraft index env * N + k (clamped into range for lanes without an aircraft)
    uint32_t slot0;     // first slot of the workgroup's tile
    bool env_valid, lane_valid, wave_full;
};
struct LaneState {      // persistent per-aircraft state held in registers
    Aircraft a;
    uint32_t la_v;      // last accepted targets (atc_gym.py:86,311) in the state's formats: speed counts,
    double la_h;        // altitude [ft] (float64),
    int la_p;           // heading counts
    bool la_changed;
};
struct Targets {        // decoded action of the current step / block (atc_gym.py:318-335) in the state's formats
    uint32_t v;
    float ah;           // the altitude's ACTION: its float64 target a * m + c is evaluated where it is used (one conversion and
    int p;              // one fma per step instead of a register pair carried across a held block)
};
struct EnvState {       // per-step env record (replicated in the W lanes of the env)
    int t, n_actions;
    float total_reward;
    uint64_t amask;     // bits 32..63 live in the per-episode record and exist only for envs of more than 32 aircraft
};
struct Float3 {
    float a, b, c;
};
struct Mid {            // what the first half of a step hands to the second
    bool active;
    float r;
    uint32_t fl;
    int acts;
    float x32, y32;
    MvaCell cell;       // MVA lookup cell, gather issued in the first half, resolved after the separation scan
    LdsCode lc;         // ... or its codes in the LDS-resident table (LDSG instantiation)
    bool repeated;      // uniform: no action bookkeeping in this step (acts == 0 in every lane)
    bool plain;         // uniform: every lane flies an aircraft under control towards valid targets (fl == 0, no refusal)
};
// Uniform products of the step parameters, evaluated ONCE on the host in fp32 (the same IEEE operations the kernel
// would do) and passed as kernel arguments: gfx950 has no scalar float ALU, so computed in the kernel they would occupy
// vector registers — and a multi-step launch would keep them there across its whole step loop.
// Grouped by the stage that consumes them: a multi-step launch re-reads each group from the kernarg segment right where its
// stage starts (QGET below) instead of keeping ~70 uniform values alive across the step loop.
struct alignas(16) QRates {   // first half of the step (20 words)
    // _denormalized_action (atc_gym.py:318-335) for the fixed-point components: counts = trunc(a * m + c) in float64 (m, c the
    // reference's factor / offset in counts: integers), see decode_targets
    double dec_mv, dec_cv, dec_mp, dec_cp;
    double dec_mh, dec_ch;  // altitude target (float64): a * m + c as one fma = the reference's operation order (see derive())
    double dh_hi, dh_lo;    // kHDotMax * dt, kHDotMin * dt in float64 (model.py:45-46,97-100)
    int rate_v, rate_p;     // rint(kAMax dt 2^23), rint(kPhiDotMax dt 2^23): symmetric limits (model.py:47-50,75-78,113-120)
    float r_base;           // -0.05 * dt                       (atc_gym.py:137)
    int pos_neg_k;          // position grid: nm = origin + fix * 2^-k (blob: ATC_C_POS_*)
};
struct alignas(16) QGrid {    // position conversion + MVA cell lookup
    double pos_x0, pos_y0;
    GridHdr gh;           // lookup grid header (7 words)
    int pad;
};
struct alignas(16) QScan {    // second half: separation scan, override chain
    float sep2;           // sep_nm ^ 2
    float sep_ft, conflict_reward;
    int timestep_limit;
    float4 tri_bbox;      // bounds of the corridor's horizontal triangle (ATC_C_TRI_BBOX)
    int n_noise;          // number of noise-abatement areas (blob header word; read from the blob inside the step it was a
    int off_spawn;        // dependent scalar load with nothing to hide its latency behind, in every step of every wavefront);
                          // word offset of the blob's spawn records (ATC_H_OFF_SPAWN)
    float sep2_h, sep_ft_h;   // the minima with the closing distance of the launch's scan horizon added (scan_horizon_limits)
};
struct alignas(16) QNorm {
    float a[ATC_OBS_DIM], b[ATC_OBS_DIM];   // ATC_C_NORM_A / ATC_C_NORM_B
};
struct alignas(64) StepDerived {
    QRates r;
    QKin k;               // float64 heading kinematics + distance scale (32 words)
    QGrid g;
    QScan s;
    ObsConst oc;          // observation / shaping constants
    int pad[2];
    QNorm n;
};
// LAT — the latency-bound instantiation of the multi-step kernels, for launches of at most two wavefronts per SIMD (65 536 x 1,
// 8 192 x 16: every BASELINE configuration but the headline and 4 096 x 64).  There a step IS one wavefront's serial instruction
// chain and the vector register file is all but empty, so the uniform FLOATING-POINT terms of StepDerived (~80 words) are copied
// into vector registers once per launch — every lane holds the same value — and the step loop reads them as ordinary operands:
// no kernarg re-read (a scalar load and its wait: ~45 cycles for a wavefront alone, 17 per step at W = 1), no scalar register
// spilled to lanes, no constant-bus limit.  Integer terms that steer scalar control flow stay scalar.
__device__ __forceinline__ float vg(float x) {
    float r;
    asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(x));
    return r;
}
__device__ __forceinline__ double vg(double x) {
    int lo, hi;
    asm("v_mov_b32 %0, %1" : "=v"(lo) : "s"(__double2loint(x)));
    asm("v_mov_b32 %0, %1" : "=v"(hi) : "s"(__double2hiint(x)));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int vg(int x) {
    int r;
    asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(x));
    return r;
}
__device__ __forceinline__ StepDerived to_vector_registers(const StepDerived& q) {
    StepDerived v = q;
    // (integers that are only ever VALU operands — limits, grid dimensions, the FAF's grid coordinates — go along: each frees a scalar
    // register the loop would otherwise spill to a vector-register lane and read back with v_readlane)
    v.r.rate_v = vg(q.r.rate_v); v.r.rate_p = vg(q.r.rate_p); v.r.pos_neg_k = vg(q.r.pos_neg_k);
    v.g.gh.nx = vg(q.g.gh.nx); v.g.gh.nx_last = vg(q.g.gh.nx_last); v.g.gh.ny_last = vg(q.g.gh.ny_last);
    v.oc.faf_x = vg(q.oc.faf_x); v.oc.faf_y = vg(q.oc.faf_y);
    v.s.timestep_limit = vg(q.s.timestep_limit);
    v.r.dec_mv = vg(q.r.dec_mv); v.r.dec_cv = vg(q.r.dec_cv); v.r.dec_mp = vg(q.r.dec_mp); v.r.dec_cp = vg(q.r.dec_cp);
    v.r.dec_mh = vg(q.r.dec_mh); v.r.dec_ch = vg(q.r.dec_ch);
    v.r.dh_hi = vg(q.r.dh_hi); v.r.dh_lo = vg(q.r.dh_lo); v.r.r_base = vg(q.r.r_base);
    v.k.inv180 = vg(q.k.inv180); v.k.neg_half_turn = vg(q.k.neg_half_turn);
    v.k.s0 = vg(q.k.s0); v.k.s1 = vg(q.k.s1); v.k.s2 = vg(q.k.s2); v.k.s3 = vg(q.k.s3); v.k.s4 = vg(q.k.s4); v.k.s5 = vg(q.k.s5);
    v.k.c1 = vg(q.k.c1); v.k.c2 = vg(q.k.c2); v.k.c3 = vg(q.k.c3); v.k.c4 = vg(q.k.c4); v.k.c5 = vg(q.k.c5);
    v.k.dist_neg = vg(q.k.dist_neg);
    v.g.pos_x0 = vg(q.g.pos_x0); v.g.pos_y0 = vg(q.g.pos_y0);
    v.g.gh.x0 = vg(q.g.gh.x0); v.g.gh.y0 = vg(q.g.gh.y0); v.g.gh.inv = vg(q.g.gh.inv);
    v.s.sep2 = vg(q.s.sep2); v.s.sep_ft = vg(q.s.sep_ft); v.s.conflict_reward = vg(q.s.conflict_reward);
    v.s.sep2_h = vg(q.s.sep2_h); v.s.sep_ft_h = vg(q.s.sep_ft_h);
    v.oc.pos_inv = vg(q.oc.pos_inv); v.oc.to_rwy = vg(q.oc.to_rwy); v.oc.on_gp_c = vg(q.oc.on_gp_c); v.oc.sig_a = vg(q.oc.sig_a);
#pragma unroll
    for (int c = 0; c < ATC_OBS_DIM; ++c) { v.n.a[c] = vg(q.n.a[c]); v.n.b[c] = vg(q.n.b[c]); }
    return v;
}

// atc_step_packet: the ONE env's action by value, a kernel argument of its own (no read over the host link on the device side)
struct InlineAction {
    float v, h, p;
    int set;
};
static thread_local const float* t_inline_action = nullptr;   // set by atc_step_packet around its launch
static InlineAction inline_action() {
    InlineAction a = {0.0f, 0.0f, 0.0f, 0};
    if (t_inline_action) a = InlineAction{t_inline_action[0], t_inline_action[1], t_inline_action[2], 1};
    return a;
}
static StepDerived derive_uncached(const atc_params_t& p, const atc_scenario* s, int horizon);
// The uniform terms depend on the sector and on a few parameters only; a caller steps the same env thousands of times with the
// same ones, so each thread remembers its last evaluation (the single-env path launches one tiny kernel per step: the
// evaluation would be a measurable part of its host time).  Thread-local: the library stays free of shared mutable state.
// `horizon`: the steps a multi-step launch may leave the separation scan out after one that found every pair far apart (0: none)
static const StepDerived& derive(const atc_params_t& p, const atc_scenario* s, int horizon) {
    struct Key {
        uint64_t uid;
        double dt;
        float sep_nm, sep_ft, conflict_reward;
        int32_t timestep_limit;
        uint32_t mode_bits;   // the mode flags the derived values depend on
        int32_t horizon;
    };
    static thread_local Key last = {0ull, 0.0, 0.0f, 0.0f, 0.0f, 0, 0u, 0};
    static thread_local StepDerived q;
    const Key k = {s->uid, p.dt, p.sep_nm, p.sep_ft, p.conflict_reward, p.timestep_limit, p.mode & (uint32_t)(ATC_M_DISCRETE | ATC_M_NORMALIZE), horizon};
    if (memcmp(&k, &last, sizeof k) != 0) {
        q = derive_uncached(p, s, horizon);
        last = k;
    }
    return q;
}
// Separation scan horizon (include/atc_step.h): thresholds S2 >= sep_nm^2 and SF >= sep_ft such that a pair of aircraft with
//   d^2 >= S2  or  |dh| >= SF     (d^2, dh as the scan evaluates them, fp32)
// now cannot satisfy (d^2 < sep_nm^2 and |dh| < sep_ft) in any of the next n steps.  Per step an aircraft moves by at most
// (v / 3600) dt (1 + 1e-9) nm + one position count per axis (advance(): |sin|, |cos| <= 1 + 1e-9, floor of displacement + dither) with
// v <= 300 kt as long as it is at most 300 kt now (valid targets lie in [100, 300], the rate-limited move never overshoots: the scan
// checks the speeds it starts from), and its altitude by at most 15 dt up / 41 dt down + half an ulp (the scan checks |h| < 2^17 ft:
// ulp <= 2^-6 ft).  So d shrinks by at most 2 (300 / 3600) dt and |dh| by at most 56 dt per step; the slack terms — 1e-5 relative
// + 1e-3 nm, 1e-5 relative + 1 ft — are hundreds of times the fp32 evaluation error of d^2 (a few ulp + 2^-17 nm per coordinate).
// Aircraft that are handed over stay where they are until their env is reset, and a reset ends the horizon (step_part_b).
static void scan_horizon_limits(const atc_params_t& p, int n, float* sep2_h, float* sep_ft_h) {
    const float sep2 = p.sep_nm * p.sep_nm;
    *sep2_h = sep2;
    *sep_ft_h = p.sep_ft;
    if (n <= 0) return;
    const double dt = p.dt;
    const double S = (sqrt((double)sep2) + n * 2.0 * ((double)kVMax / 3600.0) * dt) * (1.0 + 1e-5) + 1e-3;
    const double F = ((double)p.sep_ft + n * ((double)kHDotMax - (double)kHDotMin) * dt) * (1.0 + 1e-5) + 1.0;
    float s2 = (float)(S * S), f = (float)F;
    if ((double)s2 < S * S) s2 = nextafterf(s2, INFINITY);
    if ((double)f < F) f = nextafterf(f, INFINITY);
    // (NaN parameters: every compare of the scan is false either way; thresholds that did not come out above the minima — overflow,
    // NaN — become +inf: "some pair is near" in every scan, i.e. no step is skipped)
    *sep2_h = (s2 >= sep2) ? s2 : INFINITY;
    *sep_ft_h = (f >= p.sep_ft) ? f : INFINITY;
}
static StepDerived derive_uncached(const atc_params_t& p, const atc_scenario* s, int horizon) {
    const float* K = s->consts;
    StepDerived q;
    memset(&q, 0, sizeof q);
    int k_pos;
    {
        int e = 0;
        (void)frexpf(K[ATC_C_POS_SCALE], &e);   // 2^k = 0.5 * 2^(k + 1)
        k_pos = e - 1;
        q.r.pos_neg_k = -k_pos;
    }
    q.g.pos_x0 = (double)K[ATC_C_POS_X0];
    q.g.pos_y0 = (double)K[ATC_C_POS_Y0];
    const double dtd = p.dt, fixq = 8388608.0;   // 2^23: speed and heading counts per kt / deg
    auto rate_fix = [&](float rate) {   // rint(|rate| dt 2^23), saturating (include/atc_step.h)
        const double r = rint(fabs((double)rate) * dtd * fixq);
        return r >= 2147483647.0 ? INT32_MAX : (int32_t)r;
    };
    q.r.rate_v = rate_fix(kAMax);
    q.r.rate_p = rate_fix(kPhiDotMax);
    q.r.dh_hi = (double)kHDotMax * dtd;   // h_dot_max * timestep: an integer times a Python float (model.py:97-100)
    q.r.dh_lo = (double)kHDotMin * dtd;
    q.r.r_base = (float)(-0.05 * dtd);
    // atc_gym.py:64-78,318-335: offset (v_min, 0, 0); factor (10, 100, 1) discrete | (v_max - v_min, h_max, 360) continuous.
    //   discrete   : a * fac + off
    //   continuous : a * fac / 2 + fac / 2 + off
    // Speed and heading (fixed point): counts = a * m + c evaluated as ONE float64 fma — exact for every fp32 action in the
    // action space (24 x 31 bits) — and truncated; m = the factor, c = the offset relative to the format's origin, in counts.
    const bool discrete = (p.mode & ATC_M_DISCRETE) != 0;
    q.r.dec_mv = (discrete ? 10.0 : (double)(kVMax - kVMin) / 2.0) * fixq;
    q.r.dec_cv = (discrete ? (double)kVMin : (double)(kVMax - kVMin) / 2.0 + (double)kVMin) * fixq;
    q.r.dec_mp = (discrete ? 1.0 : 360.0 / 2.0) * fixq;
    q.r.dec_cp = ((discrete ? 0.0 : 360.0 / 2.0) - (double)ATC_PHI_FIX_OFFSET) * fixq;
    // Altitude (float64, ABI 20): the product of an fp32 action with fac (38 000: 16 bits) is exact in float64 and halving is
    // exact, so a * fac / 2 + fac / 2 + 0 is ONE rounding — the fma's (the discrete form keeps its `+ 0`: it turns a -0 product
    // into +0, like the reference's `+ offset`).
    q.r.dec_mh = discrete ? 100.0 : (double)kHMax / 2.0;
    q.r.dec_ch = discrete ? 0.0 : (double)kHMax / 2.0;
    // float64 heading kinematics (include/atc_step.h: ATC_KIN_*) and the step's distance scale, NEGATED (see atc::advance)
    q.k.inv180 = ATC_KIN_INV180;
    q.k.neg_half_turn = -ATC_KIN_HALF_TURN;
    q.k.s0 = ATC_KIN_S0; q.k.s1 = ATC_KIN_S1; q.k.s2 = ATC_KIN_S2; q.k.s3 = ATC_KIN_S3; q.k.s4 = ATC_KIN_S4; q.k.s5 = ATC_KIN_S5;
    q.k.c1 = ATC_KIN_C1; q.k.c2 = ATC_KIN_C2; q.k.c3 = ATC_KIN_C3; q.k.c4 = ATC_KIN_C4; q.k.c5 = ATC_KIN_C5;
    q.k.dist_neg = -ldexp(dtd / 3600.0, k_pos - ATC_V_FIX_SHIFT);
    q.g.gh = grid_header(s->off_grid ? s->ghdr : nullptr);
    q.s.sep2 = p.sep_nm * p.sep_nm;
    q.s.sep_ft = p.sep_ft;
    scan_horizon_limits(p, horizon, &q.s.sep2_h, &q.s.sep_ft_h);
    q.s.conflict_reward = p.conflict_reward;
    q.s.timestep_limit = p.timestep_limit;
    q.s.tri_bbox = make_float4(K[ATC_C_TRI_BBOX], K[ATC_C_TRI_BBOX + 1], K[ATC_C_TRI_BBOX + 2], K[ATC_C_TRI_BBOX + 3]);
    q.s.n_noise = (int)K[ATC_H_N_NOISE];
    q.s.off_spawn = (int)K[ATC_H_OFF_SPAWN];
    q.oc = obs_const(K);
    // atc_gym.py:187-189.  Without ATC_M_NORMALIZE the same fma runs with (1, -0): x * 1 + -0 == x for every x, signed
    // zeros included — the step has no branch on the flag and the constants can be requested ahead of the observation.
    const bool normalize = (p.mode & ATC_M_NORMALIZE) != 0;
    for (int c = 0; c < ATC_OBS_DIM; ++c) {
        q.n.a[c] = normalize ? K[ATC_C_NORM_A + c] : 1.0f;
        q.n.b[c] = normalize ? K[ATC_C_NORM_B + c] : -0.0f;
    }
    return q;
}

// The kernel's argument list as a struct: the kernarg segment is laid out by the same rules, so offsetof() names where an
// argument lives.  Multi-step launches RE-READ their by-value arguments (output pointers, parameters, derived constants, the state
// pointers needed again after the loop) from the kernarg segment inside the step instead of keeping ~60 SGPRs of them alive
// across the loop: the allocator spills those to VGPR lanes and every use becomes a v_readlane — VALU work in a launch that is
// bound by VALU issue — whereas a kernarg re-read is a scalar load.  The offset goes through an opaque zero, or the (invariant)
// loads would be hoisted out of the loop again.
struct StepArgs {
    const float* blob;
    int off_grid, B, N, T, hold;
    atc_state_t st;
    const float* actions;
    atc_out_t out;
    atc_params_t p;
    StepDerived q;
    InlineAction ia;
    LdsTab lt;
};
template <typename T>
__device__ __forceinline__ T kernarg_reread(size_t byte_off, int opaque_zero) {
#if __HIP_DEVICE_COMPILE__   // (the host pass only parses this body; the builtin exists for the device target)
    // A TYPED load through the constant address space: the alignment of T is then known to the compiler, which it needs
    // for scalar loads (a byte-wise copy from an address with an opaque term became per-lane vector loads).
    typedef __attribute__((address_space(4))) const char* karg_ptr;
    typedef __attribute__((address_space(4))) const T* typed_ptr;
    // The BASE POINTER is made opaque (an empty asm tied to the step's opaque zero), the member's offset stays an immediate of
    // the scalar load (adding `opaque_zero * alignof(T)` to the address cost five scalar instructions per re-read).
    karg_ptr base = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base) : "s"(opaque_zero));
    return *(typed_ptr)(base + byte_off);
#else
    T v;
    __builtin_memset(&v, 0, sizeof(T));
    return v;
#endif
}

// base of atc_state_t.phi_wide on the rare paths that need it: the named argument in single-step launches, a kernarg re-read in
// multi-step ones (like the per-episode record's base: nothing about it is carried across the step loop)
template <bool ONE>
__device__ __forceinline__ double* wide_base(double* named, int zk) {
    return ONE ? named : kernarg_reread<double*>(offsetof(StepArgs, st) + offsetof(atc_state_t, phi_wide), zk);
}
// WIDE headings (include/atc_step.h, ABI 19) exist only in wavefronts that are not `plain`; their wrapped counts and observation
// word 3 were left in word 2 of the side record by the first half of this step (step_part_a) — 4-byte loads behind a second
// wave-uniform test, no arithmetic and as few registers as possible where the observation's registers are all live.
// WORD 0: the wrapped counts (what kinematics and angles use in place of phi_fix), WORD 1: observation word 3 as a bit pattern.
template <bool ONE, int WORD>
__device__ __forceinline__ int wide_view(bool plain, int phi, double* named, int zk, uint32_t i, int dflt) {
    int r = dflt;
    if (ATC_RARE(!plain)) {
        if ((__builtin_amdgcn_ballot_w64(phi == INT32_MAX) | __builtin_amdgcn_ballot_w64(phi == INT32_MIN)) != 0ull) {
            if (is_wide(phi)) r = *at<int>(wide_base<ONE>(named, zk), i * 32u + (16u + 4u * WORD));   // (B N 40 < 4 GiB is guaranteed)
            // The load is WAITED FOR here, inside the rare block: left pending, the join below would carry "r may be in flight" into
            // the step's straight line, where the first use of r waits for every earlier vector load as well (one counter, in
            // order) — e.g. the lookup-cell records requested ahead of the observation arithmetic that is meant to cover them.
            asm volatile("" : "+v"(r));
        }
    }
    return r;
}
// Latency-bound instantiation: ONE look-up per step (both words, kept in the vector registers it has to spare) instead of one
// wave-uniform test per use site — a not-taken scalar branch is ~20 cycles of a lone wavefront's chain.
struct WideWords {
    int counts, obs_bits;
};
template <bool ONE>
__device__ __forceinline__ WideWords wide_words(bool plain, int phi, double* named, int zk, uint32_t i) {
    WideWords w = {phi, __float_as_int(phi_real(phi))};
    if (ATC_RARE(!plain)) {
        if ((__builtin_amdgcn_ballot_w64(phi == INT32_MAX) | __builtin_amdgcn_ballot_w64(phi == INT32_MIN)) != 0ull) {
            if (is_wide(phi)) {
                const int2 v = *at<int2>(wide_base<ONE>(named, zk), i * 32u + 16u);
                w.counts = v.x;
                w.obs_bits = v.y;
            }
            asm volatile("" : "+v"(w.counts), "+v"(w.obs_bits));   // (waited for here, inside the rare block: see wide_view)
        }
    }
    return w;
}
// the heading's counts as kinematics and angles see them (both are periodic in the heading)
template <bool ONE>
__device__ __forceinline__ int heading_counts(bool plain, int phi, double* named, int zk, uint32_t i) {
    return wide_view<ONE, 0>(plain, phi, named, zk, i, phi);
}

struct StepOut {        // per-step output bases (uniform pointers)
    float* obs;
    uint16_t* flags;
    float* reward;
    uint8_t* done;
    float *raw_obs, *ac_reward, *min_sep, *term_obs;
    uint32_t* packet;
#if ATC_TRACE
    unsigned long long* trace;
#endif
};

// ALLV ("all valid"): the launch's slots are all aircraft — N == W and B * W a multiple of the workgroup size (every BASELINE
// configuration) — so the validity flags are compile-time true: no clamped indices, no validity terms in the active masks, no
// exec-mask regions around the stores, the full-wavefront observation store unconditionally.  Fused 65 536 x 16: 11.9 vs 12.3 us
// per step, 8 192 x 16 2.55 vs 2.70 (r04, same box).  Chosen by the host per launch; other shapes take the general kernels.
template <int W, bool ALLV>
__device__ __forceinline__ LaneIds make_ids(uint32_t slot0, int B, int N) {
    LaneIds d;
    d.tid = threadIdx.x;
    d.lane = d.tid & 63;
    d.slot0 = slot0;
    const uint32_t slots = (uint32_t)B * (uint32_t)W;
    const uint32_t slot = slot0 + d.tid;
    d.k = (int)(slot % W);
    if (ALLV) {
        d.env_valid = d.lane_valid = d.wave_full = true;
        d.e = (int)(slot / W);
        d.i = slot;   // N == W: env * N + k
        return d;
    }
    d.env_valid = slot < slots;
    d.e = d.env_valid ? (int)(slot / W) : B - 1;  // clamped: loads stay in bounds, results are never stored
    d.lane_valid = d.env_valid && d.k < N;
    d.i = d.lane_valid ? (uint32_t)d.e * (uint32_t)N + (uint32_t)d.k : (uint32_t)B * (uint32_t)N - 1u;
    d.wave_full = (N == W) && (slot0 + (uint32_t)(d.tid | 63) < slots);
    return d;
}

// _denormalized_action (atc_gym.py:318-335): action -> (v, h, phi) targets in the state's formats with the host-evaluated
// (multiplier, offset) pairs of derive().  Speed and heading: ONE float64 fma and the saturating, truncating conversion of the
// hardware (NaN -> 0) — the spec of include/atc_step.h.  Altitude (ABI 20): the reference's float64 target itself, a * m + c as one
// fma (altitude_target: evaluated where the step uses it).
__device__ __forceinline__ double altitude_target(const QRates& q, float a) { return __builtin_fma((double)a, q.dec_mh, q.dec_ch); }
__device__ __forceinline__ Targets decode_targets(const QRates& q, const Float3& act) {
    Targets t;
    t.v = cvt_u32_f64(__builtin_fma((double)act.a, q.dec_mv, q.dec_cv));
    t.ah = act.b;
    t.p = cvt_i32_f64(__builtin_fma((double)act.c, q.dec_mp, q.dec_cp));
    return t;
}
// Airplane.action_h (model.py:95-102): h + max(min(target - h, 15 dt), -41 dt) in float64, the reference's operations in its order
__device__ __forceinline__ double altitude_move(double h, double target, double dh_lo, double dh_hi) {
    return h + __builtin_fmax(__builtin_fmin(target - h, dh_hi), dh_lo);
}
// max(min(d, r), -r) for integer differences (the symmetric rate limits of speed and heading)
__device__ __forceinline__ int clamp_sym(int d, int r) { return max(min(d, r), -r); }
// |d| < D for a wrapped 32-bit difference d (atc_gym.py:305-306 on counts): one addition and one unsigned compare
__device__ __forceinline__ bool within(int d, int D) { return (uint32_t)d + (uint32_t)(D - 1) < (uint32_t)(2 * D - 1); }

// ---- first half of AtcGym.step: timestep, rate limits towards the targets, kinematics, MVA floor ---------------------
template <bool ONE, bool LAT, bool LDSG = false>
__device__ __forceinline__ Mid step_part_a(const float* __restrict__ grid, const QRates& q, const QKin& qk, const QGrid& qg,
                                           const LaneIds& d, uint32_t tv, double th, int tp, float act_p, LaneState& ls, EnvState& es,
                                           bool repeated, bool all_active, double* wide_named, int zk,
                                           uint64_t& refused_blk, bool refused_known ATC_TRACE_PARAM,
                                           const char* ltab = nullptr, const LdsTab* lt = nullptr) {
    Mid m;
    Aircraft& a = ls.a;
    // `repeated` (wave-uniform): this step repeats the previous step's actions and no env of the wavefront was reset in
    // between — last_action == the accepted targets, so nothing can be counted or changed (see ATC_M_ACTIONS_HELD)
    const bool book = !repeated;
    es.t += 1;  // atc_gym.py:135
    // all_active (wave-uniform, multi-step launches): the caller knows that every lane's aircraft is under control — the masks
    // only change in steps that end an episode or hand an aircraft over — and the per-lane bit test is skipped
    const bool active = all_active ||
                        (d.lane_valid && ((d.k < 32 ? ((uint32_t)es.amask >> d.k) : ((uint32_t)(es.amask >> 32) >> (d.k - 32))) & 1u));
    uint32_t fl = 0;
    float r = q.r_base;  // -0.05 * dt, atc_gym.py:137
    int acts = 0;
    // ---- _action_with_reward x3 (atc_gym.py:139-141,299-335) -> Airplane.action_* (model.py:60-120) ------------------
    // invalid target -> ValueError -> -1 reward, nothing applied, last_action kept (atc_gym.py:303-315); valid -> rate-limited
    // move, actions_taken++ unless |target - last| < discriminator.
    // Speed and heading are 32-bit fixed point (include/atc_step.h): the move is integer arithmetic — exact, like the
    // reference's float64 — with wrapping differences for the speed (valid speeds and the initial last_action 0 are less than
    // 2^31 counts apart) and saturating ones for the heading (its targets are not validated: any action is accepted).
    constexpr double h_min = kHMin, h_max = kHMax;   // (model.py:91-94: float64 compares with the float64 target)
    const bool valid_v = !(tv < kVMinFix || tv > kVMaxFix);
    const bool valid_h = !(th < h_min || th > h_max);
    // `plain` (wave-uniform): every lane of the wavefront flies an aircraft under control towards valid targets — the normal
    // case by far (a refused target or a handed-over aircraft in 64 is the exception).  Then nothing is conditional: no
    // select per state component, no refusal penalties, no flag bits.  Otherwise the branch-free general form below.  Both
    // evaluate the same expressions on the lanes they share.
    // (one lane mask per COMPARE, combined on the scalar unit: the ballot of a compound predicate is materialised per lane and
    // compared again — two vector operations per site)
    // ... and a heading target beyond the 32-bit range (the saturated conversion, include/atc_step.h ABI 19) takes the general
    // form too, like a lane whose heading or last heading target is WIDE already (`all_active` vouches that none is)
    // (the latency-bound multi-step instantiation carries this mask across the steps of a held action block: the targets do not
    // change inside one — six compares and five scalar ORs fewer on a lone wavefront's chain)
    uint64_t refused = refused_blk;
    if (!refused_known) {
        refused = __builtin_amdgcn_ballot_w64(tv < kVMinFix) | __builtin_amdgcn_ballot_w64(tv > kVMaxFix) |
                  __builtin_amdgcn_ballot_w64(th < h_min) | __builtin_amdgcn_ballot_w64(th > h_max) |
                  (__builtin_amdgcn_ballot_w64(tp == INT32_MAX) | __builtin_amdgcn_ballot_w64(tp == INT32_MIN));
        refused_blk = refused;
    }
    uint64_t special = refused;
    if (!all_active) {
        const int hi = book ? max(a.phi, ls.la_p) : a.phi, lo = book ? min(a.phi, ls.la_p) : a.phi;
        special |= __builtin_amdgcn_ballot_w64(!active) | __builtin_amdgcn_ballot_w64(hi == INT32_MAX) | __builtin_amdgcn_ballot_w64(lo == INT32_MIN);
    }
    const bool plain = special == 0ull;
    int phi_k;   // the heading as the kinematics see it (a WIDE heading wrapped: they are periodic)
    static_assert(kAMin == -kAMax && kPhiDotMin == -kPhiDotMax, "symmetric rate limits assumed (one count limit each)");
    if (ATC_USUAL(plain)) {
        a.v = a.v + (uint32_t)clamp_sym((int)(tv - a.v), q.rate_v);
        a.h = altitude_move(a.h, th, q.dh_lo, q.dh_hi);
        a.phi = a.phi + clamp_sym(sat_sub(tp, a.phi), q.rate_p);
        phi_k = a.phi;
        if (book) {
            acts = (!within((int)(tv - ls.la_v), kDiscrVFix) ? 1 : 0) + (!(__builtin_fabs(th - ls.la_h) < (double)kDiscrH) ? 1 : 0) +
                   (!within(sat_sub(tp, ls.la_p), kDiscrPhiFix) ? 1 : 0);
            ls.la_changed = ls.la_changed || tv != ls.la_v || th != ls.la_h || tp != ls.la_p;
            ls.la_v = tv;
            ls.la_h = th;
            ls.la_p = tp;
        }
    } else {
        {
            const bool valid = valid_v;
            const bool ok = valid && active;
            a.v = ok ? a.v + (uint32_t)clamp_sym((int)(tv - a.v), q.rate_v) : a.v;
            if (book) {
                acts += (ok && !within((int)(tv - ls.la_v), kDiscrVFix)) ? 1 : 0;
                ls.la_changed = ls.la_changed || (ok && tv != ls.la_v);
                ls.la_v = ok ? tv : ls.la_v;
            }
            r = valid ? r : r - 1.0f;
            fl |= valid ? 0u : (uint32_t)ATC_F_INVALID_V;
        }
        {
            const bool valid = valid_h;
            const bool ok = valid && active;
            const double h_new = altitude_move(a.h, th, q.dh_lo, q.dh_hi);
            a.h = ok ? h_new : a.h;
            if (book) {
                acts += (ok && !(__builtin_fabs(th - ls.la_h) < (double)kDiscrH)) ? 1 : 0;
                ls.la_changed = ls.la_changed || (ok && th != ls.la_h);
                ls.la_h = ok ? th : ls.la_h;
            }
            r = valid ? r : r - 1.0f;
            fl |= valid ? 0u : (uint32_t)ATC_F_INVALID_H;
        }
        {
            // heading (model.py:104-120: no validation, no wrap).  The 32-bit saturating form is exact while heading, target and last
            // target are inside the 32-bit range; lanes where one is not redo the move in 64-bit counts with the side array
            // (include/atc_step.h, ABI 19) — in the few wavefronts that hold such a lane.
            const int dd = clamp_sym(sat_sub(tp, a.phi), q.rate_p);
            int phi_new = active ? a.phi + dd : a.phi;
            bool counted = active && !within(sat_sub(tp, ls.la_p), kDiscrPhiFix);
            int la_new = active ? tp : ls.la_p;
            phi_k = phi_new;
            // (lanes without an aircraft compute on a clamped copy of the batch's last one: they must not write its side record)
            const bool wide = d.lane_valid && (is_wide(tp) || is_wide(a.phi) || (book && is_wide(ls.la_p)));
            if (ATC_RARE(__builtin_amdgcn_ballot_w64(wide) != 0ull)) {
                if (wide) {
                    double* w = at<double>(wide_base<ONE>(wide_named, zk), d.i * 32u);
                    bool lim;
                    const double T = phi_target_wide(__builtin_fma((double)act_p, q.dec_mp, q.dec_cp), &lim);
                    const double P = is_wide(a.phi) ? w[0] : (double)a.phi;
                    const double rate = (double)q.rate_p;
                    // (a handed-over aircraft keeps its heading; its views are refreshed all the same: they are scratch)
                    const double Pn = active ? P + __builtin_fmin(__builtin_fmax(T - P, -rate), rate) : P;
                    phi_new = cvt_i32_f64(Pn);   // saturating: sat32
                    phi_k = phi_new;
                    if (is_wide(phi_new)) {   // the exact counts, and their views for the rest of this step (wide_view)
                        phi_k = phi_wrap(Pn);
                        w[0] = Pn;
                        *reinterpret_cast<int2*>(w + 2) = make_int2(phi_k, __float_as_int(phi_obs_wide(Pn)));
                    }
                    if (book && active) {
                        const double L = is_wide(ls.la_p) ? w[1] : (double)ls.la_p;
                        counted = !(__builtin_fabs(T - L) < (double)kDiscrPhiFix);
                        la_new = cvt_i32_f64(T);
                        if (is_wide(la_new)) w[1] = T;
                    }
                    fl |= (lim && active) ? (uint32_t)ATC_F_PHI_LIMIT : 0u;
                }
            }
            a.phi = phi_new;
            if (book) {
                acts += counted ? 1 : 0;
                ls.la_changed = ls.la_changed || (active && la_new != ls.la_p);
                ls.la_p = la_new;
            }
        }
    }
    ATC_STAMP_TOP(trace_row, 2);
    // ---- Airplane.step (model.py:122-129): rot_matrix(phi) . [0, (v/3600) dt], float64 from the fixed-point state -----------
    // (an aircraft that is not under control does not move: zero speed -> zero displacement -> floor(0 + dither) = 0 counts)
    uint32_t v_move = a.v;
    if (!plain) {
        v_move = active ? a.v : 0u;
        asm("" : "+v"(v_move));   // (keeps the select on the 32-bit counts: the compiler moved it behind the conversion, onto both halves of the double)
    }
    // (the kinematics are periodic in the heading: a WIDE one goes in wrapped — read back from the side record's scratch word)
    advance<LAT>(qk, phi_k, v_move, es.t, a.x, a.y);
    ATC_STAMP_TOP(trace_row, 3);
    m.x32 = pos_to_real(q.pos_neg_k, qg.pos_x0, a.x);
    m.y32 = pos_to_real(q.pos_neg_k, qg.pos_y0, a.y);
    // MVA floor, first half: only ISSUE the lookup-cell gather here; nothing until the override chain needs its result
    m.lc = LdsCode{0u, 0u};
    if (LDSG) {
        m.cell.cell = make_float2(0.0f, 0.0f);
        m.lc = lds_cell_load(ltab, *lt, m.x32, m.y32);   // (LDS: no vector-memory operation on the step's chain)
    } else {
        m.cell = mva_cell_load(grid, qg.gh, m.x32, m.y32);
    }
    m.active = active;
    m.r = r;
    m.fl = fl;
    m.acts = acts;
    m.repeated = repeated;
    m.plain = plain;
    ATC_STAMP_TOP(trace_row, 4);
    return m;
}

// Extension (README.md:62): noise-abatement areas the aircraft is inside of and below the ceiling of — ATC_F_NOISE plus one
// bit per area in bits 16.. (consumed by the reward stage).  Evaluated while the lookup-grid cell is at hand: the cell names
// the areas whose bounds meet it, and almost every wavefront has no candidate at all.
__device__ __forceinline__ uint32_t noise_areas(const float* __restrict__ K, const float* __restrict__ grid, int n_areas,
                                               const MvaCell& c, float x, float y, double h) {
    const int n_noise = n_areas;
    uint32_t bits = 0;
    if (ATC_RARE(n_noise > 0)) {
        const uint32_t cand = noise_candidates(grid, c);
        if (__builtin_amdgcn_ballot_w64(cand != 0u) != 0ull) {
            for (int q = 0; q < n_noise; ++q) {
                const float* rec = K + (int)K[ATC_H_OFF_POLY] + ((int)K[ATC_H_N_MVA] + q) * ATC_P_WORDS;
                if (((cand >> q) & 1u) && in_bounds(rec, x, y) && h < (double)rec[ATC_P_HEIGHT] &&
                    ray_tracing(x, y, K + (int)rec[ATC_P_VOFF], (int)rec[ATC_P_NVERT]))
                    bits |= (uint32_t)ATC_F_NOISE | (0x10000u << q);
            }
        }
    }
    return bits;
}

// ---- second half: separation scan, win/timeout, observation, shaping, reductions, outputs, auto-reset --------------
// A by-value kernel argument group where it is needed: the single-step kernel names the argument (the compiler places its
// kernarg load), a multi-step launch re-reads it from the kernarg segment through this step's opaque zero — a scalar load
// inside the step instead of registers held (and spilled to vector-register lanes) across the whole step loop.
#define QGET(member) ((ONE || LAT) ? q.member : kernarg_reread<decltype(q.member)>(offsetof(StepArgs, q) + offsetof(StepDerived, member), zk))

// Separation scan horizon (scan_horizon_limits).  In a multi-step launch of the fast variant a full scan asks its question with
// thresholds no pair can close within the next `horizon` steps and notes which partner batches (four partner distances of the LDS
// scan) hold a pair inside them; the following `horizon` steps scan those batches only, with the exact minima — or nothing at all
// when none was noted.  A 64-aircraft env has no pair inside the thresholds of 6 steps in 66 % of its steps: 4 096 x 64 fused
// 3.8-3.95 vs 5.1-5.3 us per step (DESIGN_HISTORY.md §4).  Ships for the LDS-staged widths only.
#define ATC_SCAN_HORIZON_LDS 6
template <int W, bool FULL, bool ONE>
constexpr int scan_horizon() {
    return (FULL || ONE) ? 0 : (W >= 32 ? ATC_SCAN_HORIZON_LDS : 0);
}
constexpr float kScanHMax = 131072.0f;   // |altitude| below which an altitude step rounds by less than 2^-7 ft (scan_horizon_limits)

template <int W, bool FULL, bool ONE, bool LAT, bool LDSG = false>
__device__ __forceinline__ bool step_part_b(const float* __restrict__ K, const float* __restrict__ grid,
                                            const atc_params_t& p, const StepDerived& q, const QScan& qs, int zk, int N,
                                            const LaneIds& d, const Mid& m, LaneState& ls,
                                            EnvState& es, const StepOut& so, int32_t* stp, double* wide_named, float4* pos, float* obs_stage,
                                            const float* act_next, Float3& a_next, QRates& qr_next, int& scan_skip, uint32_t& scan_mask,
                                            const char* ltab = nullptr, const LdsTab* lt = nullptr) {
    Aircraft& a = ls.a;
    const bool active = m.active;
    const float x32 = m.x32, y32 = m.y32;
    // the altitude as every value-only formula sees it (glide-path test, shaping terms, separation scan, observation word 2): ONE
    // rounding of the float64 state; the flags that depend on the altitude alone compare the float64 (include/atc_step.h, ABI 20)
    const float hf = (float)a.h;
    float r = m.r;
    uint32_t fl = m.fl;
    int acts = m.acts;
    const int tid = d.tid, lane = d.lane, k = d.k, e = d.e;
    const uint32_t i = d.i;

    // ---- separation scan (extension; README.md:51): 3 nm / 1000 ft among aircraft active at step start ---------------
    // Every lane visits its env's other W-1 slots (never itself); aircraft that are not under control are staged at x = 1e18 so
    // that they neither conflict nor enter the minimum — branch-free.
    // The MVA cell (gather issued in the first half) is resolved AFTER the scan from W = 16 up (the round trip overlaps the scan)
    // and for one-aircraft envs (no scan: observation and shaping terms go first); W = 2 .. 8 resolve before it (the cell in flight
    // across the unrolled xor scan only costs registers).  Edge records per round trip in dirty cells: four for W = 1
    // (latency-bound, registers to spare), two otherwise.  Measurements: DESIGN_HISTORY.md §4, profiles/r03_experiments.txt.
    constexpr int kWalkBatch = (W == 1) ? 4 : ATC_MVA_BATCH;
    constexpr bool kResolveAfterScan = W >= 16 || W == 1;   // (W = 2 .. 8: the unrolled xor scan with the cell in flight costs 4 - 22 registers)
    WideWords ww = {0, 0};
    if (LAT) ww = wide_words<ONE>(m.plain, a.phi, wide_named, zk, i);
    float mva = 0.0f;
    int pi = 0;
    if (!kResolveAfterScan) {
        float hgt = 0.0f;
        pi = mva_resolve<kWalkBatch>(K, grid, QGET(g.gh), m.cell, x32, y32, &hgt);
        mva = hgt;   // (0 when outside: mva_resolve leaves the height at 0, atc_gym.py:161)
        fl |= noise_areas(K, grid, qs.n_noise, m.cell, x32, y32, a.h);
    }
    // Multi-step launches: the NEXT step's action is requested here — behind the MVA gathers (loads return in order: issued
    // earlier it would sit in front of them and its HBM latency would be paid at the MVA wait) and with the rest of the
    // step body (scan, corridor, observation, shaping, stores) still ahead to cover it.
    if (ATC_RARE(act_next != nullptr)) a_next = *at<Float3>(act_next, times12(i));
    float min_d2 = 1e30f;
    float margin = 1e30f;  // min over partners of max(d^2 - sep^2, |dh| - sep_ft): conflict iff negative
    // Multi-step launches of the fast variant (scan horizon, see scan_horizon_limits): a FULL scan asks with the horizon thresholds
    // and notes which partner batches hold a pair inside them (`flagged`); for the next kHorizon steps (scan_skip > 0) only those
    // batches can hold a pair that lost its separation — they alone are scanned, with the exact minima, and nothing at all when
    // none was flagged.
    constexpr int kHorizon = scan_horizon<W, FULL, ONE>();
    constexpr bool kHZ = kHorizon > 0;
    const ScanLimits lim = {qs.sep2, qs.sep_ft, kHZ ? qs.sep2_h : qs.sep2, kHZ ? qs.sep_ft_h : qs.sep_ft};
    const bool in_horizon = kHZ && scan_skip > 0;             // wave-uniform
    const uint32_t batches = in_horizon ? scan_mask : ~0u;
    if (in_horizon) scan_skip -= 1;
    uint32_t flagged = 0u;
    if (W > 1 && batches != 0u) {
        float xs = x32;
        if (!m.plain) xs = active ? x32 : 1e18f;
        const float sep2 = qs.sep2;
        if (W == 16 && !FULL) {
            int conf = 0;
            NearScan16H<1>::run(xs, y32, hf, sep2, qs.sep_ft, conf);
            margin = conf ? -1.0f : margin;
        } else if (W == 16) {
            PairScan16<1, FULL>::run(xs, y32, hf, sep2, qs.sep_ft, min_d2, margin);
        } else if (W <= 8) {
            pair_scan_xor<W, FULL>(xs, y32, hf, sep2, qs.sep_ft, min_d2, margin);
        } else {
            // W = 32 / 64: partners come from LDS and every unordered pair is evaluated ONCE — lane k visits the partners
            // k + 1 .. k + W/2 (mod W) of its group.  The result reaches the partner as the compare's 64-bit lane mask rotated by
            // the distance inside each group (scalar unit), no return traffic between lanes.  Each group is staged TWICE back to
            // back, one plane per coordinate (x | y | h | d^2 minimum; 8 W floats per group): partner k + d is the float d places
            // after the lane's own whatever k, two neighbouring partners arrive as a register pair (ds_read2_b32) for the packed
            // fp32 instructions (IEEE per component: bit-identical to the scalar forms).  Conflict = (d^2 < sep^2) & (|dh| < sep_ft),
            // the oracle's expression: two compares per pair, masks anded on the scalar unit — 4.5 VALU per pair.
            typedef float v2f __attribute__((ext_vector_type(2)));
            const int gbase = tid & ~(W - 1);
            constexpr int P = 2 * W;                   // floats per plane of a group
            float* own = reinterpret_cast<float*>(pos) + 8 * gbase + k;
            own[0] = xs;          own[W] = xs;
            own[P] = y32;         own[P + W] = y32;
            own[2 * P] = hf;      own[2 * P + W] = hf;
            if (FULL) { own[3 * P] = 1e36f; own[3 * P + W] = 1e36f; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            constexpr int H = W / 2;   // distances 1 .. H
            constexpr int U = 4;       // partners per LDS batch
            // A lost separation is RARE (it ends the episode): the four compare masks of an LDS batch are only ored into one
            // "anything?" word on the common path; the rotations run behind a wave-uniform test in the few batches that found a pair.
            uint64_t hit = 0;
            const v2f xs2 = {xs, xs}, ys2 = {y32, y32}, hs2 = {hf, hf};
            const float sep_ft = qs.sep_ft;
            // thresholds of this scan: the horizon's in a full scan, the minima themselves inside a horizon (and where there is none)
            const float t2 = in_horizon ? lim.sep2 : lim.sep2_h, tf = in_horizon ? lim.sep_ft : lim.sep_ft_h;
#pragma unroll 1   // (fully unrolled, the 2 H compare masks stay live together: 140-220 spilled SGPRs)
            for (int d0 = H - U + 1; d0 >= 1; d0 -= U) {
                const int batch = (d0 - 1) / U;   // partners d0 .. d0 + U - 1
                if (kHZ && !((batches >> batch) & 1u)) continue;   // wave-uniform: nobody inside the thresholds when the horizon began
                v2f qx[U / 2], qy[U / 2], qh[U / 2];
#pragma unroll
                for (int u = 0; u < U / 2; ++u) {
                    const float* q0 = own + d0 + 2 * u;
                    qx[u] = v2f{q0[0], q0[1]};
                    qy[u] = v2f{q0[P], q0[P + 1]};
                    qh[u] = v2f{q0[2 * P], q0[2 * P + 1]};
                }
                uint64_t mk[U];   // mk[2 u + w]: distance d0 + 2 u + w
                v2f d2k[U / 2], dhk[U / 2];   // (kHZ: kept for the exact question behind the wave-uniform test)
#pragma unroll
                for (int u = U / 2 - 1; u >= 0; --u) {
                    const v2f dx = xs2 - qx[u], dy = ys2 - qy[u];
                    const v2f d2 = __builtin_elementwise_fma(dx, dx, dy * dy);
                    d2k[u] = d2;
                    {
                        const v2f dh = hs2 - qh[u];
                        dhk[u] = dh;
                        // (two ballots anded as scalars: the compare masks themselves — a ballot of the anded predicate is
                        // materialised per lane and compared again)
                        // kHZ: the horizon thresholds (scan_horizon_limits) — a superset of the pairs that lost their separation
                        mk[2 * u] = __builtin_amdgcn_ballot_w64(d2[0] < t2) & __builtin_amdgcn_ballot_w64(fabsf(dh[0]) < tf);
                        mk[2 * u + 1] = __builtin_amdgcn_ballot_w64(d2[1] < t2) & __builtin_amdgcn_ballot_w64(fabsf(dh[1]) < tf);
                    }
                    if (FULL) {   // diagnostic minimum separation: the partner needs the VALUE -> LDS float minimum
#pragma unroll
                        for (int w = 1; w >= 0; --w) {
                            const int dd = d0 + 2 * u + w;   // uniform (H is a multiple of U: no tail)
                            min_d2 = fminf(min_d2, d2[w]);
                            // d^2 >= 0: the IEEE order of non-negative floats is the order of their bit patterns
                            atomicMin(reinterpret_cast<unsigned int*>(own - k + 3 * P + ((k + dd) & (W - 1))), __float_as_uint(d2[w]));
                        }
                    }
                }
                uint64_t any = mk[0];
#pragma unroll
                for (int j = 1; j < U; ++j) any |= mk[j];
                if (ATC_RARE(any != 0ull)) {   // some pair of this batch is inside the thresholds
                    if (kHZ && !in_horizon) {   // ... the horizon's: the batch stays on the list; now the exact question for it
                        flagged |= 1u << batch;
                        any = 0ull;
#pragma unroll
                        for (int j = 0; j < U; ++j) {
                            mk[j] = __builtin_amdgcn_ballot_w64(d2k[j / 2][j % 2] < sep2) & __builtin_amdgcn_ballot_w64(fabsf(dhk[j / 2][j % 2]) < sep_ft);
                            any |= mk[j];
                        }
                    }
                    // a pair that lost its separation: both of its lanes are marked
                    hit |= any;
#pragma unroll
                    for (int j = 0; j < U; ++j) {
                        const int dd = d0 + j;   // 1 .. W / 2, uniform
                        const uint64_t t = mk[j];
                        if (W == 64) {
                            hit |= (t << dd) | (t >> (64 - dd));
                        } else {  // two groups of 32 lanes: rotate inside each half
                            const uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
                            hit |= (uint64_t)((lo << dd) | (lo >> (32 - dd))) | ((uint64_t)((hi << dd) | (hi >> (32 - dd))) << 32);
                        }
                    }
                }
            }
            if ((hit >> lane) & 1ull) margin = -1.0f;
            if (FULL) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                min_d2 = fminf(min_d2, own[3 * P]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (kHZ && !in_horizon) {
            // the horizon's premises (scan_horizon_limits): no aircraft faster than 300 kt, altitudes of ordinary magnitude (a NaN fails)
            const uint64_t odd = __builtin_amdgcn_ballot_w64(a.v > kVMaxFix) | __builtin_amdgcn_ballot_w64(!(fabsf(hf) < kScanHMax));
            scan_skip = (odd != 0ull) ? 0 : kHorizon;
            scan_mask = flagged;
        }
    }
    ATC_STAMP_B(2);
    Obs ob;
    float shaping = 0.0f;
    const ObsConst oc = QGET(oc);
    // (W = 1 has no scan to cover the gather: there the observation goes first)
    // A dirty cell's first record (the LINE record of a split cell) is requested BEFORE the observation arithmetic, which then comes
    // before the resolve: the lookup's second dependent L2 trip overlaps ~100 instructions instead of stalling the wavefront.
    // Measured per width and launch form (r04, same box, two rounds): single steps of 16-aircraft envs 17.81 vs 18.03 us at
    // 65 536 envs, one-aircraft envs 6.58-6.61 vs 6.60-6.67 single / 3.42 vs 3.47-3.50 fused; NOT the fused 16-aircraft launch
    // (8 B scratch under its 80-register bound: 2.73-2.77 vs 2.66 us at 8 192 envs) and not the 64-aircraft kernels (8.50 vs 8.30).
    constexpr bool kPrefetch = (W == 1 || (ONE && W == 16)) && kResolveAfterScan && !LDSG;
    constexpr bool kObsFirst = kPrefetch || W == 1;
    MvaPre pre;
    if (kPrefetch) pre = mva_prefetch(grid, QGET(g.gh), m.cell);
    if (kObsFirst) {
        // (observation word 3 first, then the heading for the angles — each its own rare look-up for a WIDE heading, before the rest
        // of the observation occupies its registers)
        const WideWords wo = LAT ? ww : wide_words<ONE>(m.plain, a.phi, wide_named, zk, i);   // (ONE test for both words)
        const float phi_f = phi_real(wo.counts), phi_o = __int_as_float(wo.obs_bits);
        ob = get_state(oc, a.x, a.y, x32, y32, hf, phi_f, phi_o, v_real(a.v), hf);   // (word 5 = h - mva follows the resolve)
        if (p.mode & ATC_M_REWARD_SHAPING) shaping = shaping_total(shaping_core(oc, ob.d_faf, ob.phi_rel_faf, ob.o[9], hf, ob.on_gp));
    }
    ATC_STAMP_TOP(so.trace, 5);
    // ---- MVA floor (atc_gym.py:146-161), second half ---------------------------------------------------------------------
    if (LDSG) {
        // the LDS-resident table (csrc/atc_device.h: lds_resolve); the host attaches one only to sectors without noise-abatement areas
        float hgt = 0.0f;
        uint32_t walk;
        bool band;
        pi = lds_resolve(ltab, *lt, m.lc, x32, y32, &hgt, &walk, &band);
        if (ATC_RARE(__builtin_amdgcn_ballot_w64(walk != 0u) != 0ull)) {   // a vertex sub-cell: those lanes walk its records (one trip)
            if (walk != 0u) pi = lds_walk(*lt, walk, x32, y32, &hgt);
            asm volatile("" : "+v"(pi), "+v"(hgt));   // (waited for inside the rare block: see wide_view)
        }
        if (ATC_RARE(__builtin_amdgcn_ballot_w64(band) != 0ull)) {   // inside a line's margin band: the wavefront asks the grid
            const GridHdr gh = QGET(g.gh);
            const MvaCell c = mva_cell_load(grid, gh, x32, y32);
            pi = mva_resolve<kWalkBatch>(K, grid, gh, c, x32, y32, &hgt);
            asm volatile("" : "+v"(pi), "+v"(hgt));
        }
        mva = hgt;
    } else if (kResolveAfterScan) {
        float hgt = 0.0f;
        pi = mva_resolve<kWalkBatch>(K, grid, QGET(g.gh), m.cell, x32, y32, &hgt, kPrefetch ? &pre : nullptr);
        mva = hgt;                                 // atc_gym.py:161: mva = 0 outside (mva_resolve leaves the height at 0)
        fl |= noise_areas(K, grid, qs.n_noise, m.cell, x32, y32, a.h);
    }
    ATC_STAMP_END(so.trace, 1);
    // ---- the override chain (atc_gym.py:146-173) -----------------------------------------------------------------------
    // `quiet` (wave-uniform): nothing of it applies to any lane of this wavefront — every aircraft under control with accepted
    // targets (plain), inside the airspace at or above its MVA, no separation lost, no time-out, outside the bounds of the
    // corridor's horizontal triangle (the exact early-out of Runway.inside_corridor; asked of the lookup cell, which carries a
    // candidate bit), in no noise-abatement area.  That is
    // what almost every step of almost every wavefront looks like, and then reward and flag word are what the first half left
    // (base reward, no flag): the chain's selects run only in the other wavefronts — the same expressions, lane for lane.
    const bool conflict = margin < 0.0f;
    const bool timeout = es.t > qs.timestep_limit;
    // (one lane mask per compare, combined on the scalar unit — see `plain`)
    const uint64_t in_tri = LDSG ? __builtin_amdgcn_ballot_w64(lds_corridor_candidate(m.lc))
                          : grid ? __builtin_amdgcn_ballot_w64(corridor_candidate(m.cell))
                                 : (__builtin_amdgcn_ballot_w64(x32 >= qs.tri_bbox.x) & __builtin_amdgcn_ballot_w64(x32 <= qs.tri_bbox.z) &
                                    __builtin_amdgcn_ballot_w64(y32 >= qs.tri_bbox.y) & __builtin_amdgcn_ballot_w64(y32 <= qs.tri_bbox.w));
    const double mva_d = (double)mva;   // (an integer height; 0 outside the airspace)
    const bool quiet = m.plain &&
                       (__builtin_amdgcn_ballot_w64(pi < 0) | __builtin_amdgcn_ballot_w64(a.h < mva_d) | __builtin_amdgcn_ballot_w64(margin < 0.0f) |
                        __builtin_amdgcn_ballot_w64(es.t > qs.timestep_limit) | __builtin_amdgcn_ballot_w64(fl > 0xffffu) | in_tri) == 0ull;
    if (ATC_RARE(!quiet)) {
        {
            const bool below = pi >= 0 && a.h < mva_d;   // atc_gym.py:149: float64 altitude against the integer height
            r = pi < 0 ? -50.0f : (below ? -200.0f : r);
            fl |= pi < 0 ? (uint32_t)ATC_F_OUTSIDE : (below ? (uint32_t)ATC_F_BELOW_MVA : 0u);
        }
        {   // conflict override comes after the MVA overrides in the chain
            r = conflict ? qs.conflict_reward : r;
            fl |= conflict ? (uint32_t)ATC_F_CONFLICT : 0u;
        }
        // ---- win / timeout overrides (atc_gym.py:163-173) ---------------------------------------------------------------
        if (inside_corridor(K, qs.tri_bbox, x32, y32, hf, (double)(LAT ? ww.counts : heading_counts<ONE>(m.plain, a.phi, wide_named, zk, i)))) {
            int bonus = (qs.timestep_limit - es.t) * 5;
            bonus = bonus < 0 ? 0 : bonus;
            r = (float)(10000 + bonus);
            fl |= ATC_F_WON;
        }
        {
            r = timeout ? -200.0f : r;
            fl |= timeout ? (uint32_t)ATC_F_TIMEOUT : 0u;
        }
    }
    ATC_STAMP_B(3);
    ATC_STAMP_END(so.trace, 2);
    // ---- observation, shaping, noise areas, normalisation (atc_gym.py:175-189) ------------------------------------------
    // (multi-step launches: the normalisation constants are requested here, a hundred vector operations ahead of their use —
    // requested where they are used, the scalar load's whole latency was a stall)
    const QNorm qn = QGET(n);
    float o[ATC_OBS_DIM];
    float zraw[ATC_OBS_DIM];   // FULL only: raw observation (zeros for handed-over aircraft)
    {
        if (kObsFirst) {
            ob.o[5] = alt_above(a.h, mva);
        } else {
            const WideWords wo = LAT ? ww : wide_words<ONE>(m.plain, a.phi, wide_named, zk, i);
            const float phi_f = phi_real(wo.counts), phi_o = __int_as_float(wo.obs_bits);
            ob = get_state(oc, a.x, a.y, x32, y32, hf, phi_f, phi_o, v_real(a.v), alt_above(a.h, mva));
            if (p.mode & ATC_M_REWARD_SHAPING) shaping = shaping_total(shaping_core(oc, ob.d_faf, ob.phi_rel_faf, ob.o[9], hf, ob.on_gp));
        }
        // r += pos; r += ang; r += gs (atc_gym.py:179-185, after the override chain) as one addition of the factored sum
        // (value-only, within 1e-5)
        if (p.mode & ATC_M_REWARD_SHAPING) r += shaping;
        // extension (README.md:62): noise-abatement areas — which ones the aircraft is in was decided next to the MVA lookup
        // (bits 16.. of fl); the penalties are subtracted here, after the shaping terms, in area order.
        if (ATC_RARE(!quiet && __builtin_amdgcn_ballot_w64((fl >> 16) != 0u) != 0ull)) {
            const int n_noise = qs.n_noise;
            for (int q = 0; q < n_noise; ++q)
                if ((fl >> (16 + q)) & 1u) r -= (K + (int)K[ATC_H_OFF_POLY] + ((int)K[ATC_H_N_MVA] + q) * ATC_P_WORDS)[ATC_P_PENALTY];
            fl &= 0xffffu;
        }
        if (FULL) {
#pragma unroll
            for (int c = 0; c < ATC_OBS_DIM; ++c) zraw[c] = active ? ob.o[c] : 0.0f;  // zeros for handed-over aircraft
            if (so.raw_obs && d.lane_valid) store_obs(at<float>(so.raw_obs, times40(i)), zraw);
        }
        // atc_gym.py:187-189: (s - min - max/2) / (max/2) as one fma; the identity (1, -0) without ATC_M_NORMALIZE (derive())
#pragma unroll
        for (int c = 0; c < ATC_OBS_DIM; ++c) o[c] = fmaf(ob.o[c], qn.a[c], qn.b[c]);
    }
    // lanes without an aircraft under control: nothing happened.  Almost every wavefront has none, so the selects sit
    // behind a wave-uniform test.
    if (ATC_RARE(!m.plain && __builtin_amdgcn_ballot_w64(!active) != 0ull)) {
#pragma unroll
        for (int c = 0; c < ATC_OBS_DIM; ++c) o[c] = active ? o[c] : 0.0f;
        r = active ? r : 0.0f;
        acts = active ? acts : 0;
        fl = active ? fl : (d.lane_valid ? (uint32_t)ATC_F_INACTIVE : 0u);
        if (!active) min_d2 = 1e30f;
    }

    ATC_STAMP_B(4);
    ATC_STAMP_END(so.trace, 3);
    // ---- per-env reductions over the W lanes of the group ----------------------------------------------------------------
    const float env_r = group_sum<W>(r);
    const int env_acts = m.repeated ? 0 : group_sum_i<W>(acts);
    bool done = false, env_won = false;
    if (ATC_RARE(!quiet)) {
        const uint64_t won = group_ballot<W>((fl & ATC_F_WON) != 0, lane);
        const uint64_t term = group_ballot<W>(
            (fl & (ATC_F_BELOW_MVA | ATC_F_OUTSIDE | ATC_F_CONFLICT | ATC_F_TIMEOUT)) != 0, lane);
        // extension: an aircraft that reaches the corridor is handed over; the episode is won when all are.
        // ATC_M_KEEP_ACTIVE: the reference's rule (atc_gym.py:163-169) — any win ends the episode, nobody is handed over.
        const bool keep_active = (p.mode & ATC_M_KEEP_ACTIVE) != 0;
        if (!keep_active) es.amask &= ~won;
        env_won = keep_active ? won != 0 : es.amask == 0;
        done = d.env_valid && (term != 0 || env_won);
        // an env that restarts inside this step puts its aircraft somewhere else: the scan horizon ends here
        if (kHZ && (p.mode & ATC_M_AUTO_RESET) && __builtin_amdgcn_ballot_w64(done) != 0ull) scan_skip = 0;
    }
    es.total_reward += env_r;  // atc_gym.py:194-197
    es.n_actions += env_acts;

    // (flag word, reward and done are stored at the END of the step, behind the observation: see there)
    if (FULL && ONE && W == 1 && so.packet && d.lane_valid) {
        // The step result of a single-aircraft env as 9 self-validating 16-byte chunks (include/atc_step.h, atc_out_t.packet):
        // each chunk is ONE store carrying the caller's sequence tag, so a host polling mapped memory never mixes steps.
        const uint32_t tag = p.reserved0;
        const float w[27] = {o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9], zraw[0], zraw[1], zraw[2], zraw[3],
                             zraw[4], zraw[5], zraw[6], zraw[7], zraw[8], zraw[9], env_r,
                             __uint_as_float((fl & 0xffffu) | (done ? 0x10000u : 0u)), __int_as_float(es.t),
                             __int_as_float(es.n_actions), __int_as_float(a.x), __int_as_float(a.y), 0.0f};
        uint4* pk = at<uint4>(so.packet, (uint32_t)e * (ATC_PKT_CHUNKS * 16u));
#pragma unroll
        for (int c = 0; c < ATC_PKT_CHUNKS; ++c)
            store16_system(pk + c, __float_as_uint(w[3 * c]), __float_as_uint(w[3 * c + 1]), __float_as_uint(w[3 * c + 2]), tag);
    }
    if (FULL && so.min_sep) {
        const float m2 = (W > 1) ? group_min<W>(min_d2) : 1e30f;
        if (d.env_valid && k == 0) *at<float>(so.min_sep, (uint32_t)e * 4u) = (m2 >= 1e30f) ? 1e30f : sqrtf(m2);
    }

    if (ATC_RARE(done && (p.mode & ATC_M_AUTO_RESET))) {
        // VecEnv semantics: the env restarts inside the step; the returned obs is the RAW reset state
        // (atc_gym.py:351,365: reset() returns the un-normalised state computed with mva = 0).
        // The per-episode record is read, updated and written here and nowhere else on the step path.  In a multi-step launch
        // an earlier step of this wavefront may have written it (lane k == 0 writes, all lanes of the env read): wavefront-
        // scope fences order those accesses (they compile to nothing — a wavefront's memory operations are issued in order).
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (multi-step launches fetch the record's base here, on the rare path: carried from the top of the step it sat in
        // vector-register lanes)
        int32_t* stats = ONE ? stp : kernarg_reread<int32_t*>(offsetof(StepArgs, st) + offsetof(atc_state_t, stats), zk);
        int4* sr = at<int4>(stats, (uint32_t)e * (ATC_STAT_WORDS * 4u));
        const int4 s0 = sr[0];  // episodes, ep_length, ep_return, win_bits
        const int episode = s0.x;
        if (d.env_valid && k == 0) {
            const uint32_t win_bits = (((uint32_t)s0.w << 1) | (env_won ? 1u : 0u)) & 0x3ffu;
            sr[0] = make_int4(episode + 1, es.t, __float_as_int(es.total_reward), (int)win_bits);
            *at<int>(stats, (uint32_t)e * (ATC_STAT_WORDS * 4u) + ATC_STAT_EP_ACTIONS * 4u) = es.n_actions;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (d.lane_valid) {
            if (FULL && so.term_obs) store_obs(at<float>(so.term_obs, times40(i)), o);
            // (slot through this step's opaque zero: the spawn-record address is then formed here, on the rare path, instead of
            // being carried — and spilled — across the step loop as a 64-bit per-lane pointer)
            a = spawn(K, qs.off_spawn, p, e, k + zk, episode, o);   // state + raw reset observation from the blob's spawn records
        }
        es.total_reward = 0.0f;
        es.n_actions = 0;
        es.t = 0;
        es.amask = (N >= 64) ? ~0ull : ((1ull << N) - 1ull);
    }

    ATC_STAMP_B(5);
    ATC_STAMP_END(so.trace, 4);
    // Multi-step launches: the NEXT step's rate group is requested here, with the observation store still ahead — requested at
    // the top of the step that uses it, its latency was a stall before the first instruction of the kinematics.
    if (!ONE && !LAT) {
        int zn;
        asm volatile("s_mov_b32 %0, 0" : "=s"(zn));
        qr_next = kernarg_reread<QRates>(offsetof(StepArgs, q) + offsetof(StepDerived, r), zn);
    }
    // The next block's action (requested above, behind the MVA gathers) is WAITED FOR here, in the step that requested it and ahead
    // of this step's stores: left pending across the loop's back edge, "the action may be in flight" reaches the loop header, and the
    // decode at the top of EVERY step waits for every earlier vector-memory operation — one counter, in order — i.e. for the
    // previous step's observation stores to complete.  By now the load has long arrived.
    if (ATC_RARE(act_next != nullptr)) asm volatile("" : "+v"(a_next.a), "+v"(a_next.b), "+v"(a_next.c));
    // ---- observation store: [aircraft][10] rows are 40 B apart, so per-lane stores would scatter 8-byte pieces over 20
    //      cache lines per instruction; a full wavefront instead transposes its 64 x 10 block through LDS and writes 2 560
    //      contiguous bytes as 16-byte stores.
    if (ATC_USUAL(d.wave_full)) {
        // addresses from threadIdx itself, not from the lane ids a multi-step launch re-derives through an opaque zero: the
        // compiler then knows the ranges (lane < 64: two of the three row tests fold away, 24-bit multiplies suffice) — with
        // the opaque copies it emitted a quarter-rate 64-bit multiply-add per LDS read
        const uint32_t ln = threadIdx.x & 63u, wv = threadIdx.x >> 6;
        float* tb = obs_stage + __umul24(wv, 64u * ATC_OBS_DIM);
        float2* tb2 = reinterpret_cast<float2*>(tb) + __umul24(ln, ATC_OBS_DIM / 2);   // rows are 40 B: 8-byte aligned
#pragma unroll
        for (int c = 0; c < ATC_OBS_DIM / 2; ++c) tb2[c] = make_float2(o[2 * c], o[2 * c + 1]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // first aircraft of this wavefront (N == W): wave-uniform, so the multiply runs on the scalar unit
        const uint32_t wave_off = (d.slot0 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u))) * 40u;
        const float4* src = reinterpret_cast<const float4*>(tb);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint32_t idx = (uint32_t)j * 64u + ln;
            if (idx < 64u * ATC_OBS_DIM / 4u) {
                const float4 v = src[idx];
                float* d4 = at<float>(so.obs, wave_off + idx * 16u);
                typedef float v4f __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(d4));
            }
        }
        __builtin_amdgcn_wave_barrier();
    } else if (d.lane_valid) {
        store_obs(at<float>(so.obs, times40(i)), o);
    }
    ATC_STAMP_END(so.trace, 5);
    // ---- flag word, reward, done: last.  The compiler guards the observation's staging registers with a wait for ALL vector
    //      memory operations (one counter for loads and stores; the auto-reset path's loads merge in above) — issued before the
    //      observation, these three small stores were waited for there, a store round trip in every step.  Here the next such
    //      wait is the lookup-cell gather of the following step, a kinematics stage and a separation scan later.
    if (d.lane_valid) {
        stream_store(at<uint16_t>(so.flags, i * 2u), (uint16_t)fl);
        if (FULL && so.ac_reward) *at<float>(so.ac_reward, i * 4u) = r;
    }
    if (d.env_valid && k == 0) {
        *at<float>(so.reward, (uint32_t)e * 4u) = env_r;
        *at<uint8_t>(so.done, (uint32_t)e) = done ? 1 : 0;
    }
    return quiet;
}

__device__ __forceinline__ void store_lane_state(const atc_state_t& st, const LaneIds& d, const LaneState& ls, bool la_live) {
    if (d.lane_valid) {
        *at<int4>(st.ac, d.i * 16u) = make_int4(ls.a.x, ls.a.y, ls.a.phi, (int)ls.a.v);
        *at<double>(st.alt, d.i * 8u) = ls.a.h;
    }
    // the last-action targets are typically constant for many steps (actions are held): written back only by wavefronts in which
    // one of them changed
    if (__builtin_amdgcn_ballot_w64(ls.la_changed) != 0ull && d.lane_valid && la_live)
        *at<int4>(st.last_act, d.i * 16u) = make_int4((int)ls.la_v, ls.la_p, __double2loint(ls.la_h), __double2hiint(ls.la_h));
}
template <int W>
__device__ __forceinline__ void store_env_state(const atc_state_t& st, const LaneIds& d, const EnvState& es, uint32_t hi0) {
    if (d.env_valid && d.k == 0) {
        *at<int4>(st.env, (uint32_t)d.e * (ATC_ENV_WORDS * 4u)) =
            make_int4(es.t, es.n_actions, __float_as_int(es.total_reward), (int)(uint32_t)(es.amask & 0xffffffffu));
        if (W == 64 && (uint32_t)(es.amask >> 32) != hi0)
            *at<uint32_t>(st.stats, (uint32_t)d.e * (ATC_STAT_WORDS * 4u) + ATC_STAT_MASK_HI * 4u) = (uint32_t)(es.amask >> 32);
    }
}

#define ATC_LAT_WAVES 2   // wavefronts per SIMD the latency-bound instantiation is register-budgeted for (<= 256 VGPRs)
// ONE: single-step launch (T == 1); ALLV: every slot is an aircraft (make_ids); LAT: latency-bound multi-step instantiation (above)
// LDSG: the latency-bound launch of ONE-aircraft envs with the sector's lookup table staged in LDS (csrc/atc_device.h: LdsTab) — one
// workgroup per CU (the table's staged part is 112 KB for LOWW), chosen by the host when the launch has no more workgroups than the device CUs
template <int W, bool FULL, bool ONE, bool ALLV, bool LAT = false, bool LDSG = false>
__global__ void __launch_bounds__(kBlock, (LDSG ? 1 : LAT ? ATC_LAT_WAVES : ONE ? ATC_MIN_WAVES : ((FULL || W <= 8 || W >= 32) ? ATC_MIN_WAVES_LOOP - 1 : ATC_MIN_WAVES_LOOP)))
k_step(const float* __restrict__ blob, int off_grid, int B, int N, int T, int hold, atc_state_t st,
       const float* __restrict__ actions, atc_out_t out, atc_params_t p, StepDerived q_arg, InlineAction ia, LdsTab lt_arg) {
    static_assert(!LAT || (!ONE && !FULL), "LAT is an instantiation of the fast multi-step kernels");
    static_assert(!LDSG || (LAT && W == 1 && ALLV), "LDSG is an instantiation of the latency-bound one-aircraft kernel");
    StepDerived q_vec;
    if (LAT) q_vec = to_vector_registers(q_arg);
    const StepDerived& q = LAT ? q_vec : q_arg;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* pos = reinterpret_cast<float4*>(smem);                    // [2 kBlock] pair-scan staging (W >= 32)
    float* obs_stage = smem + (W >= 32 ? 2 * kBlock * 4 : 0);  // [4 waves][64 x 10] obs transpose
    const float* __restrict__ K = blob;  // the sector: uniform-index reads -> scalar loads
    const float* __restrict__ grid = off_grid ? blob + off_grid : nullptr;
#if ATC_TRACE
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned long long* trace = reinterpret_cast<unsigned long long*>(
        ((unsigned long long)p.reserved0) | ((unsigned long long)__float_as_uint(p.reserved1) << 32));
    const unsigned long long t_launch = __builtin_amdgcn_s_memtime();
#endif
    const uint32_t BN = (uint32_t)B * (uint32_t)N;      // host guarantees B*N*40 bytes < 4 GiB: 32-bit lane offsets
// single-step launches of envs of up to 16 aircraft (r05, same box: 8 192 x 16 5.15 vs 5.34 us, 65 536 x 16 17.53 vs 17.60, 65 536 x 1
// 6.03 vs 6.13 with the argument preload; 4 096 x 64 8.0 vs 7.8 and the multi-step launches no better: off there)
#define ATC_KARG_PREFETCH(W, ONE) ((ONE) && (W) <= 16)
    int karg_touch = 0;
    if (ATC_KARG_PREFETCH(W, ONE)) {
        // Touch every 64-byte line of the kernarg segment NOW, in one burst of scalar loads: the step reads its ~530 bytes of
        // arguments at a dozen places, each just ahead of its use — in a wavefront that is the first of its dispatch on its scalar
        // cache every one of those is a miss on the step's serial chain; requested together up front the misses overlap and the
        // later reads hit.
#if __HIP_DEVICE_COMPILE__
        typedef __attribute__((address_space(4))) const int* karg_ip;
        karg_ip kb = (karg_ip)__builtin_amdgcn_kernarg_segment_ptr();
#pragma unroll
        for (int ln = 1; ln < (int)((offsetof(StepArgs, lt) + 63) / 64); ++ln) karg_touch ^= kb[16 * ln];   // (lt: read by the LDSG launch only)
#endif
    }
    const LaneIds d = make_ids<W, ALLV>(blockIdx.x * kBlock, B, N);

    // ---- load persistent state (16-byte records; the W lanes of an env share the env record) -------------------------
    const int4 e0 = *at<int4>(st.env, (uint32_t)d.e * (ATC_ENV_WORDS * 4u));
    const uint32_t hi0 = (W == 64) ? *at<uint32_t>(st.stats, (uint32_t)d.e * (ATC_STAT_WORDS * 4u) + ATC_STAT_MASK_HI * 4u) : 0u;
    EnvState es = {e0.x, e0.y, __int_as_float(e0.z), (uint64_t)(uint32_t)e0.w | ((uint64_t)hi0 << 32)};
    const int4 ps = *at<int4>(st.ac, d.i * 16u);       // x, y, heading, speed
    const double h0 = *at<double>(st.alt, d.i * 8u);
    // ATC_M_ACTIONS_HELD (single-step launches): the caller repeats the previous launch's actions, so an aircraft that was under
    // control then has last_action == its accepted targets and cannot count an action or change the record — the 16-byte
    // record is only read (and written) by envs that were reset since their last step (timesteps == 0), whose aircraft
    // may have been handed over when the action block started and still carry an older record.
    Float3 act = {0.0f, 0.0f, 0.0f};   // action of the current block (held for `hold` steps); one 12-byte load per lane
    if (ONE && W == 1 && FULL && ia.set) {   // the single env of atc_step_packet: the action came with the kernel arguments
        act = Float3{ia.v, ia.h, ia.p};
    } else if (ONE) {
        act = *at<Float3>(actions, times12(d.i));   // requested before the env record is waited for below
    }
    const bool same_actions = ONE && (p.mode & ATC_M_ACTIONS_HELD) != 0;
    const bool la_live = !same_actions || e0.x == 0;
    int4 la0 = make_int4(0, 0, 0, 0);   // v, heading, altitude target (float64)
    if (la_live) la0 = *at<int4>(st.last_act, d.i * 16u);
    LaneState ls = {{ps.x, ps.y, h0, ps.z, (uint32_t)ps.w}, (uint32_t)la0.x, __hiloint2double(la0.w, la0.z), la0.y, false};

    // A single step is its own instantiation: with the step count a run-time value everything the loop carries (aircraft
    // and env records, output bases, hoisted sector constants) stays live across the whole body — the straight-line form
    // needs 56 VGPRs (N = 16), the loop form 80 under its launch bound (94 without it).
    const int n_steps = ONE ? 1 : T;
    const float* act_t = actions;   // action block of the current step; a block is held for `hold` steps
    int left = hold;                // steps the current block is still used for
    bool block_start = true;        // this step is the first of its block
    Targets tg = {0u, 0.0f, 0};   // decoded targets of the current step / block
    uint64_t refused_blk = 0ull;  // LAT: which lanes' speed / altitude / heading targets of the current block are refused or beyond range
    bool refused_known = false;
    bool all_active = false, mask_dirty = true;
    int scan_skip = 0;          // wave-uniform: steps left in the separation scan's horizon (step_part_b) ...
    uint32_t scan_mask = 0u;    // ... and the partner batches that have to be scanned inside it
    QRates qr_next = q.r;   // the rate group of the coming step (multi-step launches fetch it one step ahead, see step_part_b)
    if (!ONE && !LAT) {
        int zn;
        asm volatile("s_mov_b32 %0, 0" : "=s"(zn));
        qr_next = kernarg_reread<QRates>(offsetof(StepArgs, q) + offsetof(StepDerived, r), zn);
    }
    // LDSG: stage the lookup table — 28 sixteen-byte pieces per lane in flight at a time (ONE round trip for LOWW's 112 KB), behind
    // the state loads issued above; the terms a step reads with the table go to vector registers like the other uniform terms (LAT)
    LdsTab ltv = lt_arg;
    const char* ltab = nullptr;
    if (LDSG) {
        uint4* dst = reinterpret_cast<uint4*>(smem + (kBlock / 64) * 64 * ATC_OBS_DIM);   // behind the observation transpose stage
        const int n16 = lt_arg.n16;
        constexpr int kDepth = 28;   // (LOWW: 7 144 pieces = ONE batch of 28 x 256)
        for (int base = 0; base < n16; base += kDepth * kBlock) {
            uint4 v[kDepth];
#pragma unroll
            for (int u = 0; u < kDepth; ++u) {
                const int j = base + u * kBlock + (int)threadIdx.x;
                v[u] = lt_arg.src[min(j, n16 - 1)];
            }
            // (every load is USED here, unconditionally: left to the conditional stores below, the compiler sinks each load into its
            // store's exec-mask region — one round trip per piece instead of one per batch)
#pragma unroll
            for (int u = 0; u < kDepth; ++u) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
#pragma unroll
            for (int u = 0; u < kDepth; ++u) {
                const int j = base + u * kBlock + (int)threadIdx.x;
                if (j < n16) dst[j] = v[u];
            }
        }
        __syncthreads();
        ltab = reinterpret_cast<const char*>(dst);
        ltv.x0 = vg(lt_arg.x0); ltv.y0 = vg(lt_arg.y0); ltv.inv = vg(lt_arg.inv);
        ltv.nx = vg(lt_arg.nx); ltv.nx_last = vg(lt_arg.nx_last); ltv.ny_last = vg(lt_arg.ny_last);
        ltv.off_l1 = vg(lt_arg.off_l1); ltv.off_sub = vg(lt_arg.off_sub); ltv.off_line = vg(lt_arg.off_line); ltv.off_hts = vg(lt_arg.off_hts);
        ltv.off_resid = vg(lt_arg.off_resid);
    }
    if (!ONE) {
        // The state loads are WAITED FOR here, before the step loop.  Left pending, "a state register may still be in flight" is
        // merged into the loop header from the pre-header, and the compiler guards the first use of each in the loop body with a
        // wait that — one counter for loads and stores, in order — also waits for the PREVIOUS step's stores in every later step.
        asm volatile("" : "+v"(ls.a.x), "+v"(ls.a.y), "+v"(ls.a.h), "+v"(ls.a.phi), "+v"(ls.a.v));
        asm volatile("" : "+v"(ls.la_v), "+v"(ls.la_h), "+v"(ls.la_p), "+v"(es.t), "+v"(es.n_actions), "+v"(es.total_reward));
        uint32_t m_lo = (uint32_t)es.amask, m_hi = (uint32_t)(es.amask >> 32);
        asm volatile("" : "+v"(m_lo), "+v"(m_hi));
        es.amask = (uint64_t)m_lo | ((uint64_t)m_hi << 32);
    }
    for (int step = 0; step < n_steps; ++step) {
#if ATC_TRACE
        unsigned long long* trow = trace ? trace + ((size_t)(blockIdx.x * (kBlock / 64) + (tid >> 6)) * n_steps + step) * 8 : nullptr;
        if (lane == 0 && trow) trow[0] = step ? __builtin_amdgcn_s_memtime() : t_launch;
#endif
        const size_t sBN = (size_t)step * BN, sB = (size_t)step * (uint32_t)B;  // uniform (scalar) per-step bases
        // Multi-step launches, history: before the by-value arguments were re-read from the kernarg segment (below), everything
        // invariant across steps (lane ids, the address arithmetic on them, the sector base) had to be re-derived from an opaque
        // zero inside the body, or LICM kept it alive around the loop and the 80-VGPR form spilled 88 B per lane.  With the
        // re-reads the plain form is the faster one (12-20 B of scratch in some widths notwithstanding: 13.70 vs 13.80 us per
        // step, 8 192 x 16 3.98 vs 4.05, profiles/r02_experiments.txt); the knob remains for A/B builds.
        LaneIds dl = d;
        const float* Kl = K;
        const float* gl = grid;
        atc_params_t pl = p;
        atc_out_t outl = out;
        int32_t* stats_l = st.stats;
        int zk = 0;   // this step's opaque zero: kernarg re-reads that depend on it cannot be hoisted out of the step loop
        if (!ONE) {   // (also makes the mode word's flag tests scalar compares inside the step, not
            asm volatile("s_mov_b32 %0, 0" : "=s"(zk));   // 64-bit lane masks kept — and spilled — across the loop)
            pl.mode += (uint32_t)zk;
        }
        StepOut so = {outl.obs + sBN * ATC_OBS_DIM, outl.flags + sBN, outl.reward + sB, outl.done + sB,
                      FULL && outl.raw_obs ? outl.raw_obs + sBN * ATC_OBS_DIM : nullptr,
                      FULL && outl.ac_reward ? outl.ac_reward + sBN : nullptr,
                      FULL && outl.min_sep ? outl.min_sep + sB : nullptr,
                      FULL && outl.term_obs ? outl.term_obs + sBN * ATC_OBS_DIM : nullptr,
                      FULL && outl.packet ? outl.packet + sB * (ATC_PKT_CHUNKS * 4) : nullptr
#if ATC_TRACE
                      , trow
#endif
        };
        // Carried across the steps of a multi-step launch: the DECODED targets and the refused-target mask of a held action block
        // (r05: 65 536 x 16 fused 11.8-12.0 -> 11.6-11.8 us) and "every lane's aircraft is under control" (re-established after
        // steps that can change a mask: 11.37 vs 11.46 us).
        const QRates qr = qr_next;
        const QScan qs = QGET(s);   // (requested here, consumed after the kinematics)
        if (!ONE && ATC_RARE(step == 0)) act = *at<Float3>(act_t, times12(dl.i));
        if (ONE || step == 0) tg = decode_targets(qr, act);
        if (ONE && !la_live) {   // held block: the record equals the accepted targets (components that are refused are not compared)
            ls.la_v = tg.v;
            ls.la_h = altitude_target(qr, tg.ah);
            ls.la_p = tg.p;
        }
        // multi-step launches know structurally which steps repeat an action block
        const bool repeated = ONE ? (same_actions && __builtin_amdgcn_ballot_w64(la_live) == 0ull)
                                  : (!block_start && __builtin_amdgcn_ballot_w64(es.t == 0) == 0ull);
        // ... and whether every lane's aircraft is under control: re-established after the steps in which a mask can change
        if (!ONE && ATC_RARE(mask_dirty)) {
            // (and no lane's heading or last heading target is WIDE: include/atc_step.h ABI 19 — such lanes take the general form)
            all_active = (__builtin_amdgcn_ballot_w64(!(dl.lane_valid && ((dl.k < 32 ? ((uint32_t)es.amask >> dl.k) : ((uint32_t)(es.amask >> 32) >> (dl.k - 32))) & 1u))) |
                          __builtin_amdgcn_ballot_w64(max(ls.a.phi, ls.la_p) == INT32_MAX) | __builtin_amdgcn_ballot_w64(min(ls.a.phi, ls.la_p) == INT32_MIN)) == 0ull;
            mask_dirty = false;
        }
        ATC_STAMP_TOP(trow, 1);
        const Mid m = step_part_a<ONE, LAT, LDSG>(gl, qr, QGET(k), QGET(g), dl, tg.v, altitude_target(qr, tg.ah), tg.p, act.c, ls, es, repeated, !ONE && all_active,
                                             st.phi_wide, zk, refused_blk, refused_known ATC_TRACE_PASS(trow), ltab, &ltv);
        refused_known = true;
        ATC_STAMP(1);
        Float3 nxt = act;
        const float* act_next = nullptr;   // the next block is fetched during the last step of the current one
        block_start = false;
        if (!ONE && ATC_RARE(--left == 0)) {   // (the block length is fetched again here, at block ends, not kept — or re-fetched — every step)
            left = kernarg_reread<int>(offsetof(StepArgs, hold), zk);
            block_start = true;
            refused_known = false;
            act_t += (size_t)BN * 3;
            if (step + 1 < n_steps) act_next = act_t;
        }
        const bool quiet = step_part_b<W, FULL, ONE, LAT, LDSG>(Kl, gl, pl, q, qs, zk, N, dl, m, ls, es, so, stats_l, st.phi_wide, pos, obs_stage, act_next, nxt, qr_next, scan_skip, scan_mask, ltab, &ltv);
        if (ATC_RARE(!quiet)) mask_dirty = true;
        if (act_next) tg = decode_targets(QGET(r), nxt);
        act = nxt;
        ATC_STAMP(6);
        ATC_STAMP_TOP(trow, 6);
        ATC_STAMP_END(trow, 6);
    }
    // ---- write back persistent state -----------------------------------------------------------------------------------
    atc_state_t st_end = st;
    // (single-step launches also fetch the state pointers again for the final stores instead of carrying ten scalar registers
    // through the body: they were spilled to vector-register lanes)
    if (ONE || loop_rereads_state(W)) {
        int zk;
        asm volatile("s_mov_b32 %0, 0" : "=s"(zk));
        st_end = kernarg_reread<atc_state_t>(offsetof(StepArgs, st), zk);
    }
    store_lane_state(st_end, d, ls, la_live);
    store_env_state<W>(st_end, d, es, hi0);
    if (ATC_KARG_PREFETCH(W, ONE)) asm volatile("" ::"s"(karg_touch));   // (keeps the touches alive; nothing waits for them before here)
#if ATC_TRACE
    if (lane == 0 && trace) trace[((size_t)(blockIdx.x * (kBlock / 64) + (tid >> 6)) * n_steps + (n_steps - 1)) * 8 + 7] = __builtin_amdgcn_s_memtime();
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent step server of ONE env x ONE aircraft (the drop-in AtcGym, atc_gym.py:128-192; include/atc_step.h: atc_serve_*).
// One resident wavefront keeps the env's state in registers and polls a mailbox in pinned mapped host memory: the host writes
// {action, sequence number} with ONE 16-byte store, the wavefront runs the single-step body (the same step_part_a / step_part_b
// as k_step<1, FULL, ONE>) and answers with the self-validating result packet (atc_out_t.packet).  No launch, no kernel-argument
// upload and no state round trip per step: what is left is two crossings of the host link and the step's own dependent chain.
// It LEAVES — state written back, mailbox status updated — on the quit command or after `lease_ticks` of s_memrealtime (100 MHz)
// without a command: a host that died or went to train for a minute never leaves a wavefront spinning.
// ---------------------------------------------------------------------------------------------------------------
enum { ATC_MB_CMD = 0, ATC_MB_SEQ = 3, ATC_MB_STATE = 4, ATC_MB_LAST = 5, ATC_MB_WORDS = 16 };
enum { ATC_SERVE_IDLE = 0, ATC_SERVE_RUNNING = 1, ATC_SERVE_LEFT_LEASE = 2, ATC_SERVE_LEFT_QUIT = 3 };
#define ATC_SERVE_QUIT 0xffffffffu
__device__ __forceinline__ uint32_t mb_load(const uint32_t* w) {
    return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(64, 1)
k_serve(const float* __restrict__ blob, int off_grid, atc_state_t st, atc_out_t out, atc_params_t p, StepDerived q_arg,
        uint32_t* mailbox, uint32_t last, unsigned long long lease_ticks) {
    // (one wavefront alone on its SIMD: the uniform terms live in vector registers like in the latency-bound instantiation — as
    // named scalar arguments kept across the serving loop, 151 of them were spilled to vector-register lanes)
    const StepDerived q = to_vector_registers(q_arg);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const float* __restrict__ K = blob;
    const float* __restrict__ grid = off_grid ? blob + off_grid : nullptr;
    const LaneIds d = make_ids<1, false>(0u, 1, 1);   // lane 0 is the aircraft; the other lanes compute on copies and store nothing
    const int4 e0 = *at<int4>(st.env, 0u);
    EnvState es = {e0.x, e0.y, __int_as_float(e0.z), (uint64_t)(uint32_t)e0.w};
    const int4 ps = *at<int4>(st.ac, 0u);
    const double h0 = *at<double>(st.alt, 0u);
    const int4 la0 = *at<int4>(st.last_act, 0u);
    LaneState ls = {{ps.x, ps.y, h0, ps.z, (uint32_t)ps.w}, (uint32_t)la0.x, __hiloint2double(la0.w, la0.z), la0.y, false};
    const StepOut so = {out.obs, out.flags, out.reward, out.done, out.raw_obs, out.ac_reward, out.min_sep, out.term_obs, out.packet
#if ATC_TRACE
                        , nullptr
#endif
    };
    if (threadIdx.x == 0) __hip_atomic_store(mailbox + ATC_MB_STATE, (uint32_t)ATC_SERVE_RUNNING, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned long long t_cmd = __builtin_amdgcn_s_memrealtime();
    uint32_t left_as = ATC_SERVE_LEFT_LEASE;
    for (;;) {
        // (every lane reads the same word: one request; made wave-uniform for the control flow)
        const uint32_t seq = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(mailbox + ATC_MB_SEQ, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM));
        if (seq == ATC_SERVE_QUIT) {
            left_as = ATC_SERVE_LEFT_QUIT;
            break;
        }
        if (seq != last + 1u) {
            if (__builtin_amdgcn_s_memrealtime() - t_cmd > lease_ticks) break;
            continue;
        }
        // the action travelled in the same 16-byte host store as its sequence number
        const Float3 act = {__uint_as_float(mb_load(mailbox + ATC_MB_CMD)), __uint_as_float(mb_load(mailbox + ATC_MB_CMD + 1)),
                            __uint_as_float(mb_load(mailbox + ATC_MB_CMD + 2))};
        last = seq;
        atc_params_t pl = p;
        pl.reserved0 = seq;   // the packet's tag (step_part_b)
        const Targets tg = decode_targets(q.r, act);
        uint64_t refused = 0ull;
        const Mid m = step_part_a<true, true>(grid, q.r, q.k, q.g, d, tg.v, altitude_target(q.r, tg.ah), tg.p, act.c, ls, es, false, false,
                                               st.phi_wide, 0, refused, false ATC_TRACE_PASS(nullptr));
        Float3 nxt = act;
        QRates qn = q.r;
        int scan_skip = 0;
        uint32_t scan_mask = 0u;
        step_part_b<1, true, true, true>(K, grid, pl, q, q.s, 0, 1, d, m, ls, es, so, st.stats, st.phi_wide, nullptr, smem, nullptr, nxt, qn,
                                          scan_skip, scan_mask);
        t_cmd = __builtin_amdgcn_s_memrealtime();
    }
    store_lane_state(st, d, ls, true);
    store_env_state<1>(st, d, es, 0u);
    __threadfence_system();
    if (threadIdx.x == 0) {
        __hip_atomic_store(mailbox + ATC_MB_LAST, last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(mailbox + ATC_MB_STATE, left_as, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#include "atc_aux_kernels.inc"   // k_reset, k_observe, k_reset_env, k_query_*

// ---------------------------------------------------------------------------------------------------------------
// host side of the C-ABI
// ---------------------------------------------------------------------------------------------------------------
static int grid_for(const atc_scenario* s, long long threads) {
    long long blocks = (threads + kBlock - 1) / kBlock;
    const long long cap = (long long)s->n_cu * ATC_GRID_CAP;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}
static size_t lds_bytes(const atc_scenario*, bool pair_scan, bool step_kernel = false) {
    size_t w = 0;  // the sector is not staged: LDS only holds the obs transpose stage and the pair-scan staging
    if (step_kernel) w += (size_t)(kBlock / 64) * 64 * ATC_OBS_DIM;
    if (pair_scan) w += (size_t)2 * kBlock * 4;
    return w * sizeof(float);
}

template <int W, bool FULL, bool ONE, bool ALLV, bool LAT = false, bool LDSG = false>
static int launch_step2(const atc_scenario* s, int B, int N, int T, int hold, const atc_state_t* st, const float* actions,
                        const atc_out_t* out, const atc_params_t* p, hipStream_t stream) {
    const size_t lds = lds_bytes(s, W >= 32, true) + (LDSG ? s->lds_tab_bytes : 0);
    if (lds > 48 * 1024) {   // (once per device and instantiation: the attribute sticks)
        static thread_local int raised_for = -1;
        if (raised_for != s->device) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step<W, FULL, ONE, ALLV, LAT, LDSG>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, s->max_lds));
            raised_for = s->device;
        }
    }
    const long long slots = (long long)B * W;
    const int grid = (int)((slots + kBlock - 1) / kBlock);  // one workgroup per 256 slots, no grid-stride loop
    LdsTab lt = s->lt;
    if (!LDSG) lt.src = nullptr;
    hipLaunchKernelGGL((k_step<W, FULL, ONE, ALLV, LAT, LDSG>), dim3(grid), dim3(kBlock), lds, stream, s->d_blob, s->off_grid, B, N, T, hold, *st, actions, *out, *p, derive(*p, s, scan_horizon<W, FULL, ONE>()), inline_action(), lt);
    HIP_TRY(hipGetLastError());
    return ATC_OK;
}
template <int W>
static int launch_step(const atc_scenario* s, int B, int N, int T, int hold, const atc_state_t* st, const float* actions,
                       const atc_out_t* out, const atc_params_t* p, hipStream_t stream) {
    const bool full = out->raw_obs || out->ac_reward || out->min_sep || out->term_obs || out->packet;
    // every slot of the launch is an aircraft: the fast variant then runs its all-valid instantiation (make_ids)
    const bool allv = !full && N == W && ((long long)B * W) % kBlock == 0;
    // Multi-step launches keep the state in registers across the steps; the run-time step loop costs the kernel its
    // occupancy (4 wavefronts per SIMD against 5-7 for the straight-line single step) but issuing T single-step launches
    // instead is slower at every size (65 536 x 16, T = 20, [T, ...] outputs: 35.0 vs 27.1 us per step).
    if (T > 1) {
        if (full) return launch_step2<W, true, false, false>(s, B, N, T, hold, st, actions, out, p, stream);
        // at most ATC_LAT_WAVES wavefronts per SIMD: the latency-bound instantiation (uniform terms in vector registers)
        const bool lat = allv && ((long long)B * W + 63) / 64 <= (long long)ATC_LAT_WAVES * 4 * s->n_cu;
        // ... of one-aircraft envs, at most one workgroup per CU, a lookup table attached: the table lives in LDS for the launch
        if constexpr (W == 1) {
            if (lat && s->lt.src && (long long)B <= (long long)kBlock * s->n_cu && s->off_grid &&
                lds_bytes(s, false, true) + s->lds_tab_bytes <= (size_t)s->max_lds)
                return launch_step2<1, false, false, true, true, true>(s, B, N, T, hold, st, actions, out, p, stream);
        }
        if (lat) return launch_step2<W, false, false, true, true>(s, B, N, T, hold, st, actions, out, p, stream);
        return allv ? launch_step2<W, false, false, true>(s, B, N, T, hold, st, actions, out, p, stream)
                    : launch_step2<W, false, false, false>(s, B, N, T, hold, st, actions, out, p, stream);
    }
    if (full) return launch_step2<W, true, true, false>(s, B, N, 1, 1, st, actions, out, p, stream);
    return allv ? launch_step2<W, false, true, true>(s, B, N, 1, 1, st, actions, out, p, stream)
                : launch_step2<W, false, true, false>(s, B, N, 1, 1, st, actions, out, p, stream);
}

// argument checks shared by every entry point that touches the env state
static int check_env_args(const atc_scenario_t* s, int B, int N, const atc_state_t* st, const atc_params_t* p) {
    if (!s || !st || !p) return fail_arg("null pointer");
    if (B < 1 || N < 1 || N > ATC_MAX_AIRCRAFT) return fail_arg("need B >= 1, 1 <= N <= 64");
    if (!st->ac || !st->alt || !st->last_act || !st->env || !st->stats || !st->phi_wide) return fail_arg("atc_state_t has a null field");
    if ((unsigned long long)B * N * ATC_OBS_DIM * 4ull >= (1ull << 32) || (unsigned long long)B * 64ull >= (1ull << 32))
        return fail_arg("B*N too large for one launch (B*N*40 bytes must stay below 4 GiB): split the batch");
    return ATC_OK;
}

static int step_common(const atc_scenario_t* s, int B, int N, int T, int hold, const atc_state_t* st, const float* actions,
                       const atc_out_t* out, const atc_params_t* p, void* stream) {
    if (!actions || !out) return fail_arg("null pointer");
    if (const int rc = check_env_args(s, B, N, st, p)) return rc;
    if (T < 1 || hold < 1) return fail_arg("need T >= 1 and hold >= 1");
    // actions is [T / hold] blocks: a partial last block would be read beyond what the header promises exists
    if (T % hold != 0) return fail_arg("T must be a multiple of hold (actions holds T / hold blocks)");
    if (!out->obs || !out->reward || !out->done || !out->flags) return fail_arg("obs/reward/done/flags are required");
    if (out->packet && (N != 1 || T != 1)) return fail_arg("atc_out_t.packet is for single steps of single-aircraft envs");
    if (!(p->dt > 0.0)) return fail_arg("dt must be > 0");
    // the fixed-point formats of include/atc_step.h: a step's displacement (<= 512 kt) must stay below 2^30 position-grid
    // counts and the speed's rate limit below 2^31 speed counts — dt up to 51 s for any sector (the reference uses 1 s)
    if (!(0.1423 * p->dt * (double)s->consts[ATC_C_POS_SCALE] < 1073741824.0) || !((double)kAMax * p->dt < 255.9))
        return fail_arg("dt too large for the fixed-point state formats (see include/atc_step.h)");
    hipStream_t q = (hipStream_t)stream;
    if (N == 1) return launch_step<1>(s, B, N, T, hold, st, actions, out, p, q);
    if (N == 2) return launch_step<2>(s, B, N, T, hold, st, actions, out, p, q);
    if (N <= 4) return launch_step<4>(s, B, N, T, hold, st, actions, out, p, q);
    if (N <= 8) return launch_step<8>(s, B, N, T, hold, st, actions, out, p, q);
    if (N <= 16) return launch_step<16>(s, B, N, T, hold, st, actions, out, p, q);
    if (N <= 32) return launch_step<32>(s, B, N, T, hold, st, actions, out, p, q);
    return launch_step<64>(s, B, N, T, hold, st, actions, out, p, q);
}

#include "atc_abi.inc"   // the extern "C" entry points (host side)
