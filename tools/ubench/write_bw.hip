// Developer micro-benchmark (gfx950): what a launch that does nothing but WRITE reaches on this part, for the output pattern of the
// fused step launch (atc_rollout_hold, 65 536 envs x 16 aircraft, T = 20) — the question behind `roofline.frac` of that launch: its
// algorithmic bytes are 97 % stores (44 B written per aircraft-step against 2.6 B read), and the 8 TB/s of the roofline is a
// read-or-mixed figure.
//   stream   : every lane stores 16 bytes, consecutive lanes consecutive addresses, one pass over the buffer (the ideal write stream)
//   pattern  : the step launch's own stores, nothing else — per wavefront and step 2 560 contiguous bytes of observation rows as
//              16-byte stores, 2 bytes of flags per lane, 4 + 1 bytes per env from every 16th lane; the T outputs of a wavefront are
//              B N 40 bytes apart (the [T][B][N][10] layout), steps in order
//   pattern + k FMAs per lane and step: the same with a dependent chain of k v_fma_f32 between the steps' stores (k = 300: the
//              instruction count of the step) — how much of the arithmetic the stores hide
// Buffers are 0.98 GB (beyond the 256 MiB Infinity Cache); times are HIP events over REPS launches after a warm-up.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/write_bw.hip -o build_variants/write_bw && build_variants/write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ void __launch_bounds__(256) k_stream(v4f* dst, size_t n16, float seed) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) {
        const v4f v = {seed, seed + 1.0f, seed + 2.0f, (float)threadIdx.x};
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}
// each thread writes `per` consecutive 16-byte pieces of a grid-strided stream (fewer, longer-lived workgroups)
template <bool NT>
__global__ void __launch_bounds__(256) k_stream_loop(v4f* dst, size_t n16, int per, float seed) {
    const size_t chunk = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int j = 0; j < per && i < n16; ++j, i += chunk) {
        const v4f v = {seed, seed + 1.0f, seed + 2.0f, (float)j};
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

// PARTS: bit 0 observation rows, bit 1 flags, bit 2 reward + done.  SWZ: workgroup -> tile mapping that keeps neighbouring tiles on
// ONE XCD (workgroups are dealt round-robin to the 8 XCDs: tile = (block mod 8) * (grid / 8) + block / 8), so that the partial
// lines of the per-env outputs (16 envs per workgroup: 64 B of reward, 16 B of done) meet in one L2 before they are written back
template <bool NT, int FMAS, int PARTS = 7, bool SWZ = false>
__global__ void __launch_bounds__(256) k_pattern(float* obs, unsigned short* flags, float* reward, unsigned char* done, int B, int T,
                                                 float seed) {
    const unsigned tid = threadIdx.x, ln = tid & 63u;
    const size_t BN = (size_t)B * 16;
    const unsigned blk = SWZ ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned i = blk * 256u + tid;                 // aircraft
    const unsigned wave_first = blk * 256u + (tid & ~63u);
    float x = seed + (float)tid, y = seed * 0.5f;
    for (int t = 0; t < T; ++t) {
#pragma unroll 16
        for (int k = 0; k < FMAS; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(y));
        char* ob = reinterpret_cast<char*>(obs) + ((size_t)t * BN + wave_first) * 40;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned idx = (unsigned)j * 64u + ln;
            if ((PARTS & 1) && idx < 160u) {
                const v4f v = {x, y, (float)t, (float)idx};
                v4f* p = reinterpret_cast<v4f*>(ob + idx * 16u);
                if (NT) __builtin_nontemporal_store(v, p);
                else *p = v;
            }
        }
        if (PARTS & 2) {
            if (NT) __builtin_nontemporal_store((unsigned short)t, flags + (size_t)t * BN + i);
            else flags[(size_t)t * BN + i] = (unsigned short)t;
        }
        if ((PARTS & 4) && (tid & 15u) == 0) {
            reward[(size_t)t * B + (i >> 4)] = x;
            done[(size_t)t * B + (i >> 4)] = (unsigned char)(t & 1);
        }
    }
}

// The step kernel also LOADS once per step (the lookup-grid gather: 4 bytes per lane from a table that lives in the L2s) and has to
// wait for the result before it can finish the step.  On gfx9 loads and stores share ONE in-order counter (vmcnt): waiting for a
// load that was issued AFTER the previous step's stores means waiting for those stores to be acknowledged too.
//   PIPE = false: gather(t) issued after stores(t - 1) — the step kernel's order (the gather needs step t's position)
//   PIPE = true : gather(t + 1) issued BEFORE stores(t): its wait leaves the younger stores outstanding
template <bool PIPE, int FMAS>
__global__ void __launch_bounds__(256) k_pattern_ld(float* obs, unsigned short* flags, float* reward, unsigned char* done,
                                                    const float* __restrict__ table, unsigned mask, int B, int T, float seed) {
    const unsigned tid = threadIdx.x, ln = tid & 63u;
    const size_t BN = (size_t)B * 16;
    const unsigned i = blockIdx.x * 256u + tid;
    const unsigned wave_first = blockIdx.x * 256u + (tid & ~63u);
    float x = seed + (float)tid, y = seed * 0.5f;
    unsigned h = i * 2654435761u;
    float g = 0.0f;
    if (PIPE) g = table[(h >> 8) & mask];
    for (int t = 0; t < T; ++t) {
        if (!PIPE) {
            asm volatile("" ::: "memory");
            g = table[(h >> 8) & mask];
        }
#pragma unroll 16
        for (int k = 0; k < FMAS; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(y));
        x += g;   // the wait
        asm volatile("" : "+v"(x));
        h = h * 1664525u + 1013904223u;
        if (PIPE) {
            g = table[(h >> 8) & mask];
            asm volatile("" ::: "memory");
        }
        char* ob = reinterpret_cast<char*>(obs) + ((size_t)t * BN + wave_first) * 40;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned idx = (unsigned)j * 64u + ln;
            if (idx < 160u) __builtin_nontemporal_store(v4f{x, y, (float)t, (float)idx}, reinterpret_cast<v4f*>(ob + idx * 16u));
        }
        __builtin_nontemporal_store((unsigned short)t, flags + (size_t)t * BN + i);
        if ((tid & 15u) == 0) {
            reward[(size_t)t * B + (i >> 4)] = x;
            done[(size_t)t * B + (i >> 4)] = (unsigned char)(t & 1);
        }
    }
}

// The same launch with the per-step lookup served from LDS instead of global memory (round-5 review, next #2 / #4: would an
// LDS-resident sector table lift the fused launch?).  A vector load shares the CU's in-order vector-memory pipe with the stores of
// every resident wavefront and the vmcnt counter with the wavefront's own; a ds_read uses neither.  Each workgroup first STAGES its
// table (LDS_KB kilobytes, 16-byte loads from a table that lives in the L2s: the cost an LDS-resident table pays per workgroup).
template <int FMAS, int LDS_KB>
__global__ void __launch_bounds__(256) k_pattern_lds(float* obs, unsigned short* flags, float* reward, unsigned char* done,
                                                     const float* __restrict__ table, int B, int T, float seed) {
    __shared__ float lds[LDS_KB * 256];
    const unsigned tid = threadIdx.x, ln = tid & 63u;
    for (unsigned j = tid; j < LDS_KB * 64u; j += 256u)
        reinterpret_cast<v4f*>(lds)[j] = reinterpret_cast<const v4f*>(table)[j];
    __syncthreads();
    const size_t BN = (size_t)B * 16;
    const unsigned i = blockIdx.x * 256u + tid;
    const unsigned wave_first = blockIdx.x * 256u + (tid & ~63u);
    float x = seed + (float)tid, y = seed * 0.5f;
    unsigned h = i * 2654435761u;
    for (int t = 0; t < T; ++t) {
        const float g = lds[(h >> 8) % (LDS_KB * 256u)];
#pragma unroll 16
        for (int k = 0; k < FMAS; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(y));
        x += g;
        asm volatile("" : "+v"(x));
        h = h * 1664525u + 1013904223u;
        char* ob = reinterpret_cast<char*>(obs) + ((size_t)t * BN + wave_first) * 40;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned idx = (unsigned)j * 64u + ln;
            if (idx < 160u) __builtin_nontemporal_store(v4f{x, y, (float)t, (float)idx}, reinterpret_cast<v4f*>(ob + idx * 16u));
        }
        __builtin_nontemporal_store((unsigned short)t, flags + (size_t)t * BN + i);
        if ((tid & 15u) == 0) {
            reward[(size_t)t * B + (i >> 4)] = x;
            done[(size_t)t * B + (i >> 4)] = (unsigned char)(t & 1);
        }
    }
}

// The observation rows in an ENV-major layout [B][T][N][10] (each env's T steps contiguous: 12.8 KB per env, a workgroup's 16 envs
// 200 KB) instead of the time-major [T][B][N][10]: is the many-fronts penalty the layout's?
__global__ void __launch_bounds__(256) k_pattern_env_major(float* obs, int B, int T, float seed) {
    const unsigned tid = threadIdx.x, ln = tid & 63u;
    const unsigned env0 = (blockIdx.x * 256u + (tid & ~63u)) >> 4;   // first env of the wavefront
    float x = seed + (float)tid, y = seed * 0.5f;
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned idx = (unsigned)j * 64u + ln;
            if (idx < 160u) {
                const unsigned env = idx / 40u, w = idx - env * 40u;
                char* p = reinterpret_cast<char*>(obs) + ((size_t)(env0 + env) * T + t) * 640 + w * 16u;
                __builtin_nontemporal_store(v4f{x, y, (float)t, (float)idx}, reinterpret_cast<v4f*>(p));
            }
        }
    }
}

// Store flavours on the observation rows of the pattern (gfx950 cache-policy bits: nt, sc0, sc1 — MI355X_MICROARCH.md: plain / sc0 / nt
// keep the line in the XCD's L2, sc1 / sc0 sc1 write it through): FLAVOUR 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 sc1 nt
template <int FLAVOUR>
__global__ void __launch_bounds__(256) k_rows_flavour(float* obs, int B, int T, float seed) {
    const unsigned tid = threadIdx.x, ln = tid & 63u;
    const size_t BN = (size_t)B * 16;
    const unsigned wave_first = blockIdx.x * 256u + (tid & ~63u);
    float x = seed + (float)tid, y = seed * 0.5f;
    for (int t = 0; t < T; ++t) {
        char* ob = reinterpret_cast<char*>(obs) + ((size_t)t * BN + wave_first) * 40;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned idx = (unsigned)j * 64u + ln;
            if (idx < 160u) {
                const v4f v = {x, y, (float)t, (float)idx};
                v4f* p = reinterpret_cast<v4f*>(ob + idx * 16u);
                if (FLAVOUR == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
                if (FLAVOUR == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
                if (FLAVOUR == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
                if (FLAVOUR == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
                if (FLAVOUR == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
                if (FLAVOUR == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
            }
        }
    }
}

template <typename F>
static double time_us(F launch, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    std::vector<double> ts;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(a, 0));
        for (int i = 0; i < reps; ++i) launch();
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        ts.push_back(ms * 1e3 / reps);
    }
    double best = ts[0];
    for (double t : ts) best = t < best ? t : best;
    return best;
}

// --json B T: three figures for bench.py's record of the fused launch (one JSON line): the ideal write stream, the launch's store
// pattern alone, and the pattern with 400 FMAs and one gather per lane-step (the step's instruction count and its one load)
static int json_mode(int B, int T) {
    const int N = 16;
    const size_t BN = (size_t)B * N;
    const size_t obs_bytes = BN * 40 * T, flag_bytes = BN * 2 * T, rew_bytes = (size_t)B * 4 * T, done_bytes = (size_t)B * T;
    const size_t total = obs_bytes + flag_bytes + rew_bytes + done_bytes;
    float *obs, *reward, *table;
    unsigned short* flags;
    unsigned char* done;
    const unsigned tmask = (1u << 20) - 1u;
    CHECK(hipMalloc(&obs, obs_bytes));
    CHECK(hipMalloc(&flags, flag_bytes));
    CHECK(hipMalloc(&reward, rew_bytes));
    CHECK(hipMalloc(&done, done_bytes));
    CHECK(hipMalloc(&table, (size_t)(tmask + 1) * 4));
    CHECK(hipMemset(table, 0, (size_t)(tmask + 1) * 4));
    const size_t n16 = obs_bytes / 16;
    const unsigned grid = (unsigned)(BN / 256);
    const int reps = 10;
    const double stream = time_us([&] { hipLaunchKernelGGL(k_stream<false>, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<v4f*>(obs), n16, 1.0f); }, reps);
    const double pattern = time_us([&] { hipLaunchKernelGGL((k_pattern<true, 0>), dim3(grid), dim3(256), 0, 0, obs, flags, reward, done, B, T, 1.0f); }, reps);
    const double loaded = time_us([&] { hipLaunchKernelGGL((k_pattern_ld<false, 400>), dim3(grid), dim3(256), 0, 0, obs, flags, reward, done, table, tmask, B, T, 1.0f); }, reps);
    // other arrangements of the same bytes (round-5 review, next #4): tiles kept XCD-contiguous, and the bound of ANY scheme that
    // buffers the small per-step streams (flags, reward, done) — the observation rows alone
    const double xcd = time_us([&] { hipLaunchKernelGGL((k_pattern<true, 0, 7, true>), dim3(grid), dim3(256), 0, 0, obs, flags, reward, done, B, T, 1.0f); }, reps);
    const double rows = time_us([&] { hipLaunchKernelGGL((k_pattern<true, 0, 1>), dim3(grid), dim3(256), 0, 0, obs, flags, reward, done, B, T, 1.0f); }, reps);
    const double rows400 = time_us([&] { hipLaunchKernelGGL((k_pattern<true, 400, 1>), dim3(grid), dim3(256), 0, 0, obs, flags, reward, done, B, T, 1.0f); }, reps);
    printf("{\"envs\": %d, \"aircraft\": %d, \"T\": %d, \"output_bytes_per_launch\": %zu, \"stream_tb_per_s\": %.3f, "
           "\"store_pattern_us_per_launch\": %.1f, \"store_pattern_tb_per_s\": %.3f, "
           "\"store_pattern_400_fma_1_gather_us_per_launch\": %.1f, "
           "\"arrangements_us_per_launch\": {\"as_launched\": %.1f, \"xcd_contiguous_tiles\": %.1f, "
           "\"observation_rows_only (bound of buffering flags / reward / done)\": %.1f, \"observation_rows_only + 400 FMAs\": %.1f}}\n",
           B, N, T, total, obs_bytes / stream / 1e6, pattern, total / pattern / 1e6, loaded, pattern, xcd, rows, rows400);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 4 && std::string(argv[1]) == "--json") return json_mode(atoi(argv[2]), atoi(argv[3]));
    const int B = 65536, N = 16, T = 20;
    const size_t BN = (size_t)B * N;
    const size_t obs_bytes = BN * 40 * T, flag_bytes = BN * 2 * T, rew_bytes = (size_t)B * 4 * T, done_bytes = (size_t)B * T;
    const size_t total = obs_bytes + flag_bytes + rew_bytes + done_bytes;
    float* obs;
    unsigned short* flags;
    float* reward;
    unsigned char* done;
    CHECK(hipMalloc(&obs, obs_bytes));
    CHECK(hipMalloc(&flags, flag_bytes));
    CHECK(hipMalloc(&reward, rew_bytes));
    CHECK(hipMalloc(&done, done_bytes));
    printf("output bytes of one T = %d launch at %d x %d: %.1f MB\n", T, B, N, total / 1e6);
    const size_t n16 = obs_bytes / 16;
    const int reps = 20;
    auto report = [&](const char* name, double us, size_t bytes) {
        printf("%-58s %8.1f us  %6.2f TB/s  (per step %.2f us)\n", name, us, bytes / us / 1e6, us / T);
    };
    report("stream, one 16-B store per lane, plain", time_us([&] { hipLaunchKernelGGL(k_stream<false>, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<v4f*>(obs), n16, 1.0f); }, reps), obs_bytes);
    report("stream, one 16-B store per lane, nontemporal", time_us([&] { hipLaunchKernelGGL(k_stream<true>, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<v4f*>(obs), n16, 1.0f); }, reps), obs_bytes);
    for (int per : {8, 32}) {
        char name[96];
        const unsigned grid = (unsigned)((n16 / per + 255) / 256);
        snprintf(name, sizeof name, "stream, %d grid-strided stores per lane, plain", per);
        report(name, time_us([&] { hipLaunchKernelGGL(k_stream_loop<false>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<v4f*>(obs), n16, per, 1.0f); }, reps), obs_bytes);
        snprintf(name, sizeof name, "stream, %d grid-strided stores per lane, nontemporal", per);
        report(name, time_us([&] { hipLaunchKernelGGL(k_stream_loop<true>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<v4f*>(obs), n16, per, 1.0f); }, reps), obs_bytes);
    }
    const unsigned grid = (unsigned)(BN / 256);
#define PAT(NTV, F, LABEL) report(LABEL, time_us([&] { hipLaunchKernelGGL((k_pattern<NTV, F>), dim3(grid), dim3(256), 0, 0, obs, flags, reward, done, B, T, 1.0f); }, reps), total)
    PAT(false, 0, "step-launch pattern, stores only, plain");
    PAT(true, 0, "step-launch pattern, stores only, nontemporal");
#define PATX(P, SW, BYTES, LABEL) report(LABEL, time_us([&] { hipLaunchKernelGGL((k_pattern<true, 0, P, SW>), dim3(grid), dim3(256), 0, 0, obs, flags, reward, done, B, T, 1.0f); }, reps), BYTES)
    PATX(1, false, obs_bytes, "  observation rows only");
    PATX(3, false, obs_bytes + flag_bytes, "  observation rows + flags");
    PATX(5, false, obs_bytes + rew_bytes + done_bytes, "  observation rows + reward / done");
    report("  observation rows only, ENV-major layout [B][T][N][10]", time_us([&] { hipLaunchKernelGGL(k_pattern_env_major, dim3(grid), dim3(256), 0, 0, obs, B, T, 1.0f); }, reps), obs_bytes);
#define FLV(F, LABEL) report(LABEL, time_us([&] { hipLaunchKernelGGL((k_rows_flavour<F>), dim3(grid), dim3(256), 0, 0, obs, B, T, 1.0f); }, reps), obs_bytes)
    FLV(0, "  observation rows only, store flavour: plain");
    FLV(1, "  observation rows only, store flavour: nt");
    FLV(2, "  observation rows only, store flavour: sc1");
    FLV(3, "  observation rows only, store flavour: sc0 sc1");
    FLV(4, "  observation rows only, store flavour: sc1 nt");
    FLV(5, "  observation rows only, store flavour: sc0 sc1 nt");
    PATX(7, true, total, "step-launch pattern, XCD-contiguous tiles");
    PATX(1, true, obs_bytes, "  observation rows only, XCD-contiguous tiles");
    PATX(5, true, obs_bytes + rew_bytes + done_bytes, "  observation rows + reward / done, XCD-contiguous tiles");
    PAT(true, 100, "step-launch pattern + 100 dependent FMAs per lane-step");
    PAT(true, 200, "step-launch pattern + 200 dependent FMAs per lane-step");
    PAT(true, 300, "step-launch pattern + 300 dependent FMAs per lane-step");
    PAT(true, 400, "step-launch pattern + 400 dependent FMAs per lane-step");
    float* table;
    const unsigned tmask = (1u << 20) - 1u;   // 4 MB of floats: resident in every XCD's L2
    CHECK(hipMalloc(&table, (size_t)(tmask + 1) * 4));
    CHECK(hipMemset(table, 0, (size_t)(tmask + 1) * 4));
#define PATL(PIPE, F, LABEL) report(LABEL, time_us([&] { hipLaunchKernelGGL((k_pattern_ld<PIPE, F>), dim3(grid), dim3(256), 0, 0, obs, flags, reward, done, table, tmask, B, T, 1.0f); }, reps), total)
    PATL(false, 300, "pattern + 300 FMAs + gather(t) issued after stores(t-1)");
    PATL(true, 300, "pattern + 300 FMAs + gather(t+1) issued before stores(t)");
    PATL(false, 400, "pattern + 400 FMAs + gather(t) issued after stores(t-1)");
    PATL(true, 400, "pattern + 400 FMAs + gather(t+1) issued before stores(t)");
#define PATS(F, KB, LABEL) report(LABEL, time_us([&] { hipLaunchKernelGGL((k_pattern_lds<F, KB>), dim3(grid), dim3(256), 0, 0, obs, flags, reward, done, table, B, T, 1.0f); }, reps), total)
    PATS(300, 16, "pattern + 300 FMAs + lookup in LDS (16 KB staged per workgroup)");
    PATS(400, 16, "pattern + 400 FMAs + lookup in LDS (16 KB staged per workgroup)");
    PATS(400, 24, "pattern + 400 FMAs + lookup in LDS (24 KB staged per workgroup: 6 workgroups per CU)");
    PATS(400, 40, "pattern + 400 FMAs + lookup in LDS (40 KB staged per workgroup: 4 workgroups per CU)");
    PATS(400, 64, "pattern + 400 FMAs + lookup in LDS (64 KB staged per workgroup: 2 workgroups per CU)");
    return 0;
}
