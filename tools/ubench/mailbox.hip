// Developer micro-benchmark (gfx950): round trip of a host <-> persistent-kernel mailbox in mapped (fine-grained) host memory.
// The host writes a sequence number, one resident wavefront polls it over the host link and answers into a second mapped
// word, the host polls that.  Bounded: the kernel leaves after `max_polls` polls without a new command.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mailbox.hip -o build_variants/mailbox && build_variants/mailbox
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <emmintrin.h>

__global__ void __launch_bounds__(64) k_server(volatile uint32_t* cmd, volatile uint32_t* ack, uint32_t last, uint32_t n_cmds, uint32_t max_polls, int work) {
    uint32_t idle = 0;
    float x = 1.0f;
    while (last < n_cmds) {
        const uint32_t seq = __hip_atomic_load(const_cast<uint32_t*>(cmd), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (seq == last + 1) {
            const float a0 = reinterpret_cast<volatile float*>(cmd)[1];
            for (int i = 0; i < work; ++i) x = x * 1.0001f + a0;   // stand-in for the step's dependent chain
            last = seq;
            idle = 0;
            if (threadIdx.x == 0) {
                reinterpret_cast<volatile float*>(ack)[1] = x;
                __hip_atomic_store(const_cast<uint32_t*>(ack), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        } else if (++idle > max_polls) {
            break;
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(const_cast<uint32_t*>(ack) + 2, last + 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    uint32_t *cmd_h, *ack_h, *cmd_d, *ack_d;
    if (hipHostMalloc(&cmd_h, 64, hipHostMallocMapped) != hipSuccess || hipHostMalloc(&ack_h, 64, hipHostMallocMapped) != hipSuccess) return 1;
    hipHostGetDevicePointer((void**)&cmd_d, cmd_h, 0);
    hipHostGetDevicePointer((void**)&ack_d, ack_h, 0);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int work : {0, 256, 512}) {
        const uint32_t N = 20000;
        for (int i = 0; i < 16; ++i) { cmd_h[i] = 0; ack_h[i] = 0; }
        hipLaunchKernelGGL(k_server, dim3(1), dim3(64), 0, st, cmd_d, ack_d, 0u, N, 2000000u, work);
        const auto t0 = std::chrono::steady_clock::now();
        bool ok = true;
        for (uint32_t s = 1; s <= N && ok; ++s) {
            reinterpret_cast<volatile float*>(cmd_h)[1] = 0.5f;
            __atomic_store_n(cmd_h, s, __ATOMIC_RELEASE);
            uint64_t spins = 0;
            while (__atomic_load_n(ack_h, __ATOMIC_ACQUIRE) != s) {
                _mm_pause();
                if (++spins > 400000000ull) { ok = false; break; }
            }
        }
        const auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(st);
        printf("work %4d dependent fmas: %s, %.2f us per round trip (%u commands); server left with %08x\n", work, ok ? "ok" : "TIMEOUT",
               std::chrono::duration<double, std::micro>(t1 - t0).count() / N, N, ack_h[2]);
    }
    // for comparison: one trivial kernel launch + stream synchronise per command
    {
        const uint32_t N = 5000;
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t s = 1; s <= N; ++s) {
            cmd_h[0] = s;
            hipLaunchKernelGGL(k_server, dim3(1), dim3(64), 0, st, cmd_d, ack_d, s - 1, s, 1000u, 0);
            while (__atomic_load_n(ack_h, __ATOMIC_ACQUIRE) != s) _mm_pause();
        }
        hipStreamSynchronize(st);
        const auto t1 = std::chrono::steady_clock::now();
        printf("one launch per command, answer polled in mapped memory: %.2f us per command\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
    return 0;
}
