// Can a kernel on a second stream start when the kernel before it has DISPATCHED its last workgroup (its queue
// has run dry) instead of when it has ended?  Two mechanisms, both fed by a flag the first kernel's last-started
// workgroup writes:  (1) hipStreamWaitValue32 on the second stream, (2) a one-wave gate kernel that polls the flag.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/chain tools/ubench/chain.hip
// Prints, per mechanism, when the follower started relative to the first kernel's last dispatch and its end.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// every workgroup spins `ticks[blockIdx.x]` wall-clock ticks (100 MHz); the one that starts LAST writes `seq` to *flag
__global__ void __launch_bounds__(64) filler(const int *ticks, unsigned *started, unsigned n, unsigned *flag, unsigned seq,
                                             unsigned long long *t_dry, unsigned long long *t_end) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        const unsigned k = atomicAdd(started, 1u);
        if (k == n - 1) {
            *t_dry = t0;
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const long long d = ticks[blockIdx.x];
    while ((long long)(wall_clock64() - t0) < d) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) atomicMax(t_end, wall_clock64());
}
__global__ void __launch_bounds__(64) gate(const unsigned *flag, unsigned seq) {
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) __builtin_amdgcn_s_sleep(32);
}
__global__ void __launch_bounds__(256) follower(unsigned long long *t_first, unsigned long long *t_last, int spin) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        atomicMin(t_first, t0);
    }
    while ((long long)(wall_clock64() - t0) < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) atomicMax(t_last, wall_clock64());
}

int main(int argc, char **argv) {
    int can = -1;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const unsigned n = 9216;  // three rounds of 3072 wave slots if the filler took all registers; it does not -- what
                              // matters here is only that the last dispatch is well before the end
    std::vector<int> ticks(n);
    for (unsigned i = 0; i < n; i++) ticks[i] = 200000 + (int)(i % 7) * 30000 + (i >= n - 64 ? 600000 : 0);  // 2-3.8 ms, stragglers 8 ms
    int *d_ticks;
    unsigned *d_started, *d_flag_dev, *d_flag_sig = nullptr;
    unsigned long long *d_t;
    CHECK(hipMalloc(&d_ticks, 4 * n));
    CHECK(hipMemcpy(d_ticks, ticks.data(), 4 * n, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_started, 4));
    CHECK(hipMalloc(&d_flag_dev, 4));
    if (hipExtMallocWithFlags((void **)&d_flag_sig, 8, hipMallocSignalMemory) != hipSuccess) {
        printf("hipExtMallocWithFlags(hipMallocSignalMemory) failed: %s\n", hipGetErrorString(hipGetLastError()));
        d_flag_sig = nullptr;
    }
    CHECK(hipMalloc(&d_t, 8 * 4));
    hipEvent_t ea;
    CHECK(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
    unsigned seq = 0;
    for (int mech = 0; mech < 4; mech++) {
        // 0: follower waits for the filler's END (event); 1: hipStreamWaitValue32 on plain device memory;
        // 2: hipStreamWaitValue32 on signal memory; 3: gate kernel
        const char *names[] = {"event (end of kernel)", "hipStreamWaitValue32, hipMalloc flag", "hipStreamWaitValue32, signal memory", "gate kernel"};
        unsigned *flag = mech == 2 ? d_flag_sig : d_flag_dev;
        if (!flag) { printf("%-40s skipped\n", names[mech]); continue; }
        for (int rep = 0; rep < 3; rep++) {
            ++seq;
            unsigned long long init[4] = {0, 0, ~0ull, 0};
            CHECK(hipMemcpy(d_t, init, sizeof init, hipMemcpyHostToDevice));
            CHECK(hipMemset(d_started, 0, 4));
            if (rep == 0) CHECK(hipMemset(flag, 0, 4));
            CHECK(hipDeviceSynchronize());
            hipLaunchKernelGGL(filler, dim3(n), dim3(64), 0, sa, d_ticks, d_started, n, flag, seq, d_t + 0, d_t + 1);
            CHECK(hipGetLastError());
            if (mech == 0) {
                CHECK(hipEventRecord(ea, sa));
                CHECK(hipStreamWaitEvent(sb, ea, 0));
            } else if (mech == 1 || mech == 2) {
                hipError_t e = hipStreamWaitValue32(sb, flag, seq, hipStreamWaitValueGte, 0xffffffffu);
                if (e != hipSuccess) { printf("%-40s hipStreamWaitValue32: %s\n", names[mech], hipGetErrorString(e)); (void)hipGetLastError(); CHECK(hipDeviceSynchronize()); break; }
            } else {
                hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, sb, flag, seq);
            }
            hipLaunchKernelGGL(follower, dim3(2048), dim3(256), 0, sb, d_t + 2, d_t + 3, 50000);
            CHECK(hipGetLastError());
            CHECK(hipDeviceSynchronize());
            unsigned long long t[4];
            CHECK(hipMemcpy(t, d_t, sizeof t, hipMemcpyDeviceToHost));
            printf("%-40s rep %d: filler dry -> end %.2f ms; follower start - dry %+.3f ms, follower start - filler end %+.3f ms, follower end - filler end %+.3f ms\n",
                   names[mech], rep, (t[1] - t[0]) * 1e-5, ((double)t[2] - (double)t[0]) * 1e-5, ((double)t[2] - (double)t[1]) * 1e-5,
                   ((double)t[3] - (double)t[1]) * 1e-5);
        }
    }
    return 0;
}
