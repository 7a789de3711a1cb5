// Micro-benchmark: issue rate of the VALU / DPP / LDS-crossbar instructions the DP kernels are made of,
// per SIMD, at 1..8 waves per SIMD (gfx950).  Prints cycles per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rate tools/ubench/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 2000;
constexpr int kUnroll = 16;  // independent chains per wave (ILP)

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, float a, float b, int n_iter) {
    float x[kUnroll];
    unsigned u[kUnroll];
#pragma unroll
    for (int i = 0; i < kUnroll; i++) {
        x[i] = a + (float)(threadIdx.x + i);
        u[i] = threadIdx.x * 7 + i;
    }
    float c = b;
    for (int it = 0; it < n_iter; it++) {
#pragma unroll
        for (int i = 0; i < kUnroll; i++) {
            if constexpr (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if constexpr (OP == 1) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if constexpr (OP == 2) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(a));
            if constexpr (OP == 3) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(c) : "vcc");
            if constexpr (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(c) : "vcc");
            if constexpr (OP == 5) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 7) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
            if constexpr (OP == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(reinterpret_cast<double *>(&x[i & ~1]))) : "v"(*(reinterpret_cast<double *>(&x[(i & ~1)]))));
            if constexpr (OP == 9) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(c) : "vcc");
            if constexpr (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(a));
            if constexpr (OP == 11) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 12) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(c));
            if constexpr (OP == 13) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(c));
            if constexpr (OP == 14) asm volatile("v_bfe_u32 %0, %0, 1, 4" : "+v"(u[i]));
            if constexpr (OP == 15) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(x[i]) : "v"(c) : "s20", "s21");
        }
    }
    float s = 0;
    unsigned t = 0;
#pragma unroll
    for (int i = 0; i < kUnroll; i++) {
        s += x[i];
        t += u[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)t;
}

// dependent chain: latency of one instruction (single wave per SIMD)
template <int OP>
__global__ void __launch_bounds__(64) klat(float *out, float a, float b, int n_iter) {
    float x = a + threadIdx.x;
    float c = b;
    for (int it = 0; it < n_iter; it++) {
#pragma unroll
        for (int i = 0; i < kUnroll; i++) {
            if constexpr (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
            if constexpr (OP == 2) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(a));
            if constexpr (OP == 3) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(c) : "vcc");
            if constexpr (OP == 7) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
            if constexpr (OP == 20) x = __int_as_float(__builtin_amdgcn_ds_bpermute((threadIdx.x - 1) << 2, __float_as_int(x)));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <int OP>
void run(const char *name, int insts_per_body, float *d_out) {
    int dev = 0;
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, dev));
    const int cus = p.multiProcessorCount;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-28s", name);
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
        const int blocks = cus * wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f, 2.0f, 10);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f, 2.0f, kIters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // wave-instructions per SIMD
        const double winst = (double)wps * kIters * kUnroll * insts_per_body;
        const double ns_per = ms * 1e6 / winst;
        printf("  %dw: %6.3f ns/winst", wps, ns_per);
    }
    printf("\n");
}

template <int OP>
void runlat(const char *name, int insts_per_body, float *d_out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(klat<OP>, dim3(256), dim3(64), 0, 0, d_out, 1.0f, 2.0f, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(klat<OP>, dim3(256), dim3(64), 0, 0, d_out, 1.0f, 2.0f, kIters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s dependent chain: %6.3f ns per instruction\n", name, ms * 1e6 / ((double)kIters * kUnroll * insts_per_body));
}

int main() {
    float *d_out;
    CHECK(hipMalloc(&d_out, 4 * 256 * 4096));
    int clk = 0;
    CHECK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    printf("clock rate attribute %d kHz (ns/winst x GHz = cycles per wave-instruction per SIMD)\n", clk);
    run<0>("v_add_f32", 1, d_out);
    run<10>("v_fma_f32", 1, d_out);
    run<1>("v_min_f32", 1, d_out);
    run<2>("v_min3_f32", 1, d_out);
    run<9>("v_cmp_lt_f32 (vcc)", 1, d_out);
    run<4>("v_cndmask_b32 (vcc)", 1, d_out);
    run<3>("v_cmp+v_cndmask (vcc)", 2, d_out);
    run<15>("v_cmp+v_cndmask (sgpr pair)", 2, d_out);
    run<5>("v_and_b32", 1, d_out);
    run<6>("v_add_u32", 1, d_out);
    run<11>("v_lshl_or_b32", 1, d_out);
    run<14>("v_bfe_u32", 1, d_out);
    run<13>("v_mov_b32", 1, d_out);
    run<7>("v_mov_b32_dpp wave_shr:1", 1, d_out);
    run<12>("v_add_f32_dpp wave_shr:1", 1, d_out);
    run<8>("v_pk_add_f32", 1, d_out);
    runlat<0>("v_add_f32", 1, d_out);
    runlat<2>("v_min3_f32", 1, d_out);
    runlat<3>("v_cmp+v_cndmask", 2, d_out);
    runlat<7>("v_mov_b32_dpp wave_shr:1", 1, d_out);
    runlat<20>("ds_bpermute_b32", 1, d_out);
    return 0;
}
