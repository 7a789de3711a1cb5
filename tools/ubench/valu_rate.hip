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
            // integer / packed 16-bit candidates for a fixed-point DP (round 3)
            if constexpr (OP == 30) asm volatile("v_min_i32 %0, %0, %1" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 31) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 32) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(c), "v"(a));
            if constexpr (OP == 33) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 34) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 35) asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(u[i]) : "v"(c), "v"(a));
            if constexpr (OP == 36) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 37) asm volatile("v_pk_ashrrev_i16 %0, 15, %0" : "+v"(u[i]));
            if constexpr (OP == 38) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(c), "v"(a));
            if constexpr (OP == 39) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(c), "v"(a));
            if constexpr (OP == 40) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 41) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(c), "v"(a));
            if constexpr (OP == 42) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(c), "v"(a));
            if constexpr (OP == 43) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 44) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u[i]));
            if constexpr (OP == 45) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 46) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(c) : "vcc");
            if constexpr (OP == 47) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 48) asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 49) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 50) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 51) asm volatile("v_min_i16 %0, %0, %1" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 52) asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 53) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]));
            if constexpr (OP == 54) asm volatile("v_min_i32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(c));
            if constexpr (OP == 55) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(u[i]) : "v"(c));
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
    run<30>("v_min_i32", 1, d_out);
    run<31>("v_min_u32", 1, d_out);
    run<48>("v_max_i32", 1, d_out);
    run<32>("v_min3_i32", 1, d_out);
    run<51>("v_min_i16", 1, d_out);
    run<46>("v_cmp_lt_i32+v_cndmask", 2, d_out);
    run<33>("v_pk_add_i16", 1, d_out);
    run<52>("v_pk_add_i16 clamp", 1, d_out);
    run<36>("v_pk_sub_i16", 1, d_out);
    run<34>("v_pk_min_i16", 1, d_out);
    run<47>("v_pk_min_u16", 1, d_out);
    run<45>("v_pk_max_i16", 1, d_out);
    run<35>("v_pk_mad_i16", 1, d_out);
    run<55>("v_pk_mul_lo_u16", 1, d_out);
    run<37>("v_pk_ashrrev_i16", 1, d_out);
    run<38>("v_bfi_b32", 1, d_out);
    run<39>("v_perm_b32", 1, d_out);
    run<40>("v_alignbit_b32", 1, d_out);
    run<41>("v_and_or_b32", 1, d_out);
    run<42>("v_add3_u32", 1, d_out);
    run<43>("v_sub_u32", 1, d_out);
    run<44>("v_lshlrev_b32", 1, d_out);
    run<49>("v_or_b32", 1, d_out);
    run<50>("v_xor_b32", 1, d_out);
    run<53>("v_mov_b32_dpp row_shr:1", 1, d_out);
    run<54>("v_min_i32_dpp row_shr:1", 1, d_out);
    runlat<0>("v_add_f32", 1, d_out);
    runlat<2>("v_min3_f32", 1, d_out);
    runlat<3>("v_cmp+v_cndmask", 2, d_out);
    runlat<7>("v_mov_b32_dpp wave_shr:1", 1, d_out);
    runlat<20>("ds_bpermute_b32", 1, d_out);
    return 0;
}
