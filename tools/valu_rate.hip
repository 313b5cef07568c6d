// Microbenchmark: sustained issue rate of a few VALU instructions on gfx950, per SIMD, as a
// function of waves per SIMD.  Prints wave-instructions per nanosecond per SIMD and, with
// s_memtime, shader cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kIters = 2000;
constexpr int kUnroll = 16;

template <int OP>
__global__ void rate_kernel(float *out, unsigned long long *cyc, float seed)
{
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
    float2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = make_float2(seed + i, seed - i);
    const float b = seed * 0.5f, c = seed * 0.25f;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll / 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
                if (OP == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
                if (OP == 4) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                if (OP == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (OP == 6) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(b), "v"(c));
                if (OP == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (OP == 10) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (OP == 11) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a[i]), "v"(b) : "s20", "s21");
                if (OP == 12) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 13) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b));
                if (OP == 14) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(a[i]) : "v"(b), "v"(c) : "s20", "s21");
                if (OP == 15) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 16) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 17) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 18) asm volatile("v_mul_f32 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b));
                if (OP == 19) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b), "v"(c));
                if (OP == 20) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 21) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
                if (OP == 22) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 23) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
                if (OP == 24) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
                if (OP == 25) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 26) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 27) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 28) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 29) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 30) asm volatile("v_fma_f32 %0, -%0, |%1|, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 31) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cmp_lt_f32 s[22:23], %0, %2\n\ts_and_b64 s[20:21], s[20:21], s[22:23]\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(a[i]) : "v"(b), "v"(c) : "s20", "s21", "s22", "s23");
                if (OP == 32) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
                if (OP == 33) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
                if (OP == 34) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 35) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 4\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int OP>
int run(const char *name, int waves_per_simd)
{
    const int cus = 256;
    const int threads = 256;                      // 4 waves = one per SIMD
    const int blocks = cus * waves_per_simd;
    float *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, sizeof(float) * blocks * threads));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 2));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 2);
    CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 2, hipMemcpyDeviceToHost));
    double avg = 0, real = 0; for (int i = 0; i < blocks; ++i) { avg += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    avg /= blocks; real /= blocks;
    const double inst_per_wave = (double)kIters * kUnroll;
    const double inst_per_simd = inst_per_wave * waves_per_simd;
    printf("%-18s waves/SIMD %d : %7.3f ms  wave-lifetime %9.0f cyc  => %5.2f cyc/inst/SIMD (s_memtime), %6.3f inst/ns/SIMD, memtime clock %5.0f MHz\n",
           name, waves_per_simd, ms, avg, avg / inst_per_simd, inst_per_simd / (ms * 1e6), avg / real * 100.0);
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
    return 0;
}

int main()
{
    run<0>("v_fma_f32", 8);
    const int w = 8;
    run<2>("v_mul_f32", w);
    run<9>("v_pk_mul_f32", w);
    run<10>("v_pk_add_f32", w);
    run<8>("v_fma_f32 (sgpr)", w);
    run<3>("v_cndmask_b32", w);
    run<4>("v_cmp_lt_f32 vcc", w);
    run<11>("v_cmp_lt_f32 sgpr", w);
    run<6>("v_min3_f32", w);
    run<5>("v_exp_f32", w);
    run<7>("v_rcp_f32", w);
    run<12>("v_mul_lo_u32", w);
    run<13>("v_cndmask_e64 sgpr", w);
    run<14>("cmp+cndmask pair", w);
    run<19>("v_cndmask vcc indep", w);
    run<15>("v_med3_f32", w);
    run<16>("v_max_f32", w);
    run<17>("v_add_f32", w);
    run<18>("v_mul_f32 clamp", w);
    run<20>("v_fmac_f32", w);
    run<21>("v_cvt_f32_i32", w);
    run<22>("v_and_b32", w);
    run<23>("cmp vcc+cndmask e32", w);
    run<32>("cmp vcc+cndmask e64", w);
    run<35>("cmp,nop4,cndmask e32", w);
    run<24>("v_cndmask_e64 vcc", w);
    run<31>("2cmp+s_and+cndmask", w);
    run<25>("v_min_f32", w);
    run<26>("v_sub_f32", w);
    run<27>("v_mad_u32_u24", w);
    run<28>("v_bfi_b32", w);
    run<29>("v_add_u32", w);
    run<30>("v_fma_f32 neg/abs", w);
    run<33>("v_mov_b32", w);
    run<34>("v_mul_legacy_f32", w);
    return 0;
}
