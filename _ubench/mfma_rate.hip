// micro-benchmark (development aid): issue interval of f32 MFMA shapes, 4 independent chains per wave, 1 wave per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int n) {
    const float a = threadIdx.x * 0.001f, b = 1.f + threadIdx.x * 0.002f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 e0 = {0}, e1 = e0;
    const long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, c3, 0, 0, 0);
        } else if (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, a, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, b, c3, 0, 0, 0);
        } else {
            e0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, e1, 0, 0, 0);
            e0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, e1, 0, 0, 0);
        }
    }
    const long long t1 = wall_clock64();
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + e0[0] + e1[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* out; long long *c, hc; hipMalloc(&out, 1024); hipMalloc(&c, 8);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const int n = 20000;
    const char* names[3] = {"16x16x4 f32 (2048 flop)", "4x4x1 16b f32 (512 flop)", "32x32x2 f32 (4096 flop)"};
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) k<0><<<256, 256>>>(out, c, n); else if (kind == 1) k<1><<<256, 256>>>(out, c, n); else k<2><<<256, 256>>>(out, c, n);
            hipDeviceSynchronize();
        }
        hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        const double sec = (double)hc / (rate * 1e3), per = sec / (4.0 * n);
        printf("%s: %.2f ns per MFMA per SIMD = %.1f cycles at %d MHz ; chip rate %.1f TFLOP/s\n", names[kind], per * 1e9, per * clk * 1e3, clk / 1000,
               (kind == 0 ? 2048 : kind == 1 ? 512 : 4096) / per * 4 * 256 / 1e12);
    }
    return 0;
}
