// layout check of v_mfma_f32_4x4x1_16b_f32 (development aid): D_b[i][j] += A_b[i] * B_b[j], 16 blocks b
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D, long long* cyc) {
    const int lane = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[lane], B[lane], acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[v * 64 + lane] = acc[v];
    // broadcast: cbsz = 4 (one block's A for all 16 blocks), abid = 5 (block 5 supplies it)
    f32x4 bc = {0.f, 0.f, 0.f, 0.f};
    bc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[lane], B[lane], bc, 4, 5, 0);
    for (int v = 0; v < 4; ++v) D[448 + v * 64 + lane] = bc[v];
    // timing: 64 dependent vs 2 x 32 interleaved
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    float a = A[lane], b = B[lane];
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 64; ++i) c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    D[256 + lane] = c0[0];
    long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, c1, 0, 0, 0);
    }
    D[320 + lane] = c0[0] + c1[0];
    long long t2 = __builtin_readcyclecounter();
    f32x4 d0 = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
    D[384 + lane] = d0[0];
    long long t3 = __builtin_readcyclecounter();
    if (lane == 0) cyc[0] = t1 - t0, cyc[1] = t2 - t1, cyc[2] = t3 - t2;
}
int main() {
    float hA[64], hB[64], hD[704], *A, *B, *D; long long *c, hc[3];
    for (int l = 0; l < 64; ++l) hA[l] = 1.f + l, hB[l] = 100.f * (1 + l);
    hipMalloc(&A, 256); hipMalloc(&B, 256); hipMalloc(&D, sizeof(hD)); hipMalloc(&c, 24);
    hipMemcpy(A, hA, 256, hipMemcpyHostToDevice); hipMemcpy(B, hB, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(A, B, D, c); hipDeviceSynchronize();
    hipMemcpy(hD, D, sizeof(hD), hipMemcpyDeviceToHost); hipMemcpy(hc, c, 24, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int v = 0; v < 4; ++v) for (int l = 0; l < 64; ++l) {
        const int b = l / 4, j = l % 4;
        const float want = hA[4 * b + v] * hB[4 * b + j];      // D[vgpr v][lane 4b+j] = A_b[v] * B_b[j]
        if (hD[v * 64 + l] != want) { if (bad < 5) printf("v %d lane %d got %g want %g\n", v, l, hD[v * 64 + l], want); ++bad; }
    }
    int bad2 = 0;
    for (int v = 0; v < 4; ++v) for (int l = 0; l < 64; ++l) {
        const int b = l / 4, j = l % 4;
        const float want = hA[4 * 5 + v] * hB[4 * b + j];      // A from block 5 for every block
        if (hD[448 + v * 64 + l] != want) { if (bad2 < 5) printf("bcast v %d lane %d got %g want %g\n", v, l, hD[448 + v * 64 + l], want); ++bad2; }
    }
    printf("cbsz=4 abid=5: D[v][4b+j] = A[4*5+v]*B[4b+j]: %s\n", bad2 ? "NO" : "yes");
    printf("layout D[v][4b+j] = A[4b+v]*B[4b+j]: %s ; cycles: 64 dependent 4x4x1 %lld, 2x32 interleaved %lld, 16 dependent 16x16x4 %lld\n", bad ? "NO" : "yes", hc[0], hc[1], hc[2]);
    return 0;
}
