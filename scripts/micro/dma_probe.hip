// Microbenchmark (development aid, not part of the library): LDS-DMA (global_load_lds_dwordx4) behaviour on
// gfx950 that mvin_keyaddr_stream.hip relies on.
//   1. does the M0 LDS offset reach beyond 64 KB inside one workgroup's allocation?
//   2. pieces (1 KB) per cycle one wave sustains when it gathers random 256-byte rows of an L2-resident
//      table with K pieces in flight, for W waves per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o dma_probe dma_probe.hip ; run: ./dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void dma16(const void* gptr, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_off) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_dma() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__global__ void far_kernel(const float* src, float* out, unsigned off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    for (int i = threadIdx.x; i < 256; i += 64) reinterpret_cast<float*>(smem + off)[i] = -1.f;
    __syncthreads();
    dma16(src + threadIdx.x * 4, lds0 + off);
    wait_dma<0>();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = reinterpret_cast<float*>(smem + off)[i];
}

// four pieces behind ONE M0 write: the instruction offset moves both addresses, so the global pointer is pre-biased
__device__ __forceinline__ void dma16x4(const void* g0, const void* g1, const void* g2, const void* g3, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %2, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %3, off offset:3072" ::"v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(lds_off) : "memory");
}

// each wave: `iters` rounds; per round issue K pieces (4 random rows of 256 B each), wait for the round before
template <int K, bool GROUPED>
__global__ __launch_bounds__(64) void rate_kernel(const float* table, const int* ids, int nrows, int iters, long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int* my = ids + (size_t)blockIdx.x * iters * K * 4;
    float acc = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < K; ++k) dma16(table + (size_t)my[k * 4 + g] * 64 + c * 4, lds0 + k * 1024);
    for (int it = 1; it < iters; ++it) {
        const int* p = my + it * K * 4;
        const int par = it & 1;
        if constexpr (GROUPED) {
#pragma unroll
            for (int k = 0; k < K; k += 4) {
                const char* b0 = reinterpret_cast<const char*>(table + (size_t)p[(k + 0) * 4 + g] * 64 + c * 4);
                const char* b1 = reinterpret_cast<const char*>(table + (size_t)p[(k + 1) * 4 + g] * 64 + c * 4) - 1024;
                const char* b2 = reinterpret_cast<const char*>(table + (size_t)p[(k + 2) * 4 + g] * 64 + c * 4) - 2048;
                const char* b3 = reinterpret_cast<const char*>(table + (size_t)p[(k + 3) * 4 + g] * 64 + c * 4) - 3072;
                dma16x4(b0, b1, b2, b3, lds0 + (par * K + k) * 1024);
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) dma16(table + (size_t)p[k * 4 + g] * 64 + c * 4, lds0 + (par * K + k) * 1024);
        }
        wait_dma<K>();
        const float4* q = reinterpret_cast<const float4*>(smem + (par ^ 1) * K * 1024);
#pragma unroll
        for (int k = 0; k < K; ++k) acc += q[k * 64 + lane].x;
    }
    wait_dma<0>();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = acc;
}

// M0 written ONCE per kernel: all pieces of the wave land inside the 8 KB window the 13-bit instruction offset reaches
// (-4096 .. +3072).  Each round issues 4 pieces into one half of the window and leaves them in flight while the
// other half is read: 4..8 pieces outstanding per wave, no M0 write in the loop.
__global__ __launch_bounds__(64) void fixed_m0_kernel(const float* table, const int* ids, int nrows, int iters, long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int* my = ids + (size_t)blockIdx.x * iters * 16;
    float acc = 0.f;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds0 + 4096) : "memory");
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 2) {
        const int* p = my + it * 16;
        const char* b0 = reinterpret_cast<const char*>(table + (size_t)p[0 + g] * 64 + c * 4);
        const char* b1 = reinterpret_cast<const char*>(table + (size_t)p[4 + g] * 64 + c * 4);
        const char* b2 = reinterpret_cast<const char*>(table + (size_t)p[8 + g] * 64 + c * 4);
        const char* b3 = reinterpret_cast<const char*>(table + (size_t)p[12 + g] * 64 + c * 4);
        asm volatile("global_load_lds_dwordx4 %0, off offset:-4096\n\tglobal_load_lds_dwordx4 %1, off offset:-3072\n\t"
                     "global_load_lds_dwordx4 %2, off offset:-2048\n\tglobal_load_lds_dwordx4 %3, off offset:-1024"
                     ::"v"(b0 + 4096), "v"(b1 + 3072), "v"(b2 + 2048), "v"(b3 + 1024) : "memory");
        wait_dma<4>();
        {
            const float4* q = reinterpret_cast<const float4*>(smem + 4096);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += q[k * 64 + lane].x;
        }
        const char* d0 = reinterpret_cast<const char*>(table + (size_t)p[16 + g] * 64 + c * 4);
        const char* d1 = reinterpret_cast<const char*>(table + (size_t)p[20 + g] * 64 + c * 4);
        const char* d2 = reinterpret_cast<const char*>(table + (size_t)p[24 + g] * 64 + c * 4);
        const char* d3 = reinterpret_cast<const char*>(table + (size_t)p[28 + g] * 64 + c * 4);
        asm volatile("global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %2, off offset:2048\n\tglobal_load_lds_dwordx4 %3, off offset:3072"
                     ::"v"(d0), "v"(d1 - 1024), "v"(d2 - 2048), "v"(d3 - 3072) : "memory");
        wait_dma<4>();
        {
            const float4* q = reinterpret_cast<const float4*>(smem);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += q[k * 64 + lane].x;
        }
    }
    wait_dma<0>();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = acc;
}

// the register path for comparison: 8 independent 16-byte loads per lane per round (4 random 256-byte rows per
// wave-instruction, as the gather waves of mvin_fused_split.hip issue them), summed
template <int LPR>     // lanes per row: 8 = 128-byte rows (D=32 fp32), 16 = 256-byte rows, 32 = 512-byte rows
__global__ __launch_bounds__(64) void reg_kernel(const float* table, const int* ids, int nrows, int iters, long long* cyc, float* sink) {
    const int lane = threadIdx.x, g = lane / LPR, c = lane % LPR;
    const int* my = ids + (size_t)blockIdx.x * iters * 16;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; it += 2) {
        const int* p = my + it * 16;
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(table + (size_t)p[(k * 4 + g) & 31] * (LPR * 4) + c * 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc.x += v[k].x;
            acc.y += v[k].y;
            acc.z += v[k].z;
            acc.w += v[k].w;
        }
    }
    sink[blockIdx.x * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}

template <int LPR>
static void run_reg(const float* table, const int* ids, int nrows, int iters, int waves_per_cu, long long* cyc, float* sink) {
    const int grid = 256 * waves_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    reg_kernel<LPR><<<grid, 64>>>(table, ids, nrows, iters, cyc, sink);
    hipEventRecord(e0);
    reg_kernel<LPR><<<grid, 64>>>(table, ids, nrows, iters, cyc, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * iters * 4 * 1024;
    printf("register loads, %d-byte rows, 8 in flight  waves/CU=%2d: %.3f ms  %.2f TB/s  %.1f B/clk/CU\n", LPR * 16, waves_per_cu, ms, bytes / ms / 1e9,
           bytes / 256 / (ms * 1e-3 * 2.4e9));
}

static void run_fixed(const float* table, const int* ids, int nrows, int iters, int waves_per_cu, long long* cyc, float* sink) {
    const int grid = 256 * waves_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    fixed_m0_kernel<<<grid, 64, 8192>>>(table, ids, nrows, iters, cyc, sink);
    hipEventRecord(e0);
    fixed_m0_kernel<<<grid, 64, 8192>>>(table, ids, nrows, iters, cyc, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * iters * 4 * 1024;
    printf("fixed M0, 4+4 pieces  waves/CU=%2d: %.3f ms  %.2f TB/s  %.1f B/clk/CU  (%.1f cycles per piece per wave)\n", waves_per_cu, ms,
           bytes / ms / 1e9, bytes / 256 / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / (iters * 4));
}

template <int K, bool GROUPED>
static void run_rate(const float* table, const int* ids, int nrows, int iters, int waves_per_cu, long long* cyc, float* sink) {
    const int grid = 256 * waves_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = 2 * K * 1024;
    rate_kernel<K, GROUPED><<<grid, 64, lds>>>(table, ids, nrows, iters, cyc, sink);
    hipEventRecord(e0);
    rate_kernel<K, GROUPED><<<grid, 64, lds>>>(table, ids, nrows, iters, cyc, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= grid;
    const double bytes = (double)grid * iters * K * 1024;
    printf("%s K=%2d waves/CU=%2d: %.3f ms  %.2f TB/s  %.1f B/clk/CU (memtime %.0f ticks/wave, %.1f ticks per piece per wave)\n", GROUPED ? "one-M0-per-4" : "M0-per-piece ", K,
           waves_per_cu, ms, bytes / ms / 1e9, bytes / 256 / (ms * 1e-3 * 2.4e9), avg, avg / (iters * K));
}

int main() {
    const int nrows = 106389;   // last-fm-sized table, 27 MB: L2 / MALL resident
    float* table;
    hipMalloc(&table, (size_t)nrows * 256);
    hipMemset(table, 0, (size_t)nrows * 256);
    float* out;
    hipMalloc(&out, 1024);
    float* src;
    hipMalloc(&src, 1024);
    std::vector<float> hs(256);
    for (int i = 0; i < 256; ++i) hs[i] = (float)i;
    hipMemcpy(src, hs.data(), 1024, hipMemcpyHostToDevice);
    for (unsigned off : {0u, 32768u, 65536u, 98304u, 131072u, 155648u}) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(far_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        far_kernel<<<1, 64, off + 1024>>>(src, out, off);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> ho(256);
        hipMemcpy(ho.data(), out, 1024, hipMemcpyDeviceToHost);
        int ok = 1;
        for (int i = 0; i < 256; ++i) ok &= ho[i] == (float)i;
        printf("LDS-DMA to offset %6u: %s (err %d, out[0]=%g out[255]=%g)\n", off, ok ? "ok" : "WRONG", (int)e, ho[0], ho[255]);
    }
    const int iters = 512, maxgrid = 256 * 16;
    std::vector<int> hid((size_t)maxgrid * iters * 16 * 4);
    srand(1);
    for (auto& v : hid) v = rand() % nrows;
    int* ids;
    hipMalloc(&ids, hid.size() * 4);
    hipMemcpy(ids, hid.data(), hid.size() * 4, hipMemcpyHostToDevice);
    long long* cyc;
    hipMalloc(&cyc, maxgrid * 8);
    float* sink;
    hipMalloc(&sink, maxgrid * 64 * 4);
    for (int w : {1, 2, 4, 8}) {
        run_rate<4, false>(table, ids, nrows, iters, w, cyc, sink);
        run_rate<4, true>(table, ids, nrows, iters, w, cyc, sink);
        run_rate<8, true>(table, ids, nrows, iters, w, cyc, sink);
        if (w <= 4) run_rate<16, true>(table, ids, nrows, iters, w, cyc, sink);
    }
    for (int w : {1, 2, 4, 8, 12, 16}) run_fixed(table, ids, nrows, iters, w, cyc, sink);
    for (int w : {4, 8, 16}) run_reg<16>(table, ids, nrows, iters, w, cyc, sink);
    // the same with the row ids folded into a small range: L1-resident (64 rows = 16 KB) and L2-resident (8192 rows = 2 MB)
    for (int range : {64, 8192}) {
        std::vector<int> h2(hid.size());
        for (size_t i = 0; i < hid.size(); ++i) h2[i] = hid[i] % range;
        hipMemcpy(ids, h2.data(), h2.size() * 4, hipMemcpyHostToDevice);
        printf("-- rows drawn from the first %d rows (%d KB)\n", range, range / 4);
        for (int w : {2, 4, 8, 16}) run_fixed(table, ids, nrows, iters, w, cyc, sink);
        for (int w : {4, 8, 16}) run_reg<16>(table, ids, nrows, iters, w, cyc, sink);
        run_reg<8>(table, ids, nrows, iters, 16, cyc, sink);
        run_reg<32>(table, ids, nrows, iters, 16, cyc, sink);
    }
    return 0;
}
