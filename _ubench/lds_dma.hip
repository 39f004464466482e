// micro-benchmark (development aid): cost of LDS-DMA row batches (global_load_lds_dword, one 256-byte row per instruction)
// per wave: issue time and landing time, by number of issuing waves, batches per wave, and with / without the M0 rewrite.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

template <int LDB, int J>
__device__ __forceinline__ void dma_row(const char* E, int idv, int lane4) {
    const int id = __builtin_amdgcn_readlane(idv, J);
    if (id >= 0) {
        const char* p = E + ((size_t)(unsigned)id << 8) - J * LDB;
        asm volatile("global_load_lds_dword %0, %1 offset:%2" ::"v"(lane4), "s"(p), "n"(J * LDB) : "memory");
    }
}
template <int LDB, int... J>
__device__ __forceinline__ void dma_rows(const char* E, int idv, int lane4, unsigned m0, bool set_m0, std::integer_sequence<int, J...>) {
    if (set_m0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0) : "memory");
    (dma_row<LDB, J>(E, idv, lane4), ...);
}

__global__ __launch_bounds__(768, 1) void k(const char* E, const int* ids, int nrows, int W, int NB, int rewrite, int wide, long long* out) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    long long t_issue = 0, t_land = 0;
    for (int it = 0; it < 20; ++it) {
        __syncthreads();
        long long t0 = 0, t1 = 0;
        // ids first (registers), outside the timed part
        int idv[3];
        for (int b = 0; b < 3; ++b) idv[b] = lane < 12 ? ids[(blockIdx.x * 7 + it * 131 + (wave * 3 + b) * 12 + lane) % nrows] : -1;
        int idw[9];
        for (int b = 0; b < 9; ++b) idw[b] = ids[(blockIdx.x * 7 + it * 131 + (wave * 9 + b) * 4 + (lane >> 4)) % nrows];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        if (wave < W) {
            if (!wide) {
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    if (b < NB) {
                        const int r0 = (wave * NB + b) * 12;
                        dma_rows<272>(E, idv[b], lane * 4, lds0 + r0 * 272, rewrite || b == 0, std::make_integer_sequence<int, 12>{});
                    }
                }
            } else {
                float4 v[9];
#pragma unroll
                for (int b = 0; b < 9; ++b)
                    if (b < NB * 3) v[b] = reinterpret_cast<const float4*>(E + ((size_t)idw[b] << 8))[lane & 15];
                t1 = __builtin_readcyclecounter();
#pragma unroll
                for (int b = 0; b < 9; ++b)
                    if (b < NB * 3) *reinterpret_cast<float4*>(smem + ((wave * NB * 3 + b) * 4 + (lane >> 4)) * 68 + 4 * (lane & 15)) = v[b];
            }
        }
        if (!wide) t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t2 = __builtin_readcyclecounter();
        if (it >= 4) t_issue += t1 - t0, t_land += t2 - t0;
    }
    if (blockIdx.x == 0 && lane == 0) out[wave * 2] = t_issue / 16, out[wave * 2 + 1] = t_land / 16;
}

int main() {
    const int nrows = 100000;
    char* E; int* ids; long long* out;
    hipMalloc(&E, (size_t)nrows * 256); hipMemset(E, 0, (size_t)nrows * 256);
    std::vector<int> h(nrows); srand(1); for (auto& v : h) v = rand() % nrows;
    hipMalloc(&ids, nrows * 4); hipMemcpy(ids, h.data(), nrows * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 24 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    struct { int W, NB, rw, wide; } cfg[] = {{1, 1, 1, 0}, {1, 3, 1, 0}, {1, 3, 0, 0}, {4, 1, 1, 0}, {4, 3, 1, 0}, {4, 3, 0, 0}, {12, 1, 1, 0},
                                             {12, 3, 1, 0}, {12, 3, 0, 0}, {1, 1, 0, 1}, {1, 3, 0, 1}, {4, 3, 0, 1}, {12, 1, 0, 1}, {12, 3, 0, 1}};
    for (auto& c : cfg) {
        long long r[24];
        k<<<256, 768, 150 * 1024>>>(E, ids, nrows, c.W, c.NB, c.rw, c.wide, out);
        hipDeviceSynchronize();
        hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
        printf("%s waves %2d batches/wave %d (rows %3d) m0-rewrite %d : wave0 issue %6lld landed %6lld | last wave issue %6lld landed %6lld\n",
               c.wide ? "b128 loads + ds_write" : "LDS-DMA dword       ", c.W, c.NB, c.W * c.NB * 12, c.rw, r[0], r[1], r[(c.W - 1) * 2], r[(c.W - 1) * 2 + 1]);
    }
    return 0;
}
