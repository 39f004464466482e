// Microbenchmark (development aid, not part of the library): does the L1 path of gfx950 move gathered 256-byte rows
// faster when a wave-instruction covers ONE row with 4-byte lanes (fully coalesced, 256 B per instruction), two rows with
// 8-byte lanes, or four rows with 16-byte lanes (what the gather waves of mvin_fused_split.hip issue)?  Same bytes in
// flight per lane (128 B) in every variant; random rows drawn from a hot set of `span` rows (16 KB: L1-resident,
// 2 MB: L2-resident, 27 MB: the last-fm table), ids preloaded, 16 waves per CU, nothing but the loads and a sum.
// build: hipcc --offload-arch=gfx950 -O3 -o lane_width_probe lane_width_probe.hip ; run: ./lane_width_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int LB> struct Vec;
template <> struct Vec<4> { using T = float; };
template <> struct Vec<8> { using T = float2; };
template <> struct Vec<16> { using T = float4; };
__device__ __forceinline__ float vsum(float v) { return v; }
__device__ __forceinline__ float vsum(float2 v) { return v.x + v.y; }
__device__ __forceinline__ float vsum(float4 v) { return v.x + v.y + v.z + v.w; }

template <int LB>
__global__ __launch_bounds__(64) void probe(const float* table, const int* ids, int iters, float* sink) {
    constexpr int LPR = 256 / LB, RPI = 64 / LPR, NL = 128 / LB;     // lanes per row, rows per instruction, loads per round
    using V = typename Vec<LB>::T;
    const int lane = threadIdx.x, g = lane / LPR, c = lane % LPR;
    const int* my = ids + (size_t)blockIdx.x * iters * 32;             // 32 rows per round in every variant
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int* p = my + it * 32;
        V v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) v[k] = *reinterpret_cast<const V*>(table + (size_t)p[k * RPI + g] * 64 + c * (LB / 4));
#pragma unroll
        for (int k = 0; k < NL; ++k) acc += vsum(v[k]);
    }
    sink[blockIdx.x * 64 + lane] = acc;
}

template <int LB>
static void run(const float* table, const int* ids, int iters, int wpc, float* sink, const char* what) {
    const int grid = 256 * wpc;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<LB><<<grid, 64>>>(table, ids, iters, sink);
    hipEventRecord(e0);
    probe<LB><<<grid, 64>>>(table, ids, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * iters * 32 * 256;
    printf("%-14s %2d-byte lanes (%d row%s per instruction) waves/CU=%2d: %.3f ms  %6.2f TB/s  %5.1f B/clk/CU\n", what, LB, 64 / (256 / LB),
           64 / (256 / LB) > 1 ? "s" : " ", wpc, ms, bytes / ms / 1e9, bytes / 256 / (ms * 1e-3 * 2.4e9));
}

int main() {
    const int nrows = 106389, iters = 512, wpc = 16, grid = 256 * wpc;
    float *table, *sink;
    int* ids;
    hipMalloc(&table, (size_t)nrows * 256);
    hipMemset(table, 0, (size_t)nrows * 256);
    hipMalloc(&sink, (size_t)grid * 64 * 4);
    hipMalloc(&ids, (size_t)grid * iters * 32 * 4);
    std::vector<int> h((size_t)grid * iters * 32);
    const struct { int span; const char* name; } sets[] = {{64, "16 KB hot set"}, {8192, "2 MB hot set"}, {nrows, "27 MB table"}};
    for (auto& s : sets) {
        srand(1);
        for (auto& x : h) x = rand() % s.span;
        hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int w : {8, 16}) {
            run<4>(table, ids, iters, w, sink, s.name);
            run<8>(table, ids, iters, w, sink, s.name);
            run<16>(table, ids, iters, w, sink, s.name);
        }
    }
    return 0;
}
