// TEST INFRASTRUCTURE: the C ABI used from plain C++ -- no Python, no torch in the process.
// Builds a tiny random problem, runs mvin_expand_ids / mvin_rel_score / mvin_gather_attn_l2_fwd
// through include/mvin_hip.h on device buffers it allocated itself, and checks the two outputs against
// a direct loop evaluation of the equations (model.py:267-305, aggregators.py:98-146):
//   ev1[n]  = E[x1[n]].W1 + c1 ,            c_e = q.W_e + b_e
//   agg1[n] = (1/K) sum_k p_k(x1[n]) (E[x2[n,k]].W2 + c2) ,   p = softmax_k(t0[rel])
//   out1[n] = relu((ev1[n] + agg1[n]).A0 + a0)
//   nagg0   = (1/K) sum_n p0[n] ev1[n] ,    nagg1 = (1/K) sum_n p1[n] out1[n]
// Exit code 0 and "C ABI OK" on success.
//   hipcc --offload-arch=gfx950 -Iinclude tests/c_abi/fused_l2_check.cpp -o /tmp/fused_l2_check \
//         -Lmvin_amd -lmvin_hip -Wl,-rpath,$PWD/mvin_amd
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mvin_hip.h"

#define HIP_OK(x)                                                                  \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));          \
            return 2;                                                              \
        }                                                                          \
    } while (0)
#define MVIN_OK(x)                                                                 \
    do {                                                                           \
        int rc_ = (x);                                                             \
        if (rc_ != 0) {                                                            \
            std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, mvin_last_error());   \
            return 3;                                                              \
        }                                                                          \
    } while (0)

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(g_state >> 33);
}
static float frand() { return (float)(rnd() & 0xFFFF) / 65536.f - 0.5f; }

template <class T>
static T* to_dev(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main() {
    const int D = 32, K = 8, nE = 300, nR = 5, B = 37;
    if (mvin_abi_version() != MVIN_ABI_VERSION) {
        std::fprintf(stderr, "ABI version %d != header %d\n", mvin_abi_version(), MVIN_ABI_VERSION);
        return 1;
    }
    if (!mvin_gather_attn_l2_supported(D, K)) return 1;
    std::vector<float> E(nE * D), Rel(nR * D), W1(D * D), W2(D * D), A0(D * D), b1(D), b2(D), a0(D), q(B * D);
    std::vector<float> urh0(3 * D), urh1(3 * D);
    for (auto* v : {&E, &Rel, &W1, &W2, &A0, &b1, &b2, &a0, &q, &urh0, &urh1})
        for (auto& x : *v) x = frand();
    std::vector<int32_t> adj_e(nE * K), adj_r(nE * K);
    for (auto& x : adj_e) x = (int32_t)(rnd() % nE);
    for (auto& x : adj_r) x = (int32_t)(rnd() % nR);
    std::vector<int64_t> items(B);
    for (auto& x : items) x = (int64_t)(rnd() % nE);

    float *dE = to_dev(E), *dRel = to_dev(Rel), *dW1 = to_dev(W1), *dW2 = to_dev(W2), *dA0 = to_dev(A0);
    float *db1 = to_dev(b1), *db2 = to_dev(b2), *da0 = to_dev(a0), *dq = to_dev(q), *du0 = to_dev(urh0),
          *du1 = to_dev(urh1);
    int32_t *dadj_e = to_dev(adj_e), *dadj_r = to_dev(adj_r);
    int64_t* ditems = to_dev(items);
    float *dt0, *dt1, *dn0, *dn1;
    int32_t *dent, *drel;
    HIP_OK(hipMalloc(&dt0, nR * 4));
    HIP_OK(hipMalloc(&dt1, nR * 4));
    HIP_OK(hipMalloc(&dn0, B * D * 4));
    HIP_OK(hipMalloc(&dn1, B * D * 4));
    HIP_OK(hipMalloc(&dent, mvin_ent_elems(B, K, 0) * 4));
    HIP_OK(hipMalloc(&drel, 4));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    // level 0 of the tree (the pairs' items as int32 parents), the two relation-logit tables, the fused kernel
    MVIN_OK(mvin_expand_ids(dadj_e, dadj_r, ditems, nullptr, B, K, 0, nE, dent, drel, st));
    MVIN_OK(mvin_rel_score(dRel, du0, nR, D, dt0, st));
    MVIN_OK(mvin_rel_score(dRel, du1, nR, D, dt1, st));
    MVIN_OK(mvin_gather_attn_l2_fwd(dE, dadj_e, dadj_r, dent, dt0, dt1, dW1, dW2, db1, db2, dq, dA0, da0, B, 1, K, D, nE,
                                    nR, dn0, dn1, nullptr, nullptr, 0, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<float> n0(B * D), n1(B * D);
    HIP_OK(hipMemcpy(n0.data(), dn0, n0.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(n1.data(), dn1, n1.size() * 4, hipMemcpyDeviceToHost));

    // ---- direct evaluation in double precision ----
    auto tscore = [&](const std::vector<float>& u, int r) {
        double s = 0;
        for (int d = 0; d < D; ++d) s += (double)Rel[r * D + d] * u[D + d];   // only the relation slice survives the softmax
        return s;
    };
    auto softmax = [&](const std::vector<float>& u, int x, std::vector<double>& p) {
        double mx = -1e300, z = 0;
        for (int k = 0; k < K; ++k) mx = std::fmax(mx, tscore(u, adj_r[x * K + k]));
        for (int k = 0; k < K; ++k) z += (p[k] = std::exp(tscore(u, adj_r[x * K + k]) - mx));
        for (int k = 0; k < K; ++k) p[k] /= z;
    };
    double worst = 0;
    for (int b = 0; b < B; ++b) {
        std::vector<double> c1(D), c2(D);
        for (int j = 0; j < D; ++j) {
            double s1 = b1[j], s2 = b2[j];
            for (int d = 0; d < D; ++d) {
                s1 += (double)q[b * D + d] * W1[d * D + j];
                s2 += (double)q[b * D + d] * W2[d * D + j];
            }
            c1[j] = s1;
            c2[j] = s2;
        }
        const int x0 = (int)items[b];
        std::vector<double> p0(K), p1(K), r0(D, 0.0), r1(D, 0.0);
        softmax(urh0, x0, p0);
        softmax(urh1, x0, p1);
        for (int n = 0; n < K; ++n) {
            const int x1 = adj_e[x0 * K + n];
            std::vector<double> pk(K), ev1(D), agg(D, 0.0), z(D), o(D);
            softmax(urh0, x1, pk);
            for (int j = 0; j < D; ++j) {
                double s = c1[j];
                for (int d = 0; d < D; ++d) s += (double)E[x1 * D + d] * W1[d * D + j];
                ev1[j] = s;
            }
            for (int k = 0; k < K; ++k) {
                const int x2 = adj_e[x1 * K + k];
                for (int j = 0; j < D; ++j) {
                    double s = c2[j];
                    for (int d = 0; d < D; ++d) s += (double)E[x2 * D + d] * W2[d * D + j];
                    agg[j] += pk[k] * s / K;
                }
            }
            for (int j = 0; j < D; ++j) z[j] = ev1[j] + agg[j];
            for (int j = 0; j < D; ++j) {
                double s = a0[j];
                for (int d = 0; d < D; ++d) s += z[d] * A0[d * D + j];
                o[j] = s > 0 ? s : 0;
                r0[j] += p0[n] * ev1[j] / K;
                r1[j] += p1[n] * o[j] / K;
            }
        }
        for (int j = 0; j < D; ++j) {
            const double e0 = std::fabs(n0[b * D + j] - r0[j]) / (1e-5 * std::fabs(r0[j]) + 1e-6);
            const double e1 = std::fabs(n1[b * D + j] - r1[j]) / (1e-5 * std::fabs(r1[j]) + 1e-6);
            worst = std::fmax(worst, std::fmax(e0, e1));
        }
    }
    // argument errors come back as codes + a message, never as a crash
    if (mvin_gather_attn_l2_fwd(nullptr, dadj_e, dadj_r, dent, dt0, dt1, dW1, dW2, db1, db2, dq, dA0, da0, B, 1, K, D, nE,
                                nR, dn0, dn1, nullptr, nullptr, 0, st) >= 0 ||
        mvin_last_error()[0] == 0) {
        std::fprintf(stderr, "null table was not rejected\n");
        return 4;
    }
    std::printf("worst error / tolerance = %.3f\n", worst);
    if (!(worst <= 1.0)) return 5;
    std::printf("C ABI OK\n");
    return 0;
}
