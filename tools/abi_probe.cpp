// Torch-free driver of the C ABI (include/ance_amd.h): proves the library is usable from plain C/C++
// and gives rocprofv3 (--kernel-trace / --pmc) a small process to profile.
//
//   abi_probe search  <n_rows> <n_queries> <k> <reps>
//   abi_probe encode  <n_passages> <seq_len> <layers> <reps> [max_tokens]
//   abi_probe gemm    <ablate> <epi> <M> <N> <K> <reps>
//
// Build: hipcc -O2 -std=c++17 tools/abi_probe.cpp -Iinclude -Lance_amd -lance_amd -Wl,-rpath,'$ORIGIN/../ance_amd' -o tools/abi_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <random>
#include <vector>

#include "ance_amd.h"

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(2);                                                               \
        }                                                                          \
    } while (0)
#define AK(x)                                                              \
    do {                                                                   \
        int r_ = (x);                                                      \
        if (r_ != 0) {                                                     \
            fprintf(stderr, "%s:%d rc=%d %s\n", __FILE__, __LINE__, r_, ance_last_error()); \
            exit(3);                                                       \
        }                                                                  \
    } while (0)

static void ln_rows(std::vector<float> &v, size_t n, int d, uint64_t seed) {
    std::mt19937_64 g(seed);
    std::normal_distribution<float> nd(0.f, 1.f);
    v.resize(n * d);
    for (size_t r = 0; r < n; ++r) {
        double s = 0, q = 0;
        float *p = &v[r * d];
        for (int j = 0; j < d; ++j) { p[j] = nd(g); s += p[j]; }
        const float m = (float)(s / d);
        for (int j = 0; j < d; ++j) { p[j] -= m; q += (double)p[j] * p[j]; }
        const float rs = 1.0f / sqrtf((float)(q / d) + 1e-5f);
        for (int j = 0; j < d; ++j) p[j] *= rs;
    }
}

static int run_search(int64_t n, int64_t nq, int k, int reps) {
    const int d = 768;
    // corpus: a 65,536-row random block tiled to n rows (fast to build, same arithmetic intensity)
    std::vector<float> blk, q;
    const size_t nb = n < 65536 ? (size_t)n : 65536;
    ln_rows(blk, nb, d, 1);
    ln_rows(q, (size_t)nq, d, 2);
    float *dx, *dq, *dD;
    int64_t *dI;
    void *ws;
    CK(hipMalloc(&dx, (size_t)n * d * 4));
    for (int64_t r0 = 0; r0 < n; r0 += (int64_t)nb) {
        const size_t rows = (size_t)((n - r0) < (int64_t)nb ? (n - r0) : (int64_t)nb);
        CK(hipMemcpy(dx + (size_t)r0 * d, blk.data(), rows * d * 4, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&dq, (size_t)nq * d * 4));
    CK(hipMemcpy(dq, q.data(), (size_t)nq * d * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dD, (size_t)nq * k * 4));
    CK(hipMalloc(&dI, (size_t)nq * k * 8));
    const size_t wsb = ance_ip_topk_workspace_bytes(n, nq, d, k);
    CK(hipMalloc(&ws, wsb));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    AK(ance_ip_topk(dx, n, 0, dq, nq, d, k, dD, dI, ws, wsb, st));
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) AK(ance_ip_topk(dx, n, 0, dq, nq, d, k, dD, dI, ws, wsb, st));
    CK(hipStreamSynchronize(st));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    std::vector<int64_t> I((size_t)k);
    CK(hipMemcpy(I.data(), dI, (size_t)k * 8, hipMemcpyDeviceToHost));
    printf("{\"probe\":\"search\",\"n\":%lld,\"nq\":%lld,\"k\":%d,\"sec_per_call\":%.6f,\"queries_per_sec\":%.1f,"
           "\"tflops\":%.2f,\"first_ids\":[%lld,%lld,%lld]}\n",
           (long long)n, (long long)nq, k, dt, nq / dt, 2.0 * n * nq * d / dt / 1e12, (long long)I[0], (long long)I[1],
           (long long)I[2]);
    return 0;
}

static int run_encode(int64_t n, int L, int layers, int reps, int max_tokens) {
    AnceEncoderDesc D;
    memset(&D, 0, sizeof(D));
    D.arch = ANCE_ARCH_ROBERTA; D.n_layers = layers; D.hidden = 768; D.n_heads = 12; D.intermediate = 3072;
    D.vocab_size = 50265; D.max_position = 514; D.pad_token_id = 1; D.ln_eps = 1e-5f; D.has_head = 1;
    D.max_seq_len = L > 512 ? 512 : L; D.max_tokens = max_tokens;
    const int nw = ANCE_ENCODER_N_WEIGHTS(layers, 1);
    std::mt19937_64 g(7);
    std::normal_distribution<float> nd(0.f, 0.02f);
    auto dev_fill = [&](size_t cnt, int kind) {  // kind 0: normal(0,.02); 1: ones; 2: zeros
        std::vector<float> h(cnt);
        for (size_t i = 0; i < cnt; ++i) h[i] = kind == 0 ? nd(g) : (kind == 1 ? 1.f : 0.f);
        float *p;
        CK(hipMalloc(&p, cnt * 4));
        CK(hipMemcpy(p, h.data(), cnt * 4, hipMemcpyHostToDevice));
        return (const void *)p;
    };
    std::vector<const void *> w;
    const size_t H = 768, I = 3072;
    w.push_back(dev_fill((size_t)D.vocab_size * H, 0));
    w.push_back(dev_fill((size_t)D.max_position * H, 0));
    w.push_back(dev_fill(H, 0));
    w.push_back(dev_fill(H, 1));
    w.push_back(dev_fill(H, 2));
    for (int i = 0; i < layers; ++i) {
        for (int j = 0; j < 4; ++j) { w.push_back(dev_fill(H * H, 0)); w.push_back(dev_fill(H, 2)); }  // q k v o
        w.push_back(dev_fill(H, 1)); w.push_back(dev_fill(H, 2));
        w.push_back(dev_fill(I * H, 0)); w.push_back(dev_fill(I, 2));
        w.push_back(dev_fill(H * I, 0)); w.push_back(dev_fill(H, 2));
        w.push_back(dev_fill(H, 1)); w.push_back(dev_fill(H, 2));
    }
    w.push_back(dev_fill(H * H, 0)); w.push_back(dev_fill(H, 2)); w.push_back(dev_fill(H, 1)); w.push_back(dev_fill(H, 2));
    if ((int)w.size() != nw) { fprintf(stderr, "weight count %zu != %d\n", w.size(), nw); return 4; }
    const size_t wb = ance_encoder_weight_bytes(&D), xb = ance_encoder_workspace_bytes(&D);
    void *arena, *ws;
    CK(hipMalloc(&arena, wb));
    CK(hipMalloc(&ws, xb));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    AnceEncoder *enc = nullptr;
    AK(ance_encoder_create(&D, w.data(), nw, arena, wb, ws, xb, st, &enc));
    CK(hipStreamSynchronize(st));
    // synthetic records: lengths lognormal(ln 70, .45) clipped to [8, L]
    std::lognormal_distribution<double> ld(log(70.0), 0.45);
    std::uniform_int_distribution<int> tok(3, 50264);
    std::vector<int32_t> rec((size_t)n * (L + 1)), lens((size_t)n);
    double tokens = 0, flops = 0;
    for (int64_t r = 0; r < n; ++r) {
        int len = (int)lrint(ld(g));
        len = len < 8 ? 8 : (len > L ? L : len);
        lens[r] = len;
        tokens += len;
        flops += ance_encoder_flops_per_sequence(len);
        int32_t *p = &rec[(size_t)r * (L + 1)];
        p[0] = (int32_t)__builtin_bswap32((uint32_t)len);
        for (int j = 0; j < L; ++j) p[1 + j] = j < len ? tok(g) : 1;
        p[1] = 0;
        p[len] = 2;
    }
    int32_t *drec;
    float *dout;
    CK(hipMalloc(&drec, rec.size() * 4));
    CK(hipMemcpy(drec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, (size_t)n * 768 * 4));
    AK(ance_encode_records(enc, drec, lens.data(), n, L, 1, dout, st));
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) AK(ance_encode_records(enc, drec, lens.data(), n, L, 1, dout, st));
    CK(hipStreamSynchronize(st));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    std::vector<float> o(4);
    CK(hipMemcpy(o.data(), dout, 16, hipMemcpyDeviceToHost));
    printf("{\"probe\":\"encode\",\"n\":%lld,\"L\":%d,\"layers\":%d,\"sec_per_call\":%.6f,\"passages_per_sec\":%.1f,"
           "\"mean_len\":%.2f,\"algorithmic_tflops\":%.1f,\"out0\":[%.5f,%.5f,%.5f,%.5f]}\n",
           (long long)n, L, layers, dt, n / dt, tokens / n, flops / dt / 1e12, o[0], o[1], o[2], o[3]);
    ance_encoder_destroy(enc);
    return 0;
}

static int run_gemm(int variant, int epi, int M, int N, int K, int reps) {
    std::mt19937_64 g(3);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<_Float16> a((size_t)M * K), b((size_t)N * K);
    // fill a 1M-element random block and tile it (host RNG is slow)
    std::vector<_Float16> blk(1 << 20);
    for (auto &v : blk) v = (_Float16)(nd(g) * 0.5f);
    for (size_t i = 0; i < a.size(); ++i) a[i] = blk[(i * 2654435761ull) & ((1 << 20) - 1)];
    for (size_t i = 0; i < b.size(); ++i) b[i] = blk[(i * 40503ull + 17) & ((1 << 20) - 1)];
    _Float16 *da, *db;
    float *dbias, *dres = nullptr;
    void *dout;
    CK(hipMalloc(&da, a.size() * 2)); CK(hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&db, b.size() * 2)); CK(hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&dbias, (size_t)N * 4)); CK(hipMemset(dbias, 0, (size_t)N * 4));
    CK(hipMalloc(&dout, (size_t)M * N * 4));
    if (epi == 2 || (variant & 32)) { CK(hipMalloc(&dres, (size_t)M * N * 4)); CK(hipMemset(dres, 0, (size_t)M * N * 4)); }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    AK(ance_debug_gemm(variant, epi, da, db, M, N, K, dbias, dout, dres, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) AK(ance_debug_gemm(variant, epi, da, db, M, N, K, dbias, dout, dres, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps;
    printf("{\"probe\":\"gemm\",\"ablate\":%d,\"epi\":%d,\"M\":%d,\"N\":%d,\"K\":%d,\"us\":%.1f,\"tflops\":%.1f}\n", variant, epi, M,
           N, K, us, 2.0 * M * N * K / us / 1e6);
    if ((variant & 32) && epi != 2) {  // timeline mode: per-workgroup phase stamps (100 MHz)
        const int nb = (M / 256) * (N / 256);
        std::vector<unsigned long long> ts((size_t)nb * 5);
        CK(hipMemcpy(ts.data(), dres, ts.size() * 8, hipMemcpyDeviceToHost));
        double ph[4] = {0, 0, 0, 0};
        unsigned long long tmin = ~0ull, tmax = 0;
        int cnt = 0;
        for (int i = 0; i < nb; ++i) {
            const unsigned long long *t = &ts[(size_t)i * 5];
            if (!t[0]) continue;
            for (int j = 0; j < 4; ++j) ph[j] += (double)(t[j + 1] - t[j]);
            if (t[0] < tmin) tmin = t[0];
            if (t[4] > tmax) tmax = t[4];
            ++cnt;
        }
        printf("{\"probe\":\"gemm_timeline\",\"workgroups\":%d,\"prologue_us\":%.2f,\"main_loop_us\":%.2f,\"epilogue_issue_us\":%.2f,"
               "\"store_drain_us\":%.2f,\"kernel_span_us\":%.1f}\n", cnt, ph[0] / cnt / 100.0, ph[1] / cnt / 100.0, ph[2] / cnt / 100.0,
               ph[3] / cnt / 100.0, (double)(tmax - tmin) / 100.0);
    }
    hipFree(da); hipFree(db); hipFree(dbias); hipFree(dout); if (dres) hipFree(dres);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 8 && !strcmp(argv[1], "gemm"))
        return run_gemm(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]));
    if (argc >= 6 && !strcmp(argv[1], "search")) return run_search(atoll(argv[2]), atoll(argv[3]), atoi(argv[4]), atoi(argv[5]));
    if (argc >= 6 && !strcmp(argv[1], "encode"))
        return run_encode(atoll(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), argc > 6 ? atoi(argv[6]) : 65536);
    fprintf(stderr, "usage: abi_probe search n nq k reps | encode n L layers reps [max_tokens] | gemm ablate epi M N K reps\n");
    return 1;
}
