// Yardstick only (not part of the product): rocBLAS/Tensile fp16 NT GEMM on the same shapes and data
// distribution as tools/abi_probe's gemm mode, so that rocprofv3 can compare cycles and clocks.
//   rocblas_probe M N K reps
// Build: hipcc -O2 tools/rocblas_probe.cpp -lrocblas -o tools/rocblas_probe
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdio.h>
#include <stdlib.h>
#include <random>
#include <vector>
#define CK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "hip error line %d\n", __LINE__); exit(2); } } while (0)
int main(int argc, char **argv) {
    if (argc < 5) return 1;
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), reps = atoi(argv[4]);
    std::mt19937 g(1);
    std::normal_distribution<float> nd(0.f, 0.5f);
    std::vector<_Float16> a((size_t)M * K), b((size_t)N * K);
    for (auto &v : a) v = (_Float16)nd(g);
    for (auto &v : b) v = (_Float16)nd(g);
    _Float16 *da, *db, *dc;
    CK(hipMalloc(&da, a.size() * 2)); CK(hipMalloc(&db, b.size() * 2)); CK(hipMalloc(&dc, (size_t)M * N * 2));
    CK(hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice));
    rocblas_handle h;
    rocblas_create_handle(&h);
    const float alpha = 1.f, beta = 0.f;
    // row-major C[M,N] = A[M,K] . B[N,K]^T  ==  column-major C^T[N,M] = B^T-as-colmajor[K,N]^T . A-as-colmajor[K,M]
    auto run = [&]() {
        return rocblas_gemm_ex(h, rocblas_operation_transpose, rocblas_operation_none, N, M, K, &alpha, db, rocblas_datatype_f16_r, K,
                               da, rocblas_datatype_f16_r, K, &beta, dc, rocblas_datatype_f16_r, N, dc, rocblas_datatype_f16_r, N,
                               rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
    };
    for (int i = 0; i < 2; ++i) if (run() != rocblas_status_success) { fprintf(stderr, "rocblas failed\n"); return 3; }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) run();
    hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("{\"probe\":\"rocblas\",\"M\":%d,\"N\":%d,\"K\":%d,\"us\":%.1f,\"tflops\":%.1f}\n", M, N, K, ms * 1e3 / reps, 2.0 * M * N * K / (ms / reps * 1e-3) / 1e12);
    return 0;
}
