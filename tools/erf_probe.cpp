// Diagnostic (not product code): accuracy of the device erff / the exact-GELU expression of the split and fp32 epilogues,
// against double precision on the host.  hipcc --offload-arch=gfx950 tools/erf_probe.cpp -o tools/erf_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float *x, float *e, float *g, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        e[i] = erff(x[i]);
        g[i] = 0.5f * x[i] * (1.0f + erff(x[i] * 0.70710678118654752440f));
    }
}
int main() {
    const int n = 12000001;
    std::vector<float> x(n), e(n), g(n);
    for (int i = 0; i < n; ++i) x[i] = -6.0f + 1e-6f * i;
    float *dx, *de, *dg;
    hipMalloc(&dx, n * 4); hipMalloc(&de, n * 4); hipMalloc(&dg, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, de, dg, n);
    hipMemcpy(e.data(), de, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(g.data(), dg, n * 4, hipMemcpyDeviceToHost);
    double me = 0, mg = 0, xe = 0, xg = 0; long bad = 0;
    for (int i = 0; i < n; ++i) {
        double de_ = fabs((double)e[i] - erf((double)x[i]));
        double dg_ = fabs((double)g[i] - 0.5 * (double)x[i] * (1.0 + erf((double)x[i] * 0.70710678118654752440)));
        if (de_ > me) { me = de_; xe = x[i]; }
        if (dg_ > mg) { mg = dg_; xg = x[i]; }
        if (dg_ > 1e-5) ++bad;
    }
    printf("device erff: max abs err %.3e at x=%.6f; gelu expr: max abs err %.3e at x=%.6f; >1e-5: %ld\n", me, xe, mg, xg, bad);
    return 0;
}
