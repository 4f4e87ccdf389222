// What the board can sustain: the fp16 MFMA rate (v_mfma_f32_32x32x16_f16) of MI355X AT ITS POWER CAP, with and without the data
// movement of a GEMM main loop around it.  Measurement tool; not part of the library.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/power_probe tools/power_probe.cpp      run: tools/power_probe [seconds per case]
// Every case runs workgroups of 512 threads, one per CU (8 waves per CU with up to 256 registers each, as the split GEMM), in a loop of launches for the given time while
// a host thread samples the amdgpu hwmon files (socket power, shader clock); prints TF executed, mean power, mean clock per case:
//   mfma          24 MFMAs per step on 8 accumulators (the split GEMM's 2 x 4 sub-tiles x 3 products), operands held in registers
//   mfma_small    the same with one operand of every second and third MFMA small in magnitude (lo halves of a pair: |x| <= 2^-11)
//   mfma_zero     the same with all-zero operands (what the matrix pipe costs without data toggling)
//   mfma+lds      + the 12 ds_read_b128 per 24 MFMAs of the split main loop, feeding the MFMAs
//   mfma+lds+dma  + 8 LDS-DMA pieces (buffer_load ... lds, 1 KiB per wave) per 48 MFMAs from a 64 MiB buffer (L2 / Infinity Cache)
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <glob.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>  // 0 registers only, 1 + LDS reads, 2 + LDS reads + DMA
__global__ void __launch_bounds__(512, 1) probe_kernel(const _Float16 *src, float *sink, int iters, int src_halves) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    f16x8 a[4], b[8];  // A: (x, hi/lo), B: (y, hi/lo)
    for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const f16x8 *>(src + ((size_t)(blockIdx.x * 512 + tid) * 12 + j) * 8 % src_halves);
    for (int j = 0; j < 8; ++j) b[j] = *reinterpret_cast<const f16x8 *>(src + ((size_t)(blockIdx.x * 512 + tid) * 12 + 4 + j) * 8 % src_halves);
    if (MODE >= 1) {
        for (int e = tid; e < 64 * 1024 / 16; e += 512) *reinterpret_cast<f16x8 *>(lds + e * 8) = *reinterpret_cast<const f16x8 *>(src + ((size_t)e * 8) % src_halves);
        __syncthreads();
    }
    f32x16 acc[2][4];
    for (int x = 0; x < 2; ++x)
        for (int y = 0; y < 4; ++y) acc[x][y] = f32x16{0};
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(src), 0, src_halves * 2, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) {  // 12 fragment reads per 24 MFMAs, addresses moving through the 64 KiB image
            const int base = ((it * 97 + w * 13) & 31) * 1024 + l * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const f16x8 *>(lds + ((base + j * 512) & (32 * 1024 - 1)));
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = *reinterpret_cast<const f16x8 *>(lds + ((base + 2048 + j * 512) & (32 * 1024 - 1)));
        }
        if (MODE >= 2 && (it & 1) == 0) {  // 8 DMA pieces of 1 KiB per wave per two steps, into the upper half of the LDS image (not read)
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const unsigned off = (unsigned)((((size_t)blockIdx.x * 64 + it * 8 + p) * 8192 + w * 1024 + l * 16) % ((size_t)src_halves * 2 - 16)) & ~15u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lds + 32 * 1024 + (w * 8 + p) * 512 % (16 * 1024)), 16, off, 0, 0, 0);
            }
        }
        // (product-major: consecutive MFMAs write different accumulators, as the split GEMM's schedule does)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * x + (p == 1)], b[2 * y + (p == 2)], acc[x][y], 0, 0, 0);  // hi x hi | lo x hi | hi x lo
        if (MODE >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    float s = 0.f;
    for (int x = 0; x < 2; ++x)
        for (int y = 0; y < 4; ++y)
            for (int e = 0; e < 16; ++e) s += acc[x][y][e];
    if (s == 12345.678f) sink[0] = s;  // keeps the chain alive
}

static std::string g_pci;  // "0000:bb:dd.f" of the HIP device in use
static std::string hwmon_file(const char *name) {
    glob_t g;
    std::string pat = (g_pci.empty() ? std::string("/sys/class/drm/card*/device/hwmon/hwmon*/") : "/sys/bus/pci/devices/" + g_pci + "/hwmon/hwmon*/") + name;
    std::string r;
    if (glob(pat.c_str(), 0, nullptr, &g) == 0 && g.gl_pathc > 0) r = g.gl_pathv[0];
    globfree(&g);
    return r;
}
static double read_num(const std::string &p) {
    FILE *f = fopen(p.c_str(), "r");
    if (!f) return -1;
    double v = -1;
    if (fscanf(f, "%lf", &v) != 1) v = -1;
    fclose(f);
    return v;
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const int src_halves = 32 << 20;  // 64 MiB
    std::vector<_Float16> h(src_halves);
    _Float16 *d_norm, *d_small, *d_zero;
    float *sink;
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)((st >> 11) & 0xFFFFFFFFFFull) / (double)0x10000000000ull; };
    auto fill = [&](int kind) {
        for (int i = 0; i < src_halves; ++i) {
            double u = rnd() + rnd() + rnd() + rnd() - 2.0;  // ~N(0, 1/3)
            // kind 1: within a thread's 12 operand vectors, the "lo" ones (odd vectors of A, odd vectors of B) are small
            const int vec = (i / 8) % 12;
            const bool lo = vec < 4 ? (vec & 1) : ((vec - 4) & 1);
            h[i] = (_Float16)(kind == 2 ? 0.0 : (kind == 1 && lo) ? u * 0.00048828125 : u);
        }
    };
    if (hipMalloc(&d_norm, (size_t)src_halves * 2) != hipSuccess || hipMalloc(&d_small, (size_t)src_halves * 2) != hipSuccess ||
        hipMalloc(&d_zero, (size_t)src_halves * 2) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 2;
    fill(0); (void)hipMemcpy(d_norm, h.data(), (size_t)src_halves * 2, hipMemcpyHostToDevice);
    fill(1); (void)hipMemcpy(d_small, h.data(), (size_t)src_halves * 2, hipMemcpyHostToDevice);
    fill(2); (void)hipMemcpy(d_zero, h.data(), (size_t)src_halves * 2, hipMemcpyHostToDevice);
    {
        char bdf[64] = {0};
        if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), 0) == hipSuccess) {
            g_pci = bdf;
            for (char &ch : g_pci) ch = (char)tolower(ch);
            if (hwmon_file("freq1_input").empty()) g_pci.clear();
        }
    }
    const std::string fp = hwmon_file("power1_input").empty() ? hwmon_file("power1_average") : hwmon_file("power1_input");
    const std::string fc = hwmon_file("freq1_input"), fcap = hwmon_file("power1_cap");
    printf("power cap %.0f W; sampling %s\n", read_num(fcap) / 1e6, fp.c_str());
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    struct Case { const char *name; int mode; const _Float16 *src; };
    const Case cases[] = {{"mfma", 0, d_norm}, {"mfma_small", 0, d_small}, {"mfma_zero", 0, d_zero}, {"mfma+lds", 1, d_small}, {"mfma+lds+dma", 2, d_small}};
    const int iters = 4096, grid = 512;
    for (const Case &c : cases) {
        std::atomic<bool> stop{false};
        std::vector<double> pw, ck;
        std::thread sampler([&]() {
            while (!stop.load()) {
                const double p = read_num(fp), f = read_num(fc);
                if (p > 0) pw.push_back(p / 1e6);
                if (f > 0) ck.push_back(f / 1e6);
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
        });
        auto launch = [&]() {
            if (c.mode == 0) hipLaunchKernelGGL(probe_kernel<0>, dim3(grid), dim3(512), 0, 0, c.src, sink, iters, src_halves);
            else if (c.mode == 1) hipLaunchKernelGGL(probe_kernel<1>, dim3(grid), dim3(512), 64 * 1024, 0, c.src, sink, iters, src_halves);
            else hipLaunchKernelGGL(probe_kernel<2>, dim3(grid), dim3(512), 64 * 1024, 0, c.src, sink, iters, src_halves);
        };
        launch();
        (void)hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        long launches = 0;
        double el = 0;
        while (el < seconds) {
            for (int r = 0; r < 4; ++r) launch();
            (void)hipDeviceSynchronize();
            launches += 4;
            el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        stop.store(true);
        sampler.join();
        // drop the first second of samples (ramp)
        auto mean_tail = [](const std::vector<double> &v) { size_t s0 = v.size() > 14 ? 10 : 0; double s = 0; for (size_t i = s0; i < v.size(); ++i) s += v[i]; return v.size() > s0 ? s / (v.size() - s0) : -1.0; };
        const double flops = (double)launches * grid * 8.0 * iters * 24.0 * 2.0 * 32 * 32 * 16;
        printf("%-14s %8.1f TF executed  (%.3f of 2.5 PF)  power %7.1f W  sclk %7.1f MHz  err=%s\n", c.name, flops / el / 1e12, flops / el / 2.5e15,
               mean_tail(pw), mean_tail(ck), hipGetErrorString(hipGetLastError()));
        fflush(stdout);
    }
    return 0;
}
