// Probe of gfx950's LDS transpose read (ds_read_b64_tr_b16), the instruction the split attention reads its V fragments with
// (csrc/attention.hip).  Every lane supplies the address of 4 contiguous halves; within each group of 16 lanes the 16 x 4 block
// is delivered transposed.  Prints, for every lane and element, which (source lane, source element) the value came from, and
// checks the rule the kernel relies on:   out[l][j] = in[16 (l / 16) + 4 j + (l % 16) / 4][(l % 16) % 4].
//   hipcc --offload-arch=gfx950 -O2 -x hip tools/tr16_probe.cpp -o tools/tr16_probe && tools/tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3))) lds_s16x4;

__global__ void probe(short *out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 4];
    const int l = threadIdx.x;
    for (int j = 0; j < 4; ++j) lds[l * 4 + j] = (short)(l * 4 + j);  // value = 4 * source lane + source element
    __syncthreads();
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(lds + l * 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

// v_permlane32_swap_b32 x, y: x[l + 32] <-> y[l] (the split attention trades register quads between the two halves of a wave with it)
__global__ void probe_swap(int *out) {
    const int l = threadIdx.x;
    int x = l, y = 100 + l;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    out[l] = x;
    out[64 + l] = y;
}

// v_fma_mix_f32 with an f16 operand taken from the low / high half of a packed register: v - (float)hi in one instruction (pair_split4)
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__global__ void probe_mix(const float *in, float *out_mix, float *out_ref, int n) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * l + 1 >= n) return;
    const float v0 = in[2 * l], v1 = in[2 * l + 1];
    f16x2_t h = {(_Float16)v0, (_Float16)v1};
    asm volatile("" : "+v"(h));
    const unsigned hp = __builtin_bit_cast(unsigned, h);
    float d0, d1;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(hp), "v"(-1.0f), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(hp), "v"(-1.0f), "v"(v1));
    out_mix[2 * l] = d0; out_mix[2 * l + 1] = d1;
    out_ref[2 * l] = v0 - (float)h[0]; out_ref[2 * l + 1] = v1 - (float)h[1];
}

int main() {
    {
        const int n = 1 << 20;
        float *din, *dm, *dr;
        float *hin = new float[n], *hm = new float[n], *hr = new float[n];
        unsigned long long st = 88172645463325252ull;
        for (int j = 0; j < n; ++j) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            // magnitudes from 2^-30 (hi and lo subnormal or zero) to 2^15, both signs
            const int e = (int)((st >> 40) % 46) - 30;
            const float m = 1.0f + (float)((st >> 8) & 0xFFFFFF) / 16777216.0f;
            hin[j] = ((st & 1) ? -1.0f : 1.0f) * m * __builtin_ldexpf(1.0f, e);
        }
        if (hipMalloc(&din, n * 4) != hipSuccess || hipMalloc(&dm, n * 4) != hipSuccess || hipMalloc(&dr, n * 4) != hipSuccess) return 2;
        (void)hipMemcpy(din, hin, n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe_mix, dim3(n / 2 / 256), dim3(256), 0, 0, din, dm, dr, n);
        if (hipMemcpy(hm, dm, n * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(hr, dr, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        int bad = 0;
        for (int j = 0; j < n; ++j) bad += __builtin_memcmp(&hm[j], &hr[j], 4) != 0;
        printf("v_fma_mix_f32 (f16 half * -1 + v) against v - (float)hi on %d values of magnitude 2^-30 .. 2^15: %d mismatches\n", n, bad);
        if (bad) return 1;
    }
    {
        int *ds, hs[128];
        if (hipMalloc(&ds, sizeof(hs)) != hipSuccess) return 2;
        hipLaunchKernelGGL(probe_swap, dim3(1), dim3(64), 0, 0, ds);
        if (hipMemcpy(hs, ds, sizeof(hs), hipMemcpyDeviceToHost) != hipSuccess) return 2;
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            const int want_x = l < 32 ? l : 100 + (l - 32), want_y = l < 32 ? l + 32 : 100 + l;
            bad += hs[l] != want_x || hs[64 + l] != want_y;
        }
        printf("v_permlane32_swap_b32: %s (x[l + 32] <-> y[l]), %d mismatches\n", bad ? "DIFFERENT RULE" : "rule confirmed", bad);
        if (bad) return 1;
    }
    short *d, h[256];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 2;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int src_lane = h[l * 4 + j] >> 2, src_elem = h[l * 4 + j] & 3;
            const int want_lane = 16 * (l / 16) + 4 * j + (l % 16) / 4, want_elem = (l % 16) % 4;
            printf("  (%2d,%d)%s", src_lane, src_elem, (src_lane == want_lane && src_elem == want_elem) ? "" : "!");
            bad += !(src_lane == want_lane && src_elem == want_elem);
        }
        printf("\n");
    }
    printf("%s: %d mismatches against out[l][j] = in[16 (l / 16) + 4 j + (l %% 16) / 4][(l %% 16) %% 4]\n", bad ? "DIFFERENT RULE" : "rule confirmed", bad);
    return bad ? 1 : 0;
}
