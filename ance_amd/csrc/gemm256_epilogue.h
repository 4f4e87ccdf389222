// Epilogues of the 256 x 256 fp16 GEMM kernel (gemm256_f16.hip).
// Swapped MFMA orientation: acc[x][y][r] holds  n = n0 + wn*64 + x*32 + (r&3) + 8*(r>>2) + 4*g,
// m = m0 + wm*128 + y*32 + i  -- a lane owns one output row m and 4 consecutive n per register
// quad, staged through a wave-private 16 KiB LDS slab so that global traffic is whole 16-byte row
// segments.  Must be entered by all 512 threads after the last LDS read of the main loop.
#pragma once
#include "common.h"
#include "gemm_f16.h"

namespace ance {

// GELU(x) = x * Phi(x) = max(x, 0) - |x| * h(|x| / sqrt 2),  h(z) = erfc(z) / 2 = 2^q(z).
// q is a degree-5 least-squares fit of log2(erfc(z) / 2) on [0, 5] weighted by z h(z) (the factor
// the error is multiplied with); beyond 5 q keeps falling, so h underflows to 0 as it should.
// |GELU error| <= 8e-6 over [-9, 9] in fp32 (the stored result is fp16: 2^-11 relative), checked
// against scipy's erf when the coefficients were fitted.  One transcendental (v_exp_f32) and, on four
// adjacent columns at a time, packed bias add and final multiply-subtract -- the erfc rational form
// (Abramowitz-Stegun 7.1.26) this replaces needed v_rcp + v_exp + 17 scalar ops per element, and this epilogue runs on 3072 columns
// per token with no MFMA work to hide behind (one workgroup per CU).
typedef float f32x2 __attribute__((ext_vector_type(2)));
// coefficients live in constant memory (scalar loads) rather than as instruction literals: with literals hipcc
// emits one v_fmaak_f32 per element, with register operands the Horner steps become v_pk_fma_f32 (two elements
// per issue slot)
__constant__ float kGeluQ[6] = {-1.00054646f, -1.62252474f, -0.934321642f, -0.129834279f, 0.0201726463f, -0.00133047544f};

__device__ __forceinline__ f32x4 gelu_erf256(f32x4 x) {
    const f32x4 ax = __builtin_elementwise_abs(x);
    f32x4 out;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f32x2 a2 = {ax[2 * p], ax[2 * p + 1]};
        const f32x2 z = a2 * 0.70710678118654752440f;
        f32x2 q = z * kGeluQ[5] + kGeluQ[4];
        q = q * z + kGeluQ[3];
        q = q * z + kGeluQ[2];
        q = q * z + kGeluQ[1];
        q = q * z + kGeluQ[0];
        const f32x2 h = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
        const f32x2 x2 = {x[2 * p], x[2 * p + 1]};
        const f32x2 r = __builtin_elementwise_max(x2, f32x2{0.0f, 0.0f}) - a2 * h;
        out[2 * p] = r[0];
        out[2 * p + 1] = r[1];
    }
    return out;
}

// WAVE_SYNC: the slabs are wave-private and, once the main loop has returned, no wave reads the stage buffers any more
// (pipe256.h: the last barrier every wave passes comes after the last LDS read of both wave groups), so the write ->
// read-back -> next-pass-write ordering inside a slab is a matter of ONE wave: LDS operations of a wave execute in order,
// and a wavefront-scope fence keeps the compiler from moving a lane's read above another lane's write.  Without it every
// pass costs two workgroup barriers that make all eight waves wait for the slowest one.
template <bool WAVE_SYNC>
__device__ __forceinline__ void epi_sync() {
    if constexpr (WAVE_SYNC) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// LDS-DMA of the epilogue parameter block (gemm_f16.h: EPB_*), issued by all 512 threads BEFORE the main loop: older
// than every LDS-DMA of the pipeline, so the pipeline's first counted wait retires it and its first barrier publishes it.
template <int EPI_>
__device__ __forceinline__ void epb_issue(const GemmArgs &G, float *smem_f, int m0, int n0, int w, int l) {
    typedef __attribute__((address_space(3))) void lds_t;
    typedef const __attribute__((address_space(1))) void glb_t;
    float *pb = smem_f + EPB_OFF;
    const int tok0 = EPI_ == EPI_VT_F ? n0 : m0;  // first of the tile's 256 tokens
    const float *psrc = G.part_in + (size_t)tok0 * 24;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int piece = w + 8 * j;  // 24 pieces of 1 KiB
        __builtin_amdgcn_global_load_lds((glb_t *)(psrc + piece * 256 + l * 4), (lds_t *)(pb + EPB_PART + piece * 256), 16, 0, 0);
    }
    const int f0 = EPI_ == EPI_VT_F ? m0 : n0;    // first of the tile's 256 features
    if (w == 0) __builtin_amdgcn_global_load_lds((glb_t *)(G.bias + f0 + l * 4), (lds_t *)(pb + EPB_VEC), 16, 0, 0);
    if (EPI_ == EPI_RESLN || EPI_ == EPI_S_RESLN) {
        if (w == 1) __builtin_amdgcn_global_load_lds((glb_t *)(G.res_gamma + f0 + l * 4), (lds_t *)(pb + EPB_VEC + 256), 16, 0, 0);
        if (w == 2) __builtin_amdgcn_global_load_lds((glb_t *)(G.res_beta + f0 + l * 4), (lds_t *)(pb + EPB_VEC + 512), 16, 0, 0);
    } else {
        if (w == 1) __builtin_amdgcn_global_load_lds((glb_t *)(G.csum + f0 + l * 4), (lds_t *)(pb + EPB_VEC + 256), 16, 0, 0);
    }
}

// first step of a folded epilogue (all 512 threads, after the main loop): thread t < 256 combines the slice partials of
// token t into (mean, rstd); one workgroup barrier publishes them.  Returns (uniformly) whether any token of the tile has
// |mean| rstd > FOLD_WIDE_MEAN: the single-fp16 token operand of the folded GEMMs rounds v, not v - mean, so its error grows
// with |mean| / std (20-80 x at an offset of 5-30 std); such a tile adds a second pass over the lo halves.
constexpr float FOLD_WIDE_MEAN = 2.0f;
__device__ __forceinline__ bool epb_stats(const GemmArgs &G, float *smem_f, int tid) {
    float *pb = smem_f + EPB_OFF;
    int wide = 0;
    if (tid < 256) {
        float mean, rstd;
        stats_from_parts(pb + EPB_PART + tid * 24, G.ln_eps, &mean, &rstd);
        pb[EPB_STATS + 2 * tid] = mean;
        pb[EPB_STATS + 2 * tid + 1] = rstd;
        wide = __builtin_fabsf(mean) * rstd > FOLD_WIDE_MEAN;
    }
    return __builtin_amdgcn_readfirstlane(__syncthreads_or(wide)) != 0;  // (readfirstlane: the compiler must know it is uniform)
}

#ifdef ANCE_MEASURE
// measurement library only (WRONG results): bit 0 -- the RESLN epilogue reads its residual rows from rows 0..31 of the tile's
// slice (cache-resident), bit 1 -- it writes its output rows there: how much of the epilogue is its HBM traffic
__device__ int g_res_ablate = 0;
#endif

template <int EPI_, bool WAVE_SYNC = false>
__device__ __forceinline__ void gemm256_epilogue(const GemmArgs &G, f32x16 (&acc)[2][4], float *smem_f, int m0, int n0,
                                                 int w, int l, unsigned long long *pass_stamps = nullptr) {
    _Float16 *smem = reinterpret_cast<_Float16 *>(smem_f);
    const int g = l >> 5, i = l & 31;
    const int wm = w >> 2, wn = w & 3;
    // acc[x][y][r]: n = n0 + wn*64 + x*32 + (r&3) + 8*(r>>2) + 4*g ;  m = m0 + wm*128 + y*32 + i
    const int mw0 = m0 + wm * 128, nw0 = n0 + wn * 64;
    constexpr bool FOLD = EPI_ == EPI_QK_F || EPI_ == EPI_GELU_F || EPI_ == EPI_VT_F;
    constexpr int EPI = EPI_ == EPI_QK_F ? EPI_QK : EPI_ == EPI_GELU_F ? EPI_GELU : EPI_ == EPI_VT_F ? EPI_VT : EPI_;
    if constexpr (EPI == EPI_VT) {
        // Output rows are A-matrix rows m (features, bias per row); columns are tokens n scattered
        // through col_map (per-sequence 8-aligned key columns, so 16-byte stores are impossible in
        // general).  Slab [64 m][64 n] halves per pass; on read-back a lane owns ONE token column and
        // walks the 64 feature rows: every store instruction writes 64 consecutive tokens = 128 B of
        // one V^T row, and col_map is read once per lane.
        _Float16 *slab = smem + w * 8192;
        constexpr int LS = 72;
        const int ntok = nw0 + l;
        const bool tok_ok = ntok < G.n_valid;
        const int col = tok_ok ? G.col_map[ntok] : 0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            epi_sync<WAVE_SYNC>();
            float bias[2], cs[2] = {0.f, 0.f};  // per-lane row (feature) constants of the two 32-row blocks of this pass
#pragma unroll
            for (int yy = 0; yy < 2; ++yy) {
                if constexpr (FOLD) {
                    bias[yy] = smem_f[EPB_OFF + EPB_VEC + wm * 128 + (2 * p + yy) * 32 + i];
                    cs[yy] = smem_f[EPB_OFF + EPB_VEC + 256 + wm * 128 + (2 * p + yy) * 32 + i];
                } else {
                    bias[yy] = G.bias[mw0 + (2 * p + yy) * 32 + i];
                }
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    // FOLD: tokens are the columns -- (mean, rstd) of 4 consecutive tokens = 32 bytes of LDS, read once per
                    // pass and column quad; r (acc - mu c) + b = acc r + (b - (mu r) c)
                    f32x4 r4 = {1.f, 1.f, 1.f, 1.f}, mr4 = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (FOLD) {
                        const float *sp = smem_f + EPB_OFF + EPB_STATS + 2 * (wn * 64 + x * 32 + 8 * rq + 4 * g);
                        const f32x4 s01 = *reinterpret_cast<const f32x4 *>(sp);
                        const f32x4 s23 = *reinterpret_cast<const f32x4 *>(sp + 4);
                        r4 = f32x4{s01[1], s01[3], s23[1], s23[3]};
                        mr4 = f32x4{s01[0] * s01[1], s01[2] * s01[3], s23[0] * s23[1], s23[2] * s23[3]};
                    }
#pragma unroll
                    for (int yy = 0; yy < 2; ++yy) {
                        const f32x16 &a = acc[x][2 * p + yy];
                        f32x4 t = f32x4{a[4 * rq], a[4 * rq + 1], a[4 * rq + 2], a[4 * rq + 3]};
                        if constexpr (FOLD) t = t * r4 + (bias[yy] - mr4 * cs[yy]);
                        else t = t + bias[yy];
                        *reinterpret_cast<f16x4 *>(slab + (yy * 32 + i) * LS + x * 32 + 8 * rq + 4 * g) =
                            f16x4{(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
                    }
                }
            epi_sync<WAVE_SYNC>();
            if (tok_ok) {
                _Float16 *obase = G.out16 + (size_t)(mw0 + p * 64) * G.ldc + col;
#pragma unroll 8
                for (int rr = 0; rr < 64; ++rr) obase[(size_t)rr * G.ldc] = slab[rr * LS + l];
            }
        }
    } else if constexpr (EPI == EPI_RES32) {
        // wave-private slab [32 m][64 n] fp32, row stride 68 floats; 4 passes over the wave's 128 rows
        float *slab = smem_f + w * 4096;  // 16 KiB per wave
        constexpr int LS = 68;
        const int c4 = l & 15;
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(G.bias + nw0 + c4 * 4);
        const bool lazy_ln = G.res_stats != nullptr;  // uniform: residual = LayerNorm(res32 row) recomputed here
        f32x4 lng = {0, 0, 0, 0}, lnb = {0, 0, 0, 0};
        if (lazy_ln) {
            lng = *reinterpret_cast<const f32x4 *>(G.res_gamma + nw0 + c4 * 4);
            lnb = *reinterpret_cast<const f32x4 *>(G.res_beta + nw0 + c4 * 4);
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            // residual rows of this pass: issued first so their latency hides behind the LDS round trip
            f32x4 res[8];
            float mean[8], rstd[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int rr = it * 4 + (l >> 4);
                const size_t row = (size_t)(mw0 + y * 32 + rr);
                res[it] = *reinterpret_cast<const f32x4 *>(G.res32 + row * G.ldc + nw0 + c4 * 4);
                if (lazy_ln) {
                    mean[it] = G.res_stats[2 * row];
                    rstd[it] = G.res_stats[2 * row + 1];
                }
            }
            if (lazy_ln) {
#pragma unroll
                for (int it = 0; it < 8; ++it) res[it] = ln_apply4(res[it], mean[it], rstd[it], lng, lnb);
            }
            epi_sync<WAVE_SYNC>();
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x16 &a = acc[x][y];
                    *reinterpret_cast<f32x4 *>(slab + i * LS + x * 32 + 8 * rq + 4 * g) =
                        f32x4{a[4 * rq], a[4 * rq + 1], a[4 * rq + 2], a[4 * rq + 3]};
                }
            epi_sync<WAVE_SYNC>();
            // read back: 16 lanes cover one row (64 floats), 4 rows per instruction, 8 instructions
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int rr = it * 4 + (l >> 4);
                const f32x4 v = *reinterpret_cast<const f32x4 *>(slab + rr * LS + c4 * 4);
                *reinterpret_cast<f32x4 *>(G.out32 + (size_t)(mw0 + y * 32 + rr) * G.ldc + nw0 + c4 * 4) = v + bias + res[it];
            }
        }
    } else if constexpr (EPI == EPI_RESLN) {
        // as EPI_RES32, with the residual stream as an fp16 pair: v = acc + bias + LayerNorm(res_hi + res_lo);
        // out16 = fp16(v) (the next GEMM's token operand AND the high half of the stream), out_lo = fp16(v - out16)
        // (v - fp16(v) is exact in fp32; the pair carries 22 bits).  Per row and 64-column slice the (mean, M2) of v go
        // to part_out[] -- the consumers combine the N / 64 slices of a row (Chan) into (mean, rstd) in their own epilogues.
        float *slab = smem_f + w * 4096;
        constexpr int LS = 68;
        const int c4 = l & 15;
        const float *pb = smem_f + EPB_OFF;
        const f32x4 lng = *reinterpret_cast<const f32x4 *>(pb + EPB_VEC + 256 + wn * 64 + c4 * 4);
        const f32x4 bias_beta = *reinterpret_cast<const f32x4 *>(pb + EPB_VEC + wn * 64 + c4 * 4) +
                                *reinterpret_cast<const f32x4 *>(pb + EPB_VEC + 512 + wn * 64 + c4 * 4);
        const int n_parts = G.N >> 6, slice = nw0 >> 6;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            f16x4 rh[8], rl[8];
            float mean[8], rstd[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int rr = it * 4 + (l >> 4);
#ifdef ANCE_MEASURE
                const size_t row = (g_res_ablate & 1) ? (size_t)rr : (size_t)(mw0 + y * 32 + rr);
#else
                const size_t row = (size_t)(mw0 + y * 32 + rr);
#endif
                rh[it] = *reinterpret_cast<const f16x4 *>(G.res_hi + row * G.ldc + nw0 + c4 * 4);
                rl[it] = *reinterpret_cast<const f16x4 *>(G.res_lo + row * G.ldc + nw0 + c4 * 4);
                mean[it] = pb[EPB_STATS + 2 * (wm * 128 + y * 32 + rr)];
                rstd[it] = pb[EPB_STATS + 2 * (wm * 128 + y * 32 + rr) + 1];
            }
            epi_sync<WAVE_SYNC>();
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x16 &a = acc[x][y];
                    *reinterpret_cast<f32x4 *>(slab + i * LS + x * 32 + 8 * rq + 4 * g) =
                        f32x4{a[4 * rq], a[4 * rq + 1], a[4 * rq + 2], a[4 * rq + 3]};
                }
            epi_sync<WAVE_SYNC>();
            // running pointers (a row step is 4 rows): 64-bit address arithmetic per store was a fifth of this loop
            const size_t row0 = (size_t)(mw0 + y * 32 + (l >> 4));
#ifdef ANCE_MEASURE
            const size_t orow0 = (g_res_ablate & 2) ? (size_t)(l >> 4) : row0;
#else
            const size_t orow0 = row0;
#endif
            _Float16 *ph = G.out16 + orow0 * G.ldc + nw0 + c4 * 4;
            _Float16 *pl = G.out_lo + orow0 * G.ldc + nw0 + c4 * 4;
            const size_t rstep = (size_t)4 * G.ldc;
            f32x4 vv[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int rr = it * 4 + (l >> 4);
                // acc + bias + LayerNorm(hi + lo) = acc + hi a + (lo a + (bias + beta - mean a)),  a = rstd gamma
                f32x4 v = *reinterpret_cast<const f32x4 *>(slab + rr * LS + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = rstd[it] * lng[e];
                    const float b0 = __builtin_fmaf(-mean[it], a, bias_beta[e]);
                    v[e] += __builtin_fmaf((float)rh[it][e], a, __builtin_fmaf((float)rl[it][e], a, b0));
                }
                const f16x4 hi = cvt_f16x4_pinned(v);
                const f16x4 lo = f16x4{(_Float16)(v[0] - (float)hi[0]), (_Float16)(v[1] - (float)hi[1]),
                                       (_Float16)(v[2] - (float)hi[2]), (_Float16)(v[3] - (float)hi[3])};
                *reinterpret_cast<f16x4 *>(ph + it * rstep) = hi;
                *reinterpret_cast<f16x4 *>(pl + it * rstep) = lo;
                vv[it] = v;
            }
            // slice statistics of the 8 rows together: eight independent 16-lane reductions interleave (a DPP add right
            // behind the add that feeds it needs wait states; one row at a time the chain was serial)
            float s8[8], q8[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) s8[it] = (vv[it][0] + vv[it][1]) + (vv[it][2] + vv[it][3]);
#pragma unroll
            for (int it = 0; it < 8; ++it) s8[it] += __builtin_amdgcn_update_dpp(0.f, s8[it], 0xB1, 0xF, 0xF, true);
#pragma unroll
            for (int it = 0; it < 8; ++it) s8[it] += __builtin_amdgcn_update_dpp(0.f, s8[it], 0x4E, 0xF, 0xF, true);
#pragma unroll
            for (int it = 0; it < 8; ++it) s8[it] += __builtin_amdgcn_update_dpp(0.f, s8[it], 0x124, 0xF, 0xF, true);
#pragma unroll
            for (int it = 0; it < 8; ++it) s8[it] += __builtin_amdgcn_update_dpp(0.f, s8[it], 0x128, 0xF, 0xF, true);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const float m64 = s8[it] * (1.0f / 64.0f);
                s8[it] = m64;
                const float d0 = vv[it][0] - m64, d1 = vv[it][1] - m64, d2 = vv[it][2] - m64, d3 = vv[it][3] - m64;
                q8[it] = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) q8[it] += __builtin_amdgcn_update_dpp(0.f, q8[it], 0xB1, 0xF, 0xF, true);
#pragma unroll
            for (int it = 0; it < 8; ++it) q8[it] += __builtin_amdgcn_update_dpp(0.f, q8[it], 0x4E, 0xF, 0xF, true);
#pragma unroll
            for (int it = 0; it < 8; ++it) q8[it] += __builtin_amdgcn_update_dpp(0.f, q8[it], 0x124, 0xF, 0xF, true);
#pragma unroll
            for (int it = 0; it < 8; ++it) q8[it] += __builtin_amdgcn_update_dpp(0.f, q8[it], 0x128, 0xF, 0xF, true);
            if (c4 == 0) {
                float *pp = G.part_out + (row0 * n_parts + slice) * 2;
                const size_t pstep = (size_t)4 * n_parts * 2;
#pragma unroll
                for (int it = 0; it < 8; ++it) *reinterpret_cast<float2 *>(pp + it * pstep) = make_float2(s8[it], q8[it]);
            }
#ifdef ANCE_MEASURE
            if (pass_stamps && w == 0 && l == 0) pass_stamps[y] = __builtin_amdgcn_s_memrealtime();
#endif
        }
    } else {
        // fp16 outputs: slab [64 m][64 n] halves, row stride 72 halves (144 B); 2 passes
        _Float16 *slab = smem + w * 8192;  // 16 KiB per wave
        constexpr int LS = 72;
        // EPI_QK: the scale (1/8 and log2 e on Q) applies to columns < scale_cols -- a multiple of 64, so a wave's 64 columns
        // are all in or all out (per element this was a compare, a select and a multiply on every output)
        const float qscale = (EPI == EPI_QK && nw0 < G.scale_cols) ? G.scale : 1.0f;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            epi_sync<WAVE_SYNC>();
            // per-lane row constants of the two 32-row blocks of this pass (FOLD: r and mu r of the token; tokens are the rows)
            float rs[2] = {1.f, 1.f}, mrs[2] = {0.f, 0.f};
            if constexpr (FOLD) {
#pragma unroll
                for (int yy = 0; yy < 2; ++yy) {
                    const float *sp = smem_f + EPB_OFF + EPB_STATS + 2 * (wm * 128 + (2 * p + yy) * 32 + i);
                    rs[yy] = sp[1];
                    mrs[yy] = sp[0] * sp[1];
                }
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int nl = x * 32 + 8 * rq + 4 * g;  // local n of element 0
                    // column constants once per column quad (they come from LDS in the folded epilogues: one read per pass
                    // and quad instead of one per 32-row block)
                    f32x4 bias, cs = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (FOLD) {
                        bias = *reinterpret_cast<const f32x4 *>(smem_f + EPB_OFF + EPB_VEC + wn * 64 + nl);
                        cs = *reinterpret_cast<const f32x4 *>(smem_f + EPB_OFF + EPB_VEC + 256 + wn * 64 + nl);
                    } else {
                        bias = *reinterpret_cast<const f32x4 *>(G.bias + nw0 + nl);
                    }
#pragma unroll
                    for (int yy = 0; yy < 2; ++yy) {
                        const f32x16 &a = acc[x][2 * p + yy];
                        f32x4 t = f32x4{a[4 * rq], a[4 * rq + 1], a[4 * rq + 2], a[4 * rq + 3]};
                        if constexpr (FOLD) t = t * rs[yy] + (bias - mrs[yy] * cs);  // r (acc - mu c) + b'
                        else t = t + bias;
                        if constexpr (EPI == EPI_GELU) {
                            t = gelu_erf256(t);
                        } else {
                            t = t * qscale;
                        }
                        const f16x4 v = f16x4{(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
                        *reinterpret_cast<f16x4 *>(slab + (yy * 32 + i) * LS + nl) = v;
                    }
                }
            epi_sync<WAVE_SYNC>();
            // read back: 8 lanes cover one row (64 halves = 128 B), 8 rows per instruction
            const int c8 = l & 7;
            _Float16 *po = G.out16 + (size_t)(mw0 + p * 64 + (l >> 3)) * G.ldc + nw0 + c8 * 8;  // running pointer: 8 rows per step
            const size_t ostep = (size_t)8 * G.ldc;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int rr = it * 8 + (l >> 3);
                const f16x8 v = *reinterpret_cast<const f16x8 *>(slab + rr * LS + c8 * 8);
                *reinterpret_cast<f16x8 *>(po + it * ostep) = v;
            }
        }
    }
}

// ---- epilogues of the SPLIT (fp32-grade) GEMM ---------------------------------------------------------------------------
// erf-form GELU (the reference's: transformers "gelu" = x Phi(x)) to fp32 grade WITHOUT the library erff.  ocml's erff is two
// branches (both executed in a 64-lane wave: ~38 VALU instructions per output) and the GELU epilogue of the split FFN1 GEMM was
// VALU-bound on it (20.6 us of a 67.6 us tile against 14.4 us for the plain fp32 store epilogue).  Here
//     Phi(x) = x >= 0 ? 1 - e : e,    e = erfc(|x| / sqrt 2) / 2 = 2^q(z),  z = min(|x| / sqrt 2, 6.6)
// with q a degree-9 polynomial fit of log2(erfc(z) / 2) on [0, 6.6] (weighted for the ABSOLUTE error of e: approximation error
// 1.1e-9; monomial in z, so that near z = 0 the sum is -1 plus small terms; beyond z = 6.6 e < 2^-66): 9 fma + v_exp_f32 + 6.
// Measured against the exact value in fp32 emulation (tests/test_gelu_poly.py, 8 M points): |error| / |x| <= 1.1e-7 everywhere
// -- torch's own fp32 erf-GELU, which is what the reference runs, is at 3.7e-7 -- mean |error| 1.7e-8 (torch 4.5e-8).
// (contraction off here and in the split epilogue: hipcc contracts a * b + c into an fma in SOME of the unrolled instances of
// a loop and not in others, so a row's last bit would depend on which pass / register slot of the tile it lands in -- and with
// it on the micro-batch split and the number of GPUs.  Every fused operation below is written out as fmaf.)
constexpr float GELU_Q[10] = {-1.0f, -1.627907395362854f, -0.918441653251648f, -0.14831341803073883f, 0.02773732878267765f,
                              6.778987153666094e-05f, -0.002261603018268943f, 0.0008423461113125086f, -0.00015156660811044276f,
                              1.1468856428109575e-05f};
__device__ __forceinline__ float gelu_exact(float x) {
#pragma clang fp contract(off)
#ifdef ANCE_GELU_ERFF  // A/B builds only (make variant NAME=erff DEFS=-DANCE_GELU_ERFF): round 4's library erff
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
#endif
    const float z = __builtin_fminf(__builtin_fabsf(x) * 0.70710678118654752440f, 6.6f);
    float q = GELU_Q[9];
#pragma unroll
    for (int k = 8; k >= 0; --k) q = __builtin_fmaf(q, z, GELU_Q[k]);
    const float e = __builtin_amdgcn_exp2f(q);
    return x * (x >= 0.0f ? 1.0f - e : e);
}

// Four at a time on the PACKED fp32 pipe (round 6): v_pk_fma_f32 runs two IEEE fmas per issue slot, and the GELU epilogue is bound by
// its vector instructions (9 of its ~21 per element are the Horner steps: they were v_fmaak_f32, one element each, because the
// coefficients were literals).  The coefficients come from constant memory here (scalar registers, as kGeluQ above), the steps are
// element-wise fmas on float2 -- the same operations in the same order: bit-identical to gelu_exact.
__constant__ float kGeluExactQ[10] = {GELU_Q[0], GELU_Q[1], GELU_Q[2], GELU_Q[3], GELU_Q[4], GELU_Q[5], GELU_Q[6], GELU_Q[7], GELU_Q[8], GELU_Q[9]};
__device__ __forceinline__ f32x4 gelu_exact4(const f32x4 x) {
#pragma clang fp contract(off)
#ifdef ANCE_GELU_ERFF
    return f32x4{gelu_exact(x[0]), gelu_exact(x[1]), gelu_exact(x[2]), gelu_exact(x[3])};
#endif
    f32x4 out;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f32x2 x2 = {x[2 * p], x[2 * p + 1]};
        const f32x2 z = __builtin_elementwise_min(__builtin_elementwise_abs(x2) * 0.70710678118654752440f, f32x2{6.6f, 6.6f});
        f32x2 q = {kGeluExactQ[9], kGeluExactQ[9]};
#pragma unroll
        for (int k = 8; k >= 0; --k) q = __builtin_elementwise_fma(q, z, f32x2{kGeluExactQ[k], kGeluExactQ[k]});
        const float e0 = __builtin_amdgcn_exp2f(q[0]), e1 = __builtin_amdgcn_exp2f(q[1]);
        out[2 * p] = x2[0] * (x2[0] >= 0.0f ? 1.0f - e0 : e0);
        out[2 * p + 1] = x2[1] * (x2[1] >= 0.0f ? 1.0f - e1 : e1);
    }
    return out;
}

// Output stores of the split epilogues are NON-TEMPORAL: the outputs of a launch (0.6-1.6 GB) are consumed by the next kernel and
// only pass through the 4 MB L2s on their way out, where they evict the operand panels the main loops re-read.  Same-box A/B
// (profiles/r05_ab_nt_store.jsonl, three alternations): FFN1 -0.7 %, the attention that follows the QKV GEMM -2.5 %, step +0.3 %.
// (ANCE_EPI_PLAIN_STORE: A/B builds with ordinary stores.)
// The pair-row epilogues move 8 columns per lane (16-byte hi and 16-byte lo accesses): an epilogue is bound by the NUMBER of
// vector-memory instructions its eight waves push through the CU's one address unit (~16 cycles each whatever their width) -- the
// fp32 store epilogue of QKV (32 dwordx4 stores per wave and tile) measured 4 us, the GELU pair epilogue with 8-byte accesses (64
// stores) 8.4 us, RESLN (64 loads + 64 stores + 32 statistics stores) 13.5 us.  Round 6: half as many, twice as wide.
__device__ __forceinline__ f16x8 cat_f16x4(const f16x4 a, const f16x4 b) { return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
#ifndef ANCE_EPI_PLAIN_STORE
__device__ __forceinline__ void epi_pair_store_nt(const f32x4 v, _Float16 *row, int W, int n) {
    f16x4 h, r;
    pair_split4(v, &h, &r);
    __builtin_nontemporal_store(h, reinterpret_cast<f16x4 *>(row + pair_hi_col(n, W)));
    __builtin_nontemporal_store(r, reinterpret_cast<f16x4 *>(row + pair_lo_col(n, W)));
}
// columns n .. n + 7 (n a multiple of 8: inside one 32-column block)
__device__ __forceinline__ void epi_pair_store8_nt(const f32x4 va, const f32x4 vb, _Float16 *row, int W, int n) {
    f16x4 ha, ra, hb, rb;
    pair_split4(va, &ha, &ra);
    pair_split4(vb, &hb, &rb);
    __builtin_nontemporal_store(cat_f16x4(ha, hb), reinterpret_cast<f16x8 *>(row + pair_hi_col(n, W)));
    __builtin_nontemporal_store(cat_f16x4(ra, rb), reinterpret_cast<f16x8 *>(row + pair_lo_col(n, W)));
}
#define EPI_PAIR_STORE(v, row, W, n) epi_pair_store_nt(v, row, W, n)
#define EPI_PAIR_STORE8(va, vb, row, W, n) epi_pair_store8_nt(va, vb, row, W, n)
#define EPI_F32_STORE(p, v) __builtin_nontemporal_store(v, p)
#else
__device__ __forceinline__ void epi_pair_store8(const f32x4 va, const f32x4 vb, _Float16 *row, int W, int n) {
    f16x4 ha, ra, hb, rb;
    pair_split4(va, &ha, &ra);
    pair_split4(vb, &hb, &rb);
    *reinterpret_cast<f16x8 *>(row + pair_hi_col(n, W)) = cat_f16x4(ha, hb);
    *reinterpret_cast<f16x8 *>(row + pair_lo_col(n, W)) = cat_f16x4(ra, rb);
}
#define EPI_PAIR_STORE(v, row, W, n) pair_store4(v, row, W, n)
#define EPI_PAIR_STORE8(va, vb, row, W, n) epi_pair_store8(va, vb, row, W, n)
#define EPI_F32_STORE(p, v) (*(p) = (v))
#endif

// slice statistics of the pair epilogues: a lane holds 8 columns of a row, 4 lanes a 32-column block, 8 lanes the 64-column slice
__device__ __forceinline__ float quad_sum(float x) {  // every lane of the quad ends with the same bits
    x += __builtin_amdgcn_update_dpp(0.f, x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0.f, x, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    return x;
}
__device__ __forceinline__ float half_mirror(float x) { return __builtin_amdgcn_update_dpp(0.f, x, 0x141, 0xF, 0xF, true); }  // lane j <- lane 7 - j
__device__ __forceinline__ float sum8(const f32x4 a, const f32x4 b) { return ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3])); }
__device__ __forceinline__ float sumsq8(const f32x4 a, const f32x4 b, float m) {
    const float a0 = a[0] - m, a1 = a[1] - m, a2 = a[2] - m, a3 = a[3] - m, b0 = b[0] - m, b1 = b[1] - m, b2 = b[2] - m, b3 = b[3] - m;
    return ((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3)) + ((b0 * b0 + b1 * b1) + (b2 * b2 + b3 * b3));
}

// One structure for the three of them: 4 passes over the wave's 128 rows, each through the wave-private fp32 slab
// [32 m][64 n] (as EPI_RES32 / EPI_RESLN), so that on read-back a lane owns 4 consecutive columns of a row and global
// traffic is whole 16-byte (fp32) or 8-byte (fp16) row segments.
//   EPI_S_QKV    out32[m][n]         = r (acc' - mu c) + b'                   folded LayerNorm, fp32 out (Q | K | V)
//   EPI_S_GELU   out16 pair [m][n]   = pair(gelu_exact(r (acc' - mu c) + b'))   pair row of 2 N halves (common.h), ldc = 2 N
//   EPI_S_RESLN  out16 pair [m][n]   = pair(acc' + bias + LayerNorm(residual pair)), + slice statistics (part_out)
// acc' = acc winv: winv is the inverse of the power of two the weight was stored with (exact; it rides on the row's rstd or in
// the one fma that adds the residual, so it costs no instruction).
// Columns per lane on read-back: 8 for the pair-row outputs (16-byte hi and 16-byte lo accesses: 4 lanes write the 64 + 64 bytes of a
// pair block), 4 for the fp32 rows of QKV (16 lanes write 256 contiguous bytes; with 8 columns a lane's two 16-byte stores would
// interleave with its neighbours' -- measured +3.5 % on that GEMM).
template <int EPI>
__device__ __forceinline__ void gemm256_epilogue_split(const GemmArgs &G, f32x16 (&acc)[2][4], float *smem_f, int m0, int n0,
                                                       int w, int l, float winv) {
#pragma clang fp contract(off)
    const int g = l >> 5, i = l & 31;
    const int wm = w >> 2, wn = w & 3;
    const int mw0 = m0 + wm * 128, nw0 = n0 + wn * 64;
    float *slab = smem_f + w * 4096;
    constexpr int LS = 68;
    constexpr int CPL = EPI == EPI_S_QKV ? 4 : 8, NV = CPL / 4;  // columns per lane, f32x4 per lane and row
    constexpr int LPR = 64 / CPL, RPI = 64 / LPR, ITS = 32 / RPI;  // lanes per row, rows per instruction, instructions per pass
    const int cl = l % LPR, rl_ = l / LPR;
    const int nc = nw0 + cl * CPL;      // first of this lane's columns
    const float *pb = smem_f + EPB_OFF;
    const float *vp = pb + EPB_VEC + wn * 64 + cl * CPL;
    f32x4 v0[NV], v1[NV], v2[NV];       // bias (b' for the folded ones) | csum or gamma | beta
#pragma unroll
    for (int h = 0; h < NV; ++h) {
        v0[h] = *reinterpret_cast<const f32x4 *>(vp + 4 * h);
        v1[h] = *reinterpret_cast<const f32x4 *>(vp + 256 + 4 * h);
        v2[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == EPI_S_RESLN) v2[h] = *reinterpret_cast<const f32x4 *>(vp + 512 + 4 * h);
    }
    const int n_parts = G.N >> 6, slice = nw0 >> 6;
    float vmax = 0.f;  // range guard: running maximum of |what this thread stores| (common.h)
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        f16x8 rh[ITS], rl[ITS];
        float mean[ITS], rstd[ITS];
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int rr = it * RPI + rl_;
            if constexpr (EPI == EPI_S_RESLN) {
                const _Float16 *rp = G.res_hi + (size_t)(mw0 + y * 32 + rr) * G.ldr;
                rh[it] = *reinterpret_cast<const f16x8 *>(rp + pair_hi_col(nc, G.N));
                rl[it] = *reinterpret_cast<const f16x8 *>(rp + pair_lo_col(nc, G.N));
            }
            mean[it] = pb[EPB_STATS + 2 * (wm * 128 + y * 32 + rr)];
            rstd[it] = pb[EPB_STATS + 2 * (wm * 128 + y * 32 + rr) + 1];
        }
        epi_sync<true>();
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x16 &a = acc[x][y];
                *reinterpret_cast<f32x4 *>(slab + i * LS + x * 32 + 8 * rq + 4 * g) =
                    f32x4{a[4 * rq], a[4 * rq + 1], a[4 * rq + 2], a[4 * rq + 3]};
            }
        epi_sync<true>();
        f32x4 vv[ITS][NV];
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int rr = it * RPI + rl_;
            const size_t row = (size_t)(mw0 + y * 32 + rr);
#pragma unroll
            for (int h = 0; h < NV; ++h) {
                f32x4 a = *reinterpret_cast<const f32x4 *>(slab + rr * LS + cl * CPL + 4 * h);
                if constexpr (EPI == EPI_S_RESLN) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float ga = rstd[it] * v1[h][e];
                        const float ra = (float)rh[it][4 * h + e] + (float)rl[it][4 * h + e] * PAIR_LO_INV;  // exact in fp32: 22 bits
                        a[e] = __builtin_fmaf(a[e], winv, __builtin_fmaf(ra - mean[it], ga, v0[h][e] + v2[h][e]));
                    }
                } else {
                    const float mr = mean[it] * rstd[it], rw = rstd[it] * winv;  // r (acc winv) = acc (r winv): winv is a power of two
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = __builtin_fmaf(a[e], rw, __builtin_fmaf(-mr, v1[h][e], v0[h][e]));
                    if constexpr (EPI == EPI_S_GELU) a = gelu_exact4(a);
                }
                range_track4(a, &vmax);  // (QKV: the attention splits K and V into pairs while it stages them)
                vv[it][h] = a;
            }
            if constexpr (EPI == EPI_S_QKV) EPI_F32_STORE(reinterpret_cast<f32x4 *>(G.out32 + row * G.ldc + nc), vv[it][0]);
            else EPI_PAIR_STORE8(vv[it][0], vv[it][NV - 1], G.out16 + row * G.ldc, G.N, nc);
        }
        if constexpr (EPI == EPI_S_RESLN) {
            // (mean, M2) of the 64 columns of every row: lane sums of 8 columns, quad sums (32 columns), the two quads of the row
            float s4[ITS], q4[ITS];
#pragma unroll
            for (int it = 0; it < ITS; ++it) {
                const float s = quad_sum(sum8(vv[it][0], vv[it][NV - 1]));
                s4[it] = (s + half_mirror(s)) * (1.0f / 64.0f);
            }
#pragma unroll
            for (int it = 0; it < ITS; ++it) {
                const float q = quad_sum(sumsq8(vv[it][0], vv[it][NV - 1], s4[it]));
                q4[it] = q + half_mirror(q);
            }
            // every lane of a row holds the row's (s, q) of all ITS row groups: lane cl < ITS stores group cl -- ONE store instruction
            // for the 32 rows of the pass instead of ITS (the epilogue is bound by its vector-memory instruction count)
            static_assert(ITS == 4 && LPR >= 4, "one lane per row group");
            const float ss = cl == 0 ? s4[0] : cl == 1 ? s4[1] : cl == 2 ? s4[2] : s4[3];
            const float qq = cl == 0 ? q4[0] : cl == 1 ? q4[1] : cl == 2 ? q4[2] : q4[3];
            if (cl < ITS)
                *reinterpret_cast<float2 *>(G.part_out + ((size_t)(mw0 + y * 32 + cl * RPI + rl_) * n_parts + slice) * 2) = make_float2(ss, qq);
        }
    }
    range_report(vmax, G.range_faults);
}

// ---- epilogues of the STREAMING (persistent) split GEMM: EPI_S_QKV and EPI_S_GELU -----------------------------------------
// The same arithmetic, bit for bit (tests/test_gpu_gemm.py compares the two kernels with array_equal), in the LDS the persistent
// kernel has left while the next output tile's first K-tiles are in flight in the stage buffers: a wave-private slab of [32 m][32 n]
// fp32 (row stride 36 floats: the 16-byte writes of 16 lanes fall into 16 different bank quads; 4.5 KiB per wave instead of the
// 16 KiB slices of the stage buffers the launch-per-tile kernel's epilogue reuses), EIGHT passes (y, x) of 32 rows x 32 columns.
// A 32-column block is exactly one [hi (32) | lo (32)] block of a pair row (common.h): on read-back a row of the pass is
// 8 lanes x 16 bytes = 128 contiguous bytes of fp32 (EPI_S_QKV) or 4 lanes x (16 + 16) = the 64 + 64 bytes of one pair block.
// (EPI_S_RESLN in this form -- residual rows requested a pass ahead, slice statistics combined over the passes x = 0, 1 -- measured
// 1.6-3.4 us per tile slower than the 32 x 64 form: commit abaa8b9, DESIGN_REJECTED.md round 6.)
constexpr int EPS_LS = 36;                       // slab row stride (floats)
constexpr int EPS_SLAB_FLOATS = 32 * EPS_LS;     // 4,608 bytes per wave

template <int EPI>
__device__ __forceinline__ void gemm256_epilogue_split32(const GemmArgs &G, f32x16 (&acc)[2][4], float *slab, const float *stats,
                                                         const float *vec, int m0, int n0, int w, int l, float winv) {
#pragma clang fp contract(off)
    static_assert(EPI == EPI_S_QKV || EPI == EPI_S_GELU, "the RESLN GEMMs run the launch-per-tile kernel (DESIGN_REJECTED.md round 6)");
    const int g = l >> 5, i = l & 31;
    const int wm = w >> 2, wn = w & 3;
    const int mw0 = m0 + wm * 128, nw0 = n0 + wn * 64;
    constexpr int LS = EPS_LS;
    constexpr int CPL = EPI == EPI_S_QKV ? 4 : 8, NV = CPL / 4;  // columns per lane (gemm256_epilogue_split), f32x4 per lane and row
    constexpr int LPR = 32 / CPL, RPI = 64 / LPR, ITS = 32 / RPI;  // lanes per row of the pass, rows per instruction, instructions per pass
    const int cl = l % LPR, rl_ = l / LPR;
    float vmax = 0.f;  // range guard (common.h)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int y = p >> 1, x = p & 1;
        const float *vp = vec + wn * 64 + x * 32 + cl * CPL;
        f32x4 v0[NV], v1[NV];  // b' | csum
#pragma unroll
        for (int h = 0; h < NV; ++h) {
            v0[h] = *reinterpret_cast<const f32x4 *>(vp + 4 * h);
            v1[h] = *reinterpret_cast<const f32x4 *>(vp + 256 + 4 * h);
        }
        epi_sync<true>();
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const f32x16 &a = acc[x][y];
            *reinterpret_cast<f32x4 *>(slab + i * LS + 8 * rq + 4 * g) = f32x4{a[4 * rq], a[4 * rq + 1], a[4 * rq + 2], a[4 * rq + 3]};
        }
        epi_sync<true>();
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int rr = it * RPI + rl_;
            const float mean = stats[2 * (wm * 128 + y * 32 + rr)], rstd = stats[2 * (wm * 128 + y * 32 + rr) + 1];
            const float mr = mean * rstd, rw = rstd * winv;  // r (acc winv) = acc (r winv): winv is a power of two
            const size_t row = (size_t)(mw0 + y * 32 + rr);
            const int n = nw0 + x * 32 + cl * CPL;
            f32x4 vv[NV];
#pragma unroll
            for (int h = 0; h < NV; ++h) {
                f32x4 a = *reinterpret_cast<const f32x4 *>(slab + rr * LS + cl * CPL + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = __builtin_fmaf(a[e], rw, __builtin_fmaf(-mr, v1[h][e], v0[h][e]));
                if constexpr (EPI == EPI_S_GELU) a = gelu_exact4(a);
                range_track4(a, &vmax);
                vv[h] = a;
            }
            if constexpr (EPI == EPI_S_QKV) EPI_F32_STORE(reinterpret_cast<f32x4 *>(G.out32 + row * G.ldc + n), vv[0]);
            else EPI_PAIR_STORE8(vv[0], vv[NV - 1], G.out16 + row * G.ldc, G.N, n);
        }
    }
    range_report(vmax, G.range_faults);
}

}  // namespace ance
