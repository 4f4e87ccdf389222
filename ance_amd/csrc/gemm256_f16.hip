// 256 x 256 x 64 fp16 MFMA GEMM for the encoder (gfx950), same contract as gemm_f16.hip:
//   C[m][n] = sum_k A[m][k] * B[n][k],  A [M,K] and B [N,K] row-major f16, fp32 accumulation.
//
// Why this tile: a 128 x 128 tile has 64 FLOP per staged byte, i.e. 39 TB/s of L2 traffic at the
// 2.5 PFLOP/s MFMA rate -- more than the 34.5 TB/s the eight L2s deliver -- so it is L2-bound by
// construction.  256 x 256 halves that.  8 waves (2 per SIMD), one workgroup per CU, 2 x 64 KiB LDS
// stages, one barrier per K-tile.
//
// MFMA orientation is SWAPPED with respect to the output: MFMA rows (A operand) are B-matrix rows
// n, MFMA columns (B operand) are A-matrix rows m.  In the C-layout a lane then owns one output
// row m and, per register quad, 4 CONSECUTIVE n -- so the epilogue writes 8/16-byte pieces into a
// wave-private row-major LDS slab and reads it back as whole 16-byte row segments: global stores
// (and the residual read) are full-width and coalesced instead of 4-byte scalars.
//
// Staging variants (template GLDS): register staged (global_load_dwordx4 -> ds_write_b128) or
// direct-to-LDS (global_load_lds_dwordx4, the LDS image is lane-linear so the XOR swizzle is
// applied to the per-lane SOURCE address).  Same LDS image, same reads.
#include "common.h"
#include "gemm_f16.h"

namespace ance {
namespace {

constexpr int TM = 256, TN = 256, TK = 64;      // TK halves = 128 B per LDS row
constexpr int OPER_HALVES = 256 * TK;           // one operand tile (32 KiB)
constexpr int STAGE_HALVES = 2 * OPER_HALVES;   // A-rows tile + B-rows tile (64 KiB)
constexpr int G256_THREADS = 512;
constexpr size_t G256_LDS_BYTES = (size_t)2 * STAGE_HALVES * sizeof(_Float16);  // 128 KiB

// GELU(x) = x * Phi(x) with Phi from the Abramowitz-Stegun 7.1.26 erfc polynomial (|abs err| of
// erf <= 1.5e-7, far below the fp16 resolution of the stored result): ~14 VALU ops instead of the
// ~30 of ocml's erff, which matters because this epilogue runs on 3072 columns per token with no
// MFMA work to hide behind.  The erfc form keeps the negative tail free of cancellation.
__device__ __forceinline__ float gelu_erf256(float x) {
    const float z = x * 0.70710678118654752440f;
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float h = 0.5f * p * t * __expf(-az * az);  // 0.5 * erfc(|z|)
    return x * (z >= 0.0f ? 1.0f - h : h);
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int EPI, bool GLDS>
__global__ void __launch_bounds__(G256_THREADS, 2) gemm256_f16_kernel(const GemmArgs G) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    _Float16 *smem = reinterpret_cast<_Float16 *>(smem_f);

    const int NT = G.N / TN, MT = G.M / TM;
    // XCD-aware tile order (speed only): blocks b, b+8, ... share an XCD.  The dimension with more
    // tiles is dealt round-robin to the XCDs, the other one is swept fastest, so the panel of the
    // outer dimension stays in that XCD's L2 while the inner panels stream through it.
    const int b = blockIdx.x, xcd = b & 7, jx = b >> 3;
    int mt, nt;
    if (MT >= NT) {
        mt = (jx / NT) * 8 + xcd;
        nt = jx % NT;
    } else {
        nt = (jx / MT) * 8 + xcd;
        mt = jx % MT;
    }
    if (mt >= MT || nt >= NT) return;
    const int m0 = mt * TM, n0 = nt * TN;

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63, g = l >> 5, i = l & 31;
    const int wm = w >> 2;  // 2 x 128 output rows m
    const int wn = w & 3;   // 4 x 64 output cols n

    // ---- staging ------------------------------------------------------------------------------
    // LDS image of an operand tile: row r (128 B), 16-byte chunk c stored in slot c ^ ((r >> 1) & 7).
    // One wave-instruction covers 8 rows x 8 chunks = 1 KiB; wave w handles row groups w, w+8, ...
    // lane L: row-in-group L >> 3, LDS slot L & 7  ->  source chunk = slot ^ swizzle(row).
    const int rg = l >> 3, slot = l & 7;
    const _Float16 *srcA[4], *srcB[4];
    int ldsoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (j * 8 + w) * 8 + rg;
        const int ch = slot ^ ((row >> 1) & 7);
        srcA[j] = G.A + (size_t)(m0 + row) * G.lda + ch * 8;
        srcB[j] = G.B + (size_t)(n0 + row) * G.ldb + ch * 8;
        ldsoff[j] = row * TK + slot * 8;  // halves; lane-linear within the wave's 1 KiB piece
    }
    f16x8 ra[4], rb[4];
    auto stage_issue = [&](int kt, int buf) {
        const int k0 = kt * TK;
        if constexpr (GLDS) {
            _Float16 *sa = smem + buf * STAGE_HALVES;
            _Float16 *sb = sa + OPER_HALVES;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int piece = ((j * 8 + w) * 8) * TK;  // wave-uniform LDS base of this 1 KiB piece
                __builtin_amdgcn_global_load_lds((glb_void_t *)(srcA[j] + k0), (lds_void_t *)(sa + piece), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_void_t *)(srcB[j] + k0), (lds_void_t *)(sb + piece), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ra[j] = *reinterpret_cast<const f16x8 *>(srcA[j] + k0);
                rb[j] = *reinterpret_cast<const f16x8 *>(srcB[j] + k0);
            }
        }
    };
    auto stage_commit = [&](int buf) {  // register path only
        if constexpr (!GLDS) {
            _Float16 *sa = smem + buf * STAGE_HALVES;
            _Float16 *sb = sa + OPER_HALVES;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<f16x8 *>(sa + ldsoff[j]) = ra[j];
                *reinterpret_cast<f16x8 *>(sb + ldsoff[j]) = rb[j];
            }
        }
    };

    // ---- fragment rows: MFMA rows <- B-matrix rows (n), MFMA cols <- A-matrix rows (m) -----------
    int nrow[2], nsw[2], mrow[4], msw[4];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        nrow[x] = wn * 64 + x * 32 + i;
        nsw[x] = (nrow[x] >> 1) & 7;
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        mrow[x] = wm * 128 + x * 32 + i;
        msw[x] = (mrow[x] >> 1) & 7;
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = f32x16{0};

    const int NK = G.K / TK;
    stage_issue(0, 0);
    if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stage_commit(0);
    __syncthreads();
    for (int kt = 0; kt < NK; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < NK) stage_issue(kt + 1, buf ^ 1);
        const _Float16 *sa = smem + buf * STAGE_HALVES;  // A-matrix rows (m)
        const _Float16 *sb = sa + OPER_HALVES;           // B-matrix rows (n)
        // fragments of k-step s+1 are requested before the MFMAs of k-step s are issued
        f16x8 fn[2][2], fm[2][4];
        auto load_frags = [&](int s, int set) {
            const int ch = 2 * s + g;
#pragma unroll
            for (int x = 0; x < 2; ++x)
                fn[set][x] = *reinterpret_cast<const f16x8 *>(sb + nrow[x] * TK + ((ch ^ nsw[x]) * 8));
#pragma unroll
            for (int y = 0; y < 4; ++y)
                fm[set][y] = *reinterpret_cast<const f16x8 *>(sa + mrow[y] * TK + ((ch ^ msw[y]) * 8));
        };
        load_frags(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) load_frags(s + 1, (s + 1) & 1);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fn[s & 1][x], fm[s & 1][y], acc[x][y], 0, 0, 0);
        }
        if (kt + 1 < NK) {
            if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stage_commit(buf ^ 1);
        }
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------------------
    // acc[x][y][r]: n = n0 + wn*64 + x*32 + (r&3) + 8*(r>>2) + 4*g ;  m = m0 + wm*128 + y*32 + i
    const int mw0 = m0 + wm * 128, nw0 = n0 + wn * 64;
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
            __syncthreads();
#pragma unroll
            for (int yy = 0; yy < 2; ++yy) {
                const float bias = G.bias[mw0 + (2 * p + yy) * 32 + i];
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const f32x16 &a = acc[x][2 * p + yy];
                        *reinterpret_cast<f16x4 *>(slab + (yy * 32 + i) * LS + x * 32 + 8 * rq + 4 * g) =
                            f16x4{(_Float16)(a[4 * rq] + bias), (_Float16)(a[4 * rq + 1] + bias),
                                  (_Float16)(a[4 * rq + 2] + bias), (_Float16)(a[4 * rq + 3] + bias)};
                    }
            }
            __syncthreads();
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
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            // residual rows of this pass: issued first so their latency hides behind the LDS round trip
            f32x4 res[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int rr = it * 4 + (l >> 4);
                res[it] = *reinterpret_cast<const f32x4 *>(G.res32 + (size_t)(mw0 + y * 32 + rr) * G.ldc + nw0 + c4 * 4);
            }
            __syncthreads();
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x16 &a = acc[x][y];
                    *reinterpret_cast<f32x4 *>(slab + i * LS + x * 32 + 8 * rq + 4 * g) =
                        f32x4{a[4 * rq], a[4 * rq + 1], a[4 * rq + 2], a[4 * rq + 3]};
                }
            __syncthreads();
            // read back: 16 lanes cover one row (64 floats), 4 rows per instruction, 8 instructions
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int rr = it * 4 + (l >> 4);
                const f32x4 v = *reinterpret_cast<const f32x4 *>(slab + rr * LS + c4 * 4);
                *reinterpret_cast<f32x4 *>(G.out32 + (size_t)(mw0 + y * 32 + rr) * G.ldc + nw0 + c4 * 4) = v + bias + res[it];
            }
        }
    } else {
        // fp16 outputs: slab [64 m][64 n] halves, row stride 72 halves (144 B); 2 passes
        _Float16 *slab = smem + w * 8192;  // 16 KiB per wave
        constexpr int LS = 72;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __syncthreads();
#pragma unroll
            for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const f32x16 &a = acc[x][2 * p + yy];
                        const int nl = x * 32 + 8 * rq + 4 * g;  // local n of element 0
                        const f32x4 bias = *reinterpret_cast<const f32x4 *>(G.bias + nw0 + nl);
                        f16x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = a[4 * rq + e] + bias[e];
                            if constexpr (EPI == EPI_GELU) t = gelu_erf256(t);
                            else t = (nw0 + nl + e) < G.scale_cols ? t * G.scale : t;
                            v[e] = (_Float16)t;
                        }
                        *reinterpret_cast<f16x4 *>(slab + (yy * 32 + i) * LS + nl) = v;
                    }
            __syncthreads();
            // read back: 8 lanes cover one row (64 halves = 128 B), 8 rows per instruction
            const int c8 = l & 7;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int rr = it * 8 + (l >> 3);
                const f16x8 v = *reinterpret_cast<const f16x8 *>(slab + rr * LS + c8 * 8);
                *reinterpret_cast<f16x8 *>(G.out16 + (size_t)(mw0 + p * 64 + rr) * G.ldc + nw0 + c8 * 8) = v;
            }
        }
    }
}

template <bool GLDS>
int launch256(int epi, const GemmArgs &G, hipStream_t st) {
    const int MT = G.M / TM, NT = G.N / TN;
    const unsigned blocks = MT >= NT ? (unsigned)((MT + 7) / 8 * 8) * (unsigned)NT : (unsigned)((NT + 7) / 8 * 8) * (unsigned)MT;
    void (*k)(const GemmArgs) = nullptr;
    switch (epi) {
        case EPI_QK: k = gemm256_f16_kernel<EPI_QK, GLDS>; break;
        case EPI_GELU: k = gemm256_f16_kernel<EPI_GELU, GLDS>; break;
        case EPI_RES32: k = gemm256_f16_kernel<EPI_RES32, GLDS>; break;
        case EPI_VT: k = gemm256_f16_kernel<EPI_VT, GLDS>; break;
        default: set_last_error("gemm256: bad epilogue"); return ANCE_E_INVALID;
    }
    static bool attr_done[4] = {false, false, false, false};
    if (!attr_done[epi]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)G256_LDS_BYTES) != hipSuccess)
            return check_launch("gemm256 attr");
        attr_done[epi] = true;
    }
    hipLaunchKernelGGL(k, dim3(blocks), dim3(G256_THREADS), G256_LDS_BYTES, st, G);
    return ANCE_OK;
}

}  // namespace

bool gemm256_applicable(const GemmArgs &G) {
    return G.M > 0 && G.N > 0 && G.K > 0 && G.M % TM == 0 && G.N % TN == 0 && G.K % TK == 0;
}

int launch_gemm256_f16(int epi, const GemmArgs &G, bool glds, hipStream_t st) {
    if (!gemm256_applicable(G)) {
        set_last_error("gemm256: M,N must be multiples of 256 and K of 64");
        return ANCE_E_INVALID;
    }
    return glds ? launch256<true>(epi, G, st) : launch256<false>(epi, G, st);
}

}  // namespace ance
