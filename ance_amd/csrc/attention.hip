// Variable-length multi-head self-attention for the encoder (gfx950), fp16 MFMA, fp32 softmax.
//
// One workgroup = one (sequence, head); 4 waves, each owning 32 query rows at a time.  The launch is bound by how many
// (sequence, head) units a CU holds (three, by LDS) while each of them waits out its chain of dependent memory round trips
// (descriptor -> K / V^T / Q rows -> LDS) -- rocprofv3: waves spend 53 % of their cycles in s_waitcnt -- not by
// arithmetic (MFMA pipe 7 % busy).  Pad tokens
// do not exist in the packed layout, so "attention_mask" (data/msmarco_data.py:282) is simply the
// sequence boundary: keys >= len never enter the softmax.
//
// Swapped product S^T = K . Q^T (rows = keys, columns = queries): one lane owns one query column,
// so the running max / sum / rescale are per-lane scalars and the only cross-lane traffic is one
// exchange with lane ^ 32 per key block.  The C-layout of S^T (lane group g holds keys 4g..4g+3,
// 8+4g.. of every 16) is consumed DIRECTLY as the B operand of O^T = V^T . P^T; V arrives already
// transposed (key-contiguous) from the V^T GEMM epilogue, so no transpose is ever performed here.
//
// Measured and rejected (round 3): K and V^T of a head resident in the registers of ONE wave as MFMA fragments loaded
// straight from global memory (no LDS, no barrier, no idle wave): 127 us against 117 -- a fragment load touches 32 rows
// per instruction, 3.0e7 L1 lookups per launch, and 232 registers leave two waves per SIMD.
// Measured and rejected (round 2): a workgroup owning 2-12 consecutive heads of a sequence with the next head's K / V^T / Q
// loads software-pipelined behind the current head's compute (48 more VGPRs: 2 workgroups per CU instead of 4) --
// 126-148 us per launch at the bench shape against 117 us for this kernel: the launch is bound by the dependent
// MFMA -> softmax -> MFMA chain inside each wave, which only residency (waves per SIMD) hides, not by the staging latency.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "attention.h"

namespace ance {
namespace {

constexpr int HD = 64;          // head dim

__device__ __forceinline__ int kswz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// One 32-query block (this wave's) of one (sequence, head) against all keys staged in LDS: online softmax in fp32,
// P in fp16, output rows written to ctx.  qf = the block's Q fragments (B operand layout).
// LDS operations of one wave execute in order; the fence keeps the compiler from moving a lane's read above another
// lane's write (same argument as the GEMM epilogue's wave-private slabs).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// COAL: the block's output goes through the wave's private 4 KiB slab ([32 queries][64 dims], chunk-swizzled like the K
// tile) and leaves as whole 128-byte rows.  Straight from the accumulators a store instruction writes 8 bytes into each of
// 32 different rows: 64 cache-line accesses per instruction, and it is the L1's tag rate -- one lookup per cycle -- that
// bounds this kernel (rocprofv3: 83 k lookups per CU and launch, 7 x what the bytes need).
template <bool COAL>
__device__ __forceinline__ void attend_qblock(const AttnArgs &A, const _Float16 *Ks, const _Float16 *Vs, const f16x8 (&qf)[4], int s,
                                              int h, int tok0, int T, int Tk, int vld, int qb0, int q_end, int g, int i,
                                              _Float16 *slab) {
    const int nkb = Tk >> 5;
    float m_run = -INFINITY, l_run = 0.0f;
    f32x16 o0 = {0}, o1 = {0};
    for (int kb = 0; kb < nkb; ++kb) {
        const int krow = kb * 32 + i;
        const int ksw = (krow >> 1) & 7;
        f32x16 st = {0};
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) {
            const f16x8 kf = *reinterpret_cast<const f16x8 *>(Ks + krow * HD + (((4 * g + sx) ^ ksw) * 8));
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[sx], st, 0, 0, 0);
        }
        // st[r] = score(key kb*32 + (r&3) + 8 (r>>2) + 4 g, query i) in the log2 domain (Q carries log2(e)/8)
        const int key_base = kb * 32 + 4 * g;
        float bm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key_base + (r & 3) + 8 * (r >> 2);
            if (kb == nkb - 1) st[r] = key < T ? st[r] : -INFINITY;  // only the last block holds keys >= T
            bm = fmaxf(bm, st[r]);
        }
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        const float m_new = fmaxf(m_run, bm);   // finite: key 0 of block 0 is always real
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.0f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = __builtin_amdgcn_exp2f(st[r] - m_new);
            psum += p[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0[r] *= alpha;
            o1[r] *= alpha;
        }
        // P^T fragments (B operand of O^T = V^T P^T): k-step u covers keys 16u..16u+15; this lane
        // group owns keys 16u + 4g + {0..3} and 16u + 8 + 4g + {0..3} = registers 8u..8u+7.
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (_Float16)p[8 * u + j];
            const int kc = kb * 32 + 16 * u + 4 * g;
            const _Float16 *v0 = Vs + i * vld + kc;
            const _Float16 *v1 = Vs + (i + 32) * vld + kc;
            const f16x4 a0 = *reinterpret_cast<const f16x4 *>(v0);
            const f16x4 a1 = *reinterpret_cast<const f16x4 *>(v0 + 8);
            const f16x4 c0 = *reinterpret_cast<const f16x4 *>(v1);
            const f16x4 c1 = *reinterpret_cast<const f16x4 *>(v1 + 8);
            const f16x8 vf0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            const f16x8 vf1 = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf0, pf, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf1, pf, o1, 0, 0, 0);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    // O^T[d][query]: d = db*32 + (r&3) + 8 (r>>2) + 4 g  ->  4 consecutive d per (db, r>>2)
    if constexpr (COAL) {
        const int l = g * 32 + i;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const f32x16 &o = db == 0 ? o0 : o1;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f16x4 v = {(_Float16)(o[4 * rq + 0] * inv), (_Float16)(o[4 * rq + 1] * inv),
                                 (_Float16)(o[4 * rq + 2] * inv), (_Float16)(o[4 * rq + 3] * inv)};
                *reinterpret_cast<f16x4 *>(slab + i * HD + kswz(i, db * 4 + rq) * 8 + 4 * g) = v;
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = j * 8 + (l >> 3), ch = l & 7;
            if (qb0 + row < q_end) {
                const size_t orow = A.cls_only ? (size_t)s : (size_t)(tok0 + qb0 + row);
                *reinterpret_cast<f16x8 *>(A.ctx + orow * A.ld_ctx + h * HD + ch * 8) =
                    *reinterpret_cast<const f16x8 *>(slab + row * HD + kswz(row, ch) * 8);
            }
        }
        wave_lds_sync();  // the slab is this wave's next Q block
    } else if (qb0 + i < q_end) {
        const size_t orow = A.cls_only ? (size_t)s : (size_t)(tok0 + qb0 + i);
        _Float16 *op = A.ctx + orow * A.ld_ctx + h * HD + 4 * g;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const f32x16 &o = db == 0 ? o0 : o1;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f16x4 v = {(_Float16)(o[4 * rq + 0] * inv), (_Float16)(o[4 * rq + 1] * inv),
                                 (_Float16)(o[4 * rq + 2] * inv), (_Float16)(o[4 * rq + 3] * inv)};
                *reinterpret_cast<f16x4 *>(op + db * 32 + 8 * rq) = v;
            }
        }
    }
}

// NW waves per workgroup (4 in the product; 32 NW queries per round).  Three workgroups per CU: at 128 tokens the 49 KB
// of LDS allow no more.
template <bool COAL, int NW>
__global__ void __launch_bounds__(64 * NW, 3) attention_kernel(const AttnArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int NT = 64 * NW;
    const int u = blockIdx.x / A.n_heads;
    const int h = blockIdx.x - u * A.n_heads;
    const int4 dsc = A.desc[u];                      // one 16-byte scalar load: (first token, length, V^T column, sequence)
    const int tok0 = dsc.x;
    const int T = dsc.y;                             // 1..max_seq_len
    const int vcol0 = dsc.z;                         // 8-aligned first key column in V^T
    const int s = dsc.w;
    const int Tk = (T + 31) & ~31;                   // keys padded to the MFMA block
    const int vld = Tk + 4;                          // V^T LDS row stride (halves): 8 * odd bytes
    _Float16 *Ks = reinterpret_cast<_Float16 *>(smem_f);  // [Tk][64], chunk-swizzled
    _Float16 *Vs = Ks + (size_t)Tk * HD;                   // [64][vld]

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63, g = l >> 5, i = l & 31;
    const int H = A.n_heads * HD;
    _Float16 *slab = Vs + (size_t)HD * vld + w * (32 * HD);  // COAL: this wave's [32][64] Q / O slab

    // Q of this wave's first query block: requested before the staging so that its latency overlaps the K / V^T loads.
    // COAL: whole 128-byte rows (8 lanes per row, 8 rows per instruction) that go through the slab; otherwise each lane
    // loads its own fragment pieces (B operand: lane (query i, group g) holds head dims 32 g + 8 s .. + 8) -- 64 cache-line
    // accesses per instruction instead of 8.
    const int q_end = A.cls_only ? 1 : T;  // last layer: only the [CLS] query feeds the head
    f16x8 qf[4];
    // (q_compact: the one query of sequence s is row s of the Q columns -- encoder.hip projects the [CLS] rows compactly)
    auto q_row = [&](int r) { return A.q_compact ? (size_t)s : (size_t)(tok0 + min(r, T - 1)); };
    auto q_request = [&](int qb0) {
        if constexpr (COAL) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = j * 8 + (l >> 3);
                qf[j] = *reinterpret_cast<const f16x8 *>(A.qk + q_row(qb0 + row) * A.ld_qk + h * HD + (l & 7) * 8);
            }
        } else {
            const _Float16 *qp = A.qk + q_row(qb0 + i) * A.ld_qk + h * HD + 32 * g;
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) qf[sx] = *reinterpret_cast<const f16x8 *>(qp + sx * 8);
        }
    };
    auto q_to_fragments = [&]() {  // COAL: rows -> slab -> B-operand fragments (wave-private, no workgroup barrier)
        if constexpr (COAL) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = j * 8 + (l >> 3);
                *reinterpret_cast<f16x8 *>(slab + row * HD + kswz(row, l & 7) * 8) = qf[j];
            }
            wave_lds_sync();
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) qf[sx] = *reinterpret_cast<const f16x8 *>(slab + i * HD + kswz(i, 4 * g + sx) * 8);
            wave_lds_sync();  // the slab takes the block's output next
        }
    };
    if (w * 32 < q_end) q_request(w * 32);

    // ---- stage K (rows = keys) and V^T (rows = head dims) into LDS; columns >= T of V^T are zeroed ---
    // 32 NW keys per pass, eight independent 16-byte loads per thread in flight before the first LDS store (a plain
    // load -> store loop waits one full memory latency per iteration).  K: 8 lanes cover a 128-byte row; V^T: a wave
    // reads one 16-byte chunk of each of the 64 rows.  No integer division by the runtime chunk count: this prologue used to cost
    // more VALU instructions than the attention itself, and the launch is bound by VALU issue and L1 lookups.
    {
        // one pass = 32 NW keys = 4 NW chunks of every V^T row: K element e = j NT + tid -> row e / 8, chunk e % 8;
        // V^T element e -> row e / (4 NW), chunk e % (4 NW) (consecutive lanes walk along a row; compile-time divisor)
        const _Float16 *kbase = A.qk + (size_t)tok0 * A.ld_qk + H + h * HD + (tid & 7) * 8;
        const _Float16 *vbase = A.vt + (size_t)(h * HD) * A.ld_vt + vcol0;
        for (int base = 0; base < Tk; base += 32 * NW) {
            f16x8 kv[4], vv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = base + j * (8 * NW) + (tid >> 3);
                const int ev = j * NT + tid, dd = ev / (4 * NW);
                const int key0 = base + (ev - dd * (4 * NW)) * 8;
                kv[j] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                vv[j] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (row < T) kv[j] = *reinterpret_cast<const f16x8 *>(kbase + (size_t)row * A.ld_qk);
                if (key0 < T) vv[j] = *reinterpret_cast<const f16x8 *>(vbase + (size_t)dd * A.ld_vt + key0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = base + j * (8 * NW) + (tid >> 3);
                const int ev = j * NT + tid, dd = ev / (4 * NW);
                const int key0 = base + (ev - dd * (4 * NW)) * 8;
                if (row < Tk) *reinterpret_cast<f16x8 *>(Ks + row * HD + kswz(row, tid & 7) * 8) = kv[j];
                if (key0 < Tk) {
                    if (key0 + 8 > T) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (key0 + e >= T) vv[j][e] = (_Float16)0.0f;
                    }
                    _Float16 *dst = Vs + dd * vld + key0;  // 8-byte aligned
                    *reinterpret_cast<f16x4 *>(dst) = f16x4{vv[j][0], vv[j][1], vv[j][2], vv[j][3]};
                    *reinterpret_cast<f16x4 *>(dst + 4) = f16x4{vv[j][4], vv[j][5], vv[j][6], vv[j][7]};
                }
            }
        }
    }
    if (w * 32 < q_end) q_to_fragments();
    __syncthreads();

    for (int qb0 = w * 32; qb0 < q_end; qb0 += 32 * NW) {
        if (qb0 != w * 32) {  // later query blocks of long sequences (the first one was prefetched above)
            q_request(qb0);
            q_to_fragments();
        }
        attend_qblock<COAL>(A, Ks, Vs, qf, s, h, tok0, T, Tk, vld, qb0, q_end, g, i, slab);
    }
}



// ---- split (fp32-grade) attention: fp16 (hi, lo') pairs on the matrix cores, fp32 softmax ---------------------------------------
// The split encoder mode (encoder.hip) keeps Q | K | V in fp32 (qkv: [T, 2304] rows, as the fp32 path of precise32.h) and wants
// the context rows back as pair rows (common.h: pair_store4).  Same structure as attention_kernel -- one workgroup per (sequence,
// head), K and V^T of the head staged once in LDS, S^T = K Q^T, online softmax per lane, O^T = V^T P^T -- but every operand is
// split while it is staged (v = hi + lo' 2^-11 -- internal to this kernel, two accumulator sets; the split GEMM's pair rows carry
// unscaled lo halves, see common.h) and every product is three MFMAs:
//     S^T = 2^-11 (K_hi Q_lo'^T + K_lo' Q_hi^T) + K_hi Q_hi^T          (Q carries log2(e) / 8: the softmax runs on exp2)
//     O^T = 2^-11 (V_hi P_lo'^T + V_lo' P_hi^T) + V_hi P_hi^T          (two accumulator sets, combined once at the end)
// LDS: 4 x keys x 64 halves = 64 KB at 128 keys, 128 KB at 256; longer sequences stage their keys 256 at a time (the
// online softmax iterates key blocks anyway) and re-stage them for every pass of 128 queries.  Replaces the vector-unit kernel of
// precise32.h (launch_attention32, kept behind ANCE_SPLIT_ATTN=0) at 8 x the speed for the lengths of config 2.
// V in LDS (round 6, TR): ROW-major like K -- [key][64 dims + 8 pad] halves, staged by 16-byte writes -- and read as V^T fragments
// with gfx950's LDS transpose read (ds_read_b64_tr_b16: every lane gives the address of 4 contiguous halves, within each group of
// 16 lanes the 16 x 4 block arrives transposed: out[l][j] = in[16 (l / 16) + 4 j + (l % 16) / 4][l % 4], tools/tr16_probe.cpp).  The
// first form transposed while staging: sixteen 2-byte LDS writes per 8 dims of a key, four-way bank-conflicted -- 44 % of the kernel's
// LDS-array cycles were conflict cycles (profiles/r05_sq_counters_split_gemms.json).  Same fragment values, same MFMAs: bit-identical
// to that form (checked with both compiled in, commit abaa8b9: tests/test_gpu_encoder.py::test_split_attention_transpose_reads_change_no_bit).
// Row stride of V in LDS (halves) and where dims d .. d + 7 of key `key` sit in the row.  ds_read_b64_tr_b16 is serviced in two groups
// of 32 lanes over 64 banks (MI355X_MICROARCH.md, LDS): a group here reads 4 keys x 32 dims = 4 x 16 banks.  With 128-byte rows
// (32 banks) keys k and k + 2 start on the same bank; the first form of this layout padded the rows to 144 bytes (36 banks), which
// still overlaps keys k and k + 2 on half of their banks -- every transpose read cost two LDS cycles per group instead of one
// (0.22 conflict cycles per LDS-active cycle).  Now: unpadded rows, the two 64-byte halves of a row trade places when bit 1 of the key
// is set -- keys k .. k + 3 of one group fall on banks 0-15 | 32-47 | 16-31 | 48-63: SQ_LDS_BANK_CONFLICT = 0 for the whole kernel,
// 22 % fewer LDS-active cycles (profiles/r06_attention_lds_counters.json; the padded form: commit fff5260, ANCE_ATTN_V_PAD).  The
// kernel's time does not move (the LDS wait is 0.8 % of a wave's cycles): it is bound by its HBM traffic, see DESIGN.md 3.6.
constexpr int VS = HD;
__device__ __forceinline__ int vswz(int key, int d) { return d ^ (((key >> 1) & 1) << 5); }
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef s16x4_t __attribute__((address_space(3))) lds_s16x4_t;
__device__ __forceinline__ f16x4 lds_read_tr16(const _Float16 *p) {
    const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(p));
    return __builtin_bit_cast(f16x4, v);
}

// v_permlane32_swap_b32 x, y: lanes 32-63 of x trade places with lanes 0-31 of y (x[l + 32] <-> y[l]).  Inline asm: the builtin of this
// compiler (ROCm 7.2: __builtin_amdgcn_permlane32_swap) returns its first result twice (tools/tr16_probe.cpp checks the instruction).
// The s_nops cover the VALU write -> permlane-swap read and permlane-swap write -> VALU read wait states the hazard recognizer cannot
// see through an asm statement.
__device__ __forceinline__ void permlane32_swap(float &x, float &y) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}

template <int NW>
__global__ void __launch_bounds__(64 * NW, 2) attention_split_kernel(const float *qkv, _Float16 *ctx_pair, const int4 *desc, int n_heads,
                                                                     int cls_only, int kchunk) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int u = blockIdx.x / n_heads;
    const int h = blockIdx.x - u * n_heads;
    const int4 dsc = desc[u];
    const int tok0 = dsc.x, T = dsc.y, s = dsc.w;
    const int Tk = (T + 31) & ~31;
    // keys are staged kc keys at a time (kchunk = the launch's LDS budget, a multiple of 32): all of them at once for the
    // sequences that fit (every sequence of config 2), 256 at a time for longer ones -- the online softmax does not care
    const int kc = Tk < kchunk ? Tk : kchunk;
    _Float16 *Kh = reinterpret_cast<_Float16 *>(smem_f);  // [kc][64] chunk-swizzled
    _Float16 *Kl = Kh + (size_t)kc * HD;
    _Float16 *Vh = Kl + (size_t)kc * HD;                  // [kc][VS] row-major
    _Float16 *Vl = Vh + (size_t)kc * VS;
    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63, g = l >> 5, i = l & 31;
    constexpr int NT = 64 * NW;
    const int ld = 3 * n_heads * HD;
    constexpr float SC = 2048.0f, SI = 1.0f / 2048.0f;
    // ---- stage K (rows = keys) and V^T (rows = head dims) of keys [k0, k0 + kc), split on the way; keys >= T are zeros ----
    // UNR rounds of loads are issued before the first one is consumed: a sequence's K / V panel then costs one HBM latency, not
    // one per round (4 rounds at 128 keys).  The chunked path (T > 256) stages under live accumulators and keeps UNR = 1.
    auto stage = [&](int k0, auto unr_tag) {
        constexpr int UNR = decltype(unr_tag)::value;
#if defined(ATTN_DIAG_NO_STAGE)  // measurement builds only (scripts/gpu_r6_attn_phases.sh): what the kernel costs without one of its phases
        if (n_heads > 0) {  // no panel loads, no split: the LDS writes only (zeros)
            for (int e = tid; e < kc * 32; e += NT) *reinterpret_cast<f16x8 *>(Kh + e * 8) = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            return;
        }
#endif
        for (int e0 = tid; e0 < kc * 8; e0 += UNR * NT) {
            f32x4 kv[UNR][4];
#pragma unroll
            for (int it = 0; it < UNR; ++it) {
                const int e = e0 + it * NT;
                const int kl_ = e >> 3, ch = e & 7, key = k0 + kl_;
#pragma unroll
                for (int q = 0; q < 4; ++q) kv[it][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#if defined(ATTN_DIAG_NO_LOAD)  // the split and the LDS writes, without the panel's global loads
                if (n_heads > 0) {
                    for (int q = 0; q < 4; ++q) kv[it][q] = f32x4{0.01f * ch, 0.02f, -0.01f * kl_, 0.005f};
                } else
#endif
                if (e < kc * 8 && key < T) {
                    const float *kp = qkv + (size_t)(tok0 + key) * ld + n_heads * HD + h * HD + ch * 8;
                    kv[it][0] = *reinterpret_cast<const f32x4 *>(kp);
                    kv[it][1] = *reinterpret_cast<const f32x4 *>(kp + 4);
                    kv[it][2] = *reinterpret_cast<const f32x4 *>(kp + n_heads * HD);
                    kv[it][3] = *reinterpret_cast<const f32x4 *>(kp + n_heads * HD + 4);
                }
            }
#pragma unroll
            for (int it = 0; it < UNR; ++it) {
                const int e = e0 + it * NT;
                if (e >= kc * 8) break;
                const int kl_ = e >> 3, ch = e & 7;
                const f32x4 k0v = kv[it][0], k1v = kv[it][1], v0 = kv[it][2], v1 = kv[it][3];
                const f16x4 a0 = cvt_f16x4_pinned(k0v), a1 = cvt_f16x4_pinned(k1v), b0 = cvt_f16x4_pinned(v0), b1 = cvt_f16x4_pinned(v1);
                f16x8 kh, kl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    kh[j] = a0[j]; kh[4 + j] = a1[j];
                    kl[j] = (_Float16)((k0v[j] - (float)a0[j]) * SC);
                    kl[4 + j] = (_Float16)((k1v[j] - (float)a1[j]) * SC);
                }
                *reinterpret_cast<f16x8 *>(Kh + kl_ * HD + kswz(kl_, ch) * 8) = kh;
                *reinterpret_cast<f16x8 *>(Kl + kl_ * HD + kswz(kl_, ch) * 8) = kl;
                f16x8 vh, vl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vh[j] = b0[j]; vh[4 + j] = b1[j];
                    vl[j] = (_Float16)((v0[j] - (float)b0[j]) * SC);
                    vl[4 + j] = (_Float16)((v1[j] - (float)b1[j]) * SC);
                }
                *reinterpret_cast<f16x8 *>(Vh + kl_ * VS + vswz(kl_, ch * 8)) = vh;
                *reinterpret_cast<f16x8 *>(Vl + kl_ * VS + vswz(kl_, ch * 8)) = vl;
            }
        }
    };
    const bool single = Tk <= kc;  // one chunk: staged once (inside the first query pass, behind its Q loads), shared by every pass
    const int q_end = cls_only ? 1 : T;
    const float qscale = 0.125f * 1.44269504088896340736f;
    for (int qp0 = 0; qp0 < q_end; qp0 += 32 * NW) {  // query pass: 32 queries per wave (uniform loop: the barriers below)
        const int qb0 = qp0 + w * 32;
        const bool active = qb0 < q_end;
        // Q fragments (B operand: lane (query i, group g) holds head dims 32 g + 8 sx .. + 8), scaled, then split
        f16x8 qh[4], ql[4];
        {
            // (cls_only: the one query of sequence s is row s of the Q columns -- encoder.hip projects the [CLS] rows compactly)
            const float *qp = qkv + (cls_only ? (size_t)s : (size_t)(tok0 + min(qb0 + i, T - 1))) * ld + h * HD + 32 * g;
            f32x4 xq[8];
#pragma unroll
            for (int sx = 0; sx < 8; ++sx) xq[sx] = *reinterpret_cast<const f32x4 *>(qp + sx * 4);
            if (single && qp0 == 0) {  // uniform; the Q loads above are in flight while the panel is fetched
                stage(0, std::integral_constant<int, 4>{});
                __syncthreads();
            }
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                f32x4 x0 = xq[2 * sx], x1 = xq[2 * sx + 1];
                x0 = x0 * qscale;
                x1 = x1 * qscale;
                const f16x4 a0 = cvt_f16x4_pinned(x0), a1 = cvt_f16x4_pinned(x1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    qh[sx][j] = a0[j]; qh[sx][4 + j] = a1[j];
                    ql[sx][j] = (_Float16)((x0[j] - (float)a0[j]) * SC);
                    ql[sx][4 + j] = (_Float16)((x1[j] - (float)a1[j]) * SC);
                }
            }
        }
        float m_run = -INFINITY, l_run = 0.0f;
        f32x16 om0 = {0}, om1 = {0}, oc0 = {0}, oc1 = {0};
        for (int k0 = 0; k0 < Tk; k0 += kc) {
            if (!single) {
                __syncthreads();  // the previous chunk (or pass) is no longer read
                stage(k0, std::integral_constant<int, 1>{});
                __syncthreads();
            }
            if (!active) continue;
#if defined(ATTN_DIAG_NO_COMPUTE)
            const int nkb = n_heads > 0 ? 0 : 1;
#else
            const int nkb = min(kc, Tk - k0) >> 5;
#endif
            for (int kb = 0; kb < nkb; ++kb) {
                const int krow = kb * 32 + i;
                const int ksw = (krow >> 1) & 7;
                f32x16 st = {0};
                f16x8 kfh[4];
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) {
                    kfh[sx] = *reinterpret_cast<const f16x8 *>(Kh + krow * HD + (((4 * g + sx) ^ ksw) * 8));
                    const f16x8 kfl = *reinterpret_cast<const f16x8 *>(Kl + krow * HD + (((4 * g + sx) ^ ksw) * 8));
                    st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[sx], ql[sx], st, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl, qh[sx], st, 0, 0, 0);
                }
                st *= SI;
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[sx], qh[sx], st, 0, 0, 0);
                const int key_base = k0 + kb * 32 + 4 * g;
                float bm = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key_base + (r & 3) + 8 * (r >> 2);
                    if (k0 + kb * 32 + 32 > T) st[r] = key < T ? st[r] : -INFINITY;  // only the last block holds keys >= T
                    bm = fmaxf(bm, st[r]);
                }
                bm = fmaxf(bm, __shfl_xor(bm, 32));
                const float m_new = fmaxf(m_run, bm);   // finite: key 0 of block 0 is always real
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                float psum = 0.0f;
                float p[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(st[r] - m_new);
                    psum += p[r];
                }
                l_run = l_run * alpha + psum;
                m_run = m_new;
                om0 *= alpha; om1 *= alpha; oc0 *= alpha; oc1 *= alpha;
#pragma unroll
                for (int uu = 0; uu < 2; ++uu) {
                    const f16x4 h0 = cvt_f16x4_pinned(f32x4{p[8 * uu], p[8 * uu + 1], p[8 * uu + 2], p[8 * uu + 3]});
                    const f16x4 h1 = cvt_f16x4_pinned(f32x4{p[8 * uu + 4], p[8 * uu + 5], p[8 * uu + 6], p[8 * uu + 7]});
                    f16x8 ph, pl;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        ph[j] = h0[j]; ph[4 + j] = h1[j];
                        pl[j] = (_Float16)((p[8 * uu + j] - (float)h0[j]) * SC);
                        pl[4 + j] = (_Float16)((p[8 * uu + 4 + j] - (float)h1[j]) * SC);
                    }
                    const int kcol = kb * 32 + 16 * uu + 4 * g;
                    // V^T fragment of head dims row .. (lane i), keys kcol + {0..3, 8..11}
                    auto vfrag = [&](const _Float16 *V, int row) {
                        // lane q of a 16-lane group points at key kcol + q / 4, dims (row - i) + 16 (lane / 16 % 2) + 4 (q % 4) .. + 3
                        // (keys key and key + 8 share bit 1: one swizzle for both reads)
                        const int q = l & 15, key = kcol + (q >> 2);
                        const _Float16 *vp = V + key * VS + vswz(key, (row - i) + 16 * ((l >> 4) & 1) + 4 * (q & 3));
                        const f16x4 a = lds_read_tr16(vp), b = lds_read_tr16(vp + 8 * VS);
                        return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                    };
                    const f16x8 v0h = vfrag(Vh, i), v1h = vfrag(Vh, i + 32), v0l = vfrag(Vl, i), v1l = vfrag(Vl, i + 32);
                    oc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, pl, oc0, 0, 0, 0);
                    oc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0l, ph, oc0, 0, 0, 0);
                    oc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, pl, oc1, 0, 0, 0);
                    oc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1l, ph, oc1, 0, 0, 0);
                    om0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, ph, om0, 0, 0, 0);
                    om1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, ph, om1, 0, 0, 0);
                }
            }
        }
        if (!active) continue;
#if defined(ATTN_DIAG_NO_COMPUTE)
        l_run = 1.0f;
#endif
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = 1.0f / l_tot;
        {
            // C-layout: a lane owns 4 consecutive head dims per register quad (8 rq + 4 g + j); lanes l and l + 32 (g = 0 / 1) own the two
            // halves of every 8-dim group.  v_permlane32_swap trades them so that lane g = 0 ends with dims 16 k .. 16 k + 7 and lane
            // g = 1 with 16 k + 8 .. + 15: 16-byte hi and lo stores instead of 8-byte ones -- half as many vector-memory instructions
            // (all 64 lanes take part in the swaps; rows past the sequence's end are simply not stored).
            const bool store = qb0 + i < q_end;
            const size_t orow = cls_only ? (size_t)s : (size_t)(tok0 + min(qb0 + i, T - 1));
            _Float16 *orp = ctx_pair + orow * (size_t)(2 * n_heads * HD);  // pair row (common.h) of n_heads * 64 columns
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const f32x16 &om = db == 0 ? om0 : om1;
                const f32x16 &oc = db == 0 ? oc0 : oc1;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    f32x4 va, vb;  // register quads rq = 2 k2 and 2 k2 + 1
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xa = __builtin_fmaf(oc[8 * k2 + j], SI, om[8 * k2 + j]) * inv;
                        const float xb = __builtin_fmaf(oc[8 * k2 + 4 + j], SI, om[8 * k2 + 4 + j]) * inv;
                        // after the swap -- lanes < 32: (own quad 2 k2, partner's quad 2 k2); lanes >= 32: (partner's quad 2 k2 + 1, own quad 2 k2 + 1)
                        float sa = xa, sb = xb;
                        permlane32_swap(sa, sb);
                        va[j] = sa;
                        vb[j] = sb;
                    }
                    if (store) {
                        f16x4 ha, ra, hb, rb;
                        pair_split4(va, &ha, &ra);
                        pair_split4(vb, &hb, &rb);
                        const int n = h * HD + db * 32 + 16 * k2 + 8 * g;
                        *reinterpret_cast<f16x8 *>(orp + pair_hi_col(n, n_heads * HD)) = f16x8{ha[0], ha[1], ha[2], ha[3], hb[0], hb[1], hb[2], hb[3]};
                        *reinterpret_cast<f16x8 *>(orp + pair_lo_col(n, n_heads * HD)) = f16x8{ra[0], ra[1], ra[2], ra[3], rb[0], rb[1], rb[2], rb[3]};
                    }
                }
            }
        }
    }
}

}  // namespace

size_t attention_split_lds_bytes(int max_seq_len) {
    const int Tk = (max_seq_len + 31) & ~31;
    return (size_t)2 * Tk * HD * 2 + (size_t)2 * Tk * VS * 2;  // K hi | lo' [Tk][64], V hi | lo' [Tk][VS]
}

// qkv [T, 3 n_heads 64] fp32 -> ctx_pair [T or n_seq, 2 n_heads 64] fp16 pair rows; any sequence length (keys staged 256 at a time).
// Measured and rejected (round 4): the sequences of <= 64 tokens in a second launch of 2-wave / 33 KB workgroups (4 per CU, no idle
// waves): 179.6 + 66.3 us against 242.1 us in one launch.
int launch_attention_split(const float *qkv, _Float16 *ctx_pair, const int4 *desc, int n_seq, int n_heads, int max_seq_len,
                           int cls_only, hipStream_t st) {
    if (n_seq <= 0) return ANCE_OK;
    const int Tk = (max_seq_len + 31) & ~31;
    const int kchunk = Tk <= 256 ? Tk : 256;  // keys staged at a time: 64 KB at 128 (two workgroups per CU), 128 KB at 256
    const size_t lds = attention_split_lds_bytes(kchunk);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static size_t attr_set[64] = {0};
    const bool tracked = dev >= 0 && dev < 64;
    if (!tracked || lds > __atomic_load_n(&attr_set[dev], __ATOMIC_ACQUIRE)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(attention_split_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return check_launch("attention_split attr");
        if (tracked) __atomic_store_n(&attr_set[dev], lds, __ATOMIC_RELEASE);
    }
    hipLaunchKernelGGL((attention_split_kernel<4>), dim3((unsigned)n_seq * n_heads), dim3(256), lds, st, qkv, ctx_pair, desc, n_heads,
                       cls_only, kchunk);
    return ANCE_OK;
}

size_t attention_lds_bytes(int max_seq_len, int n_waves) {
    const int Tk = (max_seq_len + 31) & ~31;
    return (size_t)Tk * HD * 2 + (size_t)HD * (Tk + 4) * 2 + (size_t)n_waves * 32 * HD * 2;  // K, V^T, wave-private Q / O slabs
}

// One launch for every length (n_seq descriptors in args.desc, longest sequences first: the last workgroups to start
// are the short ones).  Measured and rejected (round 3): one launch per length bucket with ceil(T / 32) waves and the LDS
// the bucket needs (12 / 6 / 4 / 3 workgroups per CU instead of 3) -- 90 us of kernel time against 97, eaten by the
// ramp and tail of four launches (112 us).
int launch_attention(const AttnArgs &A, int n_seq, int max_seq_len, hipStream_t st) {
    if (n_seq <= 0) return ANCE_OK;
    const bool coal = A.coalesced != 0;
    const size_t lds = attention_lds_bytes(max_seq_len, 4);
    if (lds > 160 * 1024) {
        set_last_error("attention: sequence too long for LDS");
        return ANCE_E_INVALID;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    static size_t attr_set[64] = {0};  // the attribute is per device (ordinals >= 64: set on every call)
    const bool tracked = dev >= 0 && dev < 64;
    if (!tracked || lds > __atomic_load_n(&attr_set[dev], __ATOMIC_ACQUIRE)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(attention_kernel<true, 4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(attention_kernel<false, 4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return check_launch("attention attr");
        if (tracked) __atomic_store_n(&attr_set[dev], lds, __ATOMIC_RELEASE);
    }
    if (coal) hipLaunchKernelGGL((attention_kernel<true, 4>), dim3((unsigned)n_seq * A.n_heads), dim3(256), lds, st, A);
    else hipLaunchKernelGGL((attention_kernel<false, 4>), dim3((unsigned)n_seq * A.n_heads), dim3(256), lds, st, A);
    return ANCE_OK;
}

}  // namespace ance
