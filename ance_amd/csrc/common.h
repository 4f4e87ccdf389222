// Shared device/host helpers for the ance_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "../../include/ance_amd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64;

namespace ance {

void set_last_error(const char *msg);
int check_launch(const char *what);

// ---- optional per-kernel timing with HIP events on the launch stream (bench.py's roofline) ----
enum ProfCat {
    PC_PLAN = 0, PC_EMBED, PC_GEMM_QK, PC_GEMM_VT, PC_ATTN, PC_GEMM_OUT, PC_LN, PC_GEMM_FFN1, PC_GEMM_FFN2, PC_HEAD,
    PC_SCAN, PC_FINALIZE, PC_RESCORE, PC_COUNT
};
bool prof_enabled();
void prof_begin(int cat, hipStream_t st, double work);
void prof_end(hipStream_t st);
struct ProfScope {  // brackets the launches made while it is alive (one category)
    hipStream_t st;
    bool on;
    ProfScope(int cat, hipStream_t s, double work = 0.0) : st(s), on(prof_enabled()) {
        if (on) prof_begin(cat, st, work);
    }
    ~ProfScope() {
        if (on) prof_end(st);
    }
};

// ---- order-preserving packing of (score, row) into one u64 key -------------------------------
// larger key  <=>  ranks earlier in the canonical order (score desc, row asc).
// key 0 is the "empty" sentinel: every non-NaN score maps to a high word >= 0x007FFFFF.
__host__ __device__ inline uint32_t order_f32(float s) {
    s = s + 0.0f;  // -0.0 -> +0.0 so the two zeros compare equal
    uint32_t u = __builtin_bit_cast(uint32_t, s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float unorder_f32(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __builtin_bit_cast(float, u);
}
__host__ __device__ inline u64 pack_key(float s, uint32_t row) {
    return ((u64)order_f32(s) << 32) | (u64)(0xFFFFFFFFu - row);
}
__host__ __device__ inline float key_score(u64 k) { return unorder_f32((uint32_t)(k >> 32)); }
__host__ __device__ inline uint32_t key_row(u64 k) { return 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull); }

// fp32 x4 -> fp16 x4 (round to nearest even) with the result PINNED in its registers.  An (hi, lo) pair is only as good as the
// agreement between the hi that is stored and the hi that  lo = fp16(v - hi)  was formed from.  Without the barrier hipcc is
// free to convert v twice -- a packed conversion feeding the 8-byte store and a scalar one feeding the subtraction -- and on
// gfx950 the two disagree on exact ties: the stored pair is then off by one fp16 ulp of v (found by tests/test_gpu_gemm.py::
// test_split_gemm_pair_epilogues on 43 of 1.5 M GELU outputs; it cost the split mode a factor 10 of its accuracy).
__device__ __forceinline__ f16x4 cvt_f16x4_pinned(const f32x4 v) {
    f16x4 h = f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    asm volatile("" : "+v"(h));
    return h;
}

// ---- fp16 (hi, lo) pair rows of the split (fp32-grade) encoder mode ---------------------------------------------------------
// A W-wide fp32 row v travels as 2 W halves, v = hi + lo PAIR_LO_INV with hi = fp16(v), lo = fp16((v - hi) PAIR_LO_SCALE)
// (v - hi is exact in fp32).  Round 5 layout ("blocked"): 32-column blocks [hi (32) | lo (32)], lo UNSCALED -- a 64-half LDS row
// of the GEMM's K-tile then holds a 32-deep k-slice of BOTH halves of an operand row, 128 contiguous bytes in memory, and the
// three products hi x hi, lo x hi, hi x lo of a k-step accumulate into ONE fp32 accumulator (same scale): each operand tile is
// staged once and read from LDS once for three MFMAs (gemm256_f16.hip: gemm256_split_kernel).  Unscaled lo halves of small
// elements are fp16 subnormals (kept by the conversion and by the MFMA: tests/test_gpu_gemm.py::test_mfma_keeps_f16_subnormals);
// below 2^-24 they lose at most 2^-25 ABSOLUTE, which is why weights are stored times a per-matrix power of two that puts
// their largest element in [2^13, 2^14) (encoder.hip: split_weight_kernel; undone in the GEMM epilogue).
constexpr float PAIR_LO_SCALE = 1.0f, PAIR_LO_INV = 1.0f;
__host__ __device__ __forceinline__ int pair_hi_col(int n, int W) { (void)W; return ((n >> 5) << 6) + (n & 31); }
__host__ __device__ __forceinline__ int pair_lo_col(int n, int W) { (void)W; return ((n >> 5) << 6) + (n & 31) + 32; }
// v (4 consecutive columns) -> the two f16x4 of its pair; the hi that is stored is the hi the residual was formed from
__device__ __forceinline__ void pair_split4(const f32x4 v, f16x4 *hi, f16x4 *lo) {
#pragma clang fp contract(off)
    const f16x4 h = cvt_f16x4_pinned(v);
    *hi = h;
    *lo = f16x4{(_Float16)((v[0] - (float)h[0]) * PAIR_LO_SCALE), (_Float16)((v[1] - (float)h[1]) * PAIR_LO_SCALE),
                (_Float16)((v[2] - (float)h[2]) * PAIR_LO_SCALE), (_Float16)((v[3] - (float)h[3]) * PAIR_LO_SCALE)};
}
// store / load columns n .. n + 3 (n a multiple of 4) of a pair row of width W
__device__ __forceinline__ void pair_store4(const f32x4 v, _Float16 *row, int W, int n) {
    f16x4 h, r;
    pair_split4(v, &h, &r);
    *reinterpret_cast<f16x4 *>(row + pair_hi_col(n, W)) = h;
    *reinterpret_cast<f16x4 *>(row + pair_lo_col(n, W)) = r;
}
__device__ __forceinline__ f32x4 pair_load4(const _Float16 *row, int W, int n) {
    const f16x4 h = *reinterpret_cast<const f16x4 *>(row + pair_hi_col(n, W)), r = *reinterpret_cast<const f16x4 *>(row + pair_lo_col(n, W));
    return f32x4{(float)h[0] + (float)r[0] * PAIR_LO_INV, (float)h[1] + (float)r[1] * PAIR_LO_INV, (float)h[2] + (float)r[2] * PAIR_LO_INV,
                 (float)h[3] + (float)r[3] * PAIR_LO_INV};
}

// ---- range guard of the split mode (include/ance_amd.h: ance_encoder_range_faults) -------------------------------------------
// The hi half of a pair is an fp16: a value above 65,504 in magnitude overflows it, where the reference's fp32 would carry on.  Every
// pair-forming stage folds the magnitudes of what it stores into one running maximum per thread (v_max3_f32 with |.| source
// modifiers: half an instruction per element; hipcc's fmaxf canonicalises its operands first -- twice the count) and reports once per
// wave when the kernel ends.  A NaN operand is dropped by the maximum (IEEE mode returns the other operand): NaN rows are counted where
// the output rows are written (encoder.hip: head kernels); an overflow is always seen here first -- it is finite or infinite before
// anything turns into a NaN.
constexpr float RANGE_LIMIT = 65504.0f;
__device__ __forceinline__ void range_track4(const f32x4 v, float *mx) {
    float m = *mx;
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(v[0]), "v"(v[1]));
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(v[2]), "v"(v[3]));
    *mx = m;
}
// faults[0] += the number of lanes of this wave whose running maximum left the range (one atomic per wave, none in the normal case)
__device__ __forceinline__ void range_report(float mx, unsigned *faults) {
    const u64 bad = __builtin_amdgcn_ballot_w64(!(mx <= RANGE_LIMIT));
    if (bad != 0 && faults != nullptr && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(bad)) atomicAdd(faults, (unsigned)__builtin_popcountll(bad));
}

__device__ inline int lane_id() { return (int)(threadIdx.x & 63); }

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// hipFuncSetAttribute is per device: one bit per device ordinal in a mask that is read and updated atomically (two host
// threads may drive two GPUs).  attr_needed: the caller has to set its attributes for the current device; attr_mark: they are
// set -- called only AFTER hipFuncSetAttribute has succeeded, so a failed attempt is retried by the next call.  Ordinals above
// 62 have no bit: their attributes are set on every call.
inline int attr_device_bit() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return (dev < 0 || dev >= 63) ? -1 : dev;
}
inline bool attr_needed(unsigned long long *done) {
    const int bit = attr_device_bit();
    return bit < 0 || !(__atomic_load_n(done, __ATOMIC_ACQUIRE) & (1ull << bit));
}
inline void attr_mark(unsigned long long *done) {
    const int bit = attr_device_bit();
    if (bit >= 0) (void)__atomic_fetch_or(done, 1ull << bit, __ATOMIC_RELEASE);
}

}  // namespace ance
