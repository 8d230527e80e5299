// frame_kernels.hip -- per-frame kernels of the proposal-evaluation path.
//
//  K1  sp_moments_kernel   pixels -> per-superpixel int64 moment sums   (HBM-bound)
//  K1b local_fit_kernel    moments -> centred fit record + theta_i + A1 filter
//  K2  edge_prob_kernel    per CSR entry q_ij
//      segment_prob_kernel Segment(superPixels) for an explicit member list
//
// Reference behaviour restated (not code): OverSegmentation.cpp:145-162,198-222,
// MyMath.h:120-170,189-275,284-296, Graph.cpp:69-138, Problem.cpp:372-445.
#include "device_math.hpp"
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace mc {

// ---------------------------------------------------------------------------
// K1: moment accumulation.
//
// HBM layout: disp int16[F][H][W], label uint16[F][H][W]; 4 bytes per pixel are
// read exactly once with 16-byte loads (8 pixels per lane per plane).  The
// measured kernel is sp_moments_lr_kernel below; Slot is the plain accumulator
// of the generic fallback (any width).  All sums are integers, so the result is
// bit-identical for any band count, block size or arrival order.
// ---------------------------------------------------------------------------
struct Slot {
    uint32_t lab;
    int32_t n, su, sv, sb;
    int64_t suu, suv, svv, sub, svb, sbb;
};

MC_D void slot_reset(Slot& s, uint32_t lab)
{
    s.lab = lab; s.n = 0; s.su = 0; s.sv = 0; s.sb = 0;
    s.suu = 0; s.suv = 0; s.svv = 0; s.sub = 0; s.svb = 0; s.sbb = 0;
}

template <bool USE_LDS>
MC_D void slot_flush(const Slot& s, unsigned long long* tab, unsigned long long* gmom, int S)
{
    if (s.n == 0 || s.lab >= (uint32_t)S) return;  // never address outside the S x 10 table
    unsigned long long* t = (USE_LDS ? tab : gmom) + (size_t)s.lab * kMomStride;
    atomicAdd(t + 0, (unsigned long long)(long long)s.n);
    atomicAdd(t + 1, (unsigned long long)(long long)s.su);
    atomicAdd(t + 2, (unsigned long long)(long long)s.sv);
    atomicAdd(t + 3, (unsigned long long)s.suu);
    atomicAdd(t + 4, (unsigned long long)s.suv);
    atomicAdd(t + 5, (unsigned long long)s.svv);
    atomicAdd(t + 6, (unsigned long long)(long long)s.sb);
    atomicAdd(t + 7, (unsigned long long)s.sub);
    atomicAdd(t + 8, (unsigned long long)s.svb);
    atomicAdd(t + 9, (unsigned long long)s.sbb);
}

MC_D void slot_add1(Slot& s, int u, int v, int b)
{
    s.n += 1;
    s.su += u;
    s.sv += v;
    s.suu += (int64_t)(u * u);
    s.suv += (int64_t)(u * v);
    s.svv += (int64_t)(v * v);
    s.sb += b;
    s.sub += (int64_t)(u * b);
    s.svb += (int64_t)(v * b);
    s.sbb += (int64_t)(b * b);
}

// packed 16-bit helpers: two pixels per dword, exactly as they sit in HBM
typedef short s16x2 __attribute__((ext_vector_type(2)));
MC_D uint32_t pk_mul_lo_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
MC_D int sdot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}
constexpr uint32_t kOnes = 0x00010001u;
// unsigned saturating packed subtract: max(a - b, 0) per 16-bit half
MC_D uint32_t pk_sub_u16_sat(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }
// (1,1) in the halves of lw that equal the label replicated in pat: 1 - (lw ^ pat) saturates to 0 unless the
// half is 0 (two instructions; the min / subtract form takes three)
MC_D uint32_t match01(uint32_t lw, uint32_t pat) { return pk_sub_u16_sat(kOnes, lw ^ pat); }

// ---------------------------------------------------------------------------
// K1, left/right form (the measured path).
//
// A lane owns one 8-pixel-wide column strip of a sub-band of at most 128 rows and
// walks down it.  Two register accumulators: L follows the label of the strip's
// LEFT pixel (pixel 0 of the chunk), R the label of its RIGHT pixel (pixel 7) when
// that differs.  A superpixel boundary that wanders through the strip therefore
// never causes a slot shuffle: L and R are only replaced when the left / right
// label itself changes (about once per 20 rows).  The rare pixels that belong to
// neither (a third label inside 8 pixels, 2 % of chunks) are not handled in the
// hot loop at all: the chunk's position goes to a small per-wave queue that is
// worked off 64 chunks at a time, one chunk per lane.
//
// Accumulators are strip-local and 32 bit (k = column 0..7, r = row 0..127 in the
// sub-band): n, sum k, sum k^2, sum r n, sum r^2 n, sum r k, sum b, sum k b,
// sum r b; only sum b^2 needs 64 bits.  Everything a pixel contributes is a
// packed 16-bit multiply or a v_dot2 on two pixels per dword, exactly as they sit
// in HBM.  Replaced accumulators go to a per-wave LDS log with plain 16-byte
// stores; the log is applied up to 64 records at a time (kRecCap), one record per
// lane, which is where strip-local sums become the int64 frame moments: the
// accumulators keep RUNNING sums, so a record is added to the label it leaves
// (10 ds_add_u64 into the workgroup's table) and subtracted from the label it
// takes up (10 ds_sub_u64); the records that close a strip are added only.
// ---------------------------------------------------------------------------
#ifndef MC_K1_RECCAP
#define MC_K1_RECCAP 64
#endif
constexpr int kRecCap = MC_K1_RECCAP;        // records per wave log (one eviction per lane always fits)
constexpr int kRecDwords = 12;     // 48 B per record
constexpr int kQueueCap = 128;     // queued chunk positions per wave
// row stride of the moment table in 64-bit words: 10 in HBM; 11 in LDS, where a stride of 10 words puts the same field of
// all rows on half of the banks (ds_add_u64 on distinct rows: 9.0 cycles per wave-instruction at stride 10, 7.4 at an odd one)
constexpr int kTabStrideLds = 11;
constexpr uint32_t kFlushL = 0xFFFEFFFFu, kFlushR = 0xFFFDFFFFu;   // never equal to a 16-bit label; low half 0xFFFF >= S

// Running sums of one accumulator since the START of the strip: they are never restarted.  When the label an
// accumulator follows changes from X to Y at row t, the sums S(t) so far are logged once with both labels and the
// table gets  T[X] += S(t),  T[Y] -= S(t)  (the conversion to frame moments is linear): X has then received
// S(t) - S(t_X), the sums of exactly its rows, without a lane-dependent reset of twelve registers in the hot loop.
struct Acc {
    uint32_t lab;
    int32_t n, k1, k2, rn, rrn, rk, b0, bk, rb;
    uint32_t bb32;              // sum b^2 of the rows whose disparities all fit 11 bits: < 2^32 over a whole strip
                                // (the squares of the other rows are added to the table by the queue path)
};

MC_D void acc_reset(Acc& s, uint32_t lab)
{
    s.lab = lab; s.n = 0; s.k1 = 0; s.k2 = 0; s.rn = 0; s.rrn = 0; s.rk = 0; s.b0 = 0; s.bk = 0; s.rb = 0;
    s.bb32 = 0;
}

MC_D int mad24(int a, int b, int c) { return __mul24(a, b) + c; }

MC_D int udot4(uint32_t a, uint32_t b, int c) { return (int)__builtin_amdgcn_udot4(a, b, (uint32_t)c, false); }
MC_D int udot4z(uint32_t a, uint32_t b)
{
    int d;
    asm("v_dot4_u32_u8 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
MC_D int sdot2z(uint32_t a, uint32_t b)   // dot2 with a zero addend, no register to clear
{
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// One half (8 pixels, columns KB .. KB+7 of the lane's chunk) of a row: the pixels selected by m[] (pairs of 0/1),
// bw[] = their disparities (0 elsewhere).  Count, sum k and sum b are chained through cnt / kr / br (the row's
// totals feed the row-weighted sums once per row); sum k^2, sum k b and sum b^2 go straight to the accumulator.
// Squares: four dot products accumulate them in 32 bits (1024 pixels x 2047^2 < 2^32); bsq[] is bw[] on rows whose
// disparities all fit 11 bits and 0 on the others (wave-uniform), whose squares the queue path adds.
template <int KB>
MC_D void acc_add_half(Acc& s, const uint32_t m[4], const uint32_t bw[4], bool squares, int& cnt, int& kr, int& br)
{
    constexpr uint32_t K = (uint32_t)KB;
    constexpr uint32_t w01 = (K) | ((K + 1) << 8) | ((K + 2) << 16) | ((K + 3) << 24);
    constexpr uint32_t w23 = (K + 4) | ((K + 5) << 8) | ((K + 6) << 16) | ((K + 7) << 24);
    constexpr uint32_t q01 = (K * K) | (((K + 1) * (K + 1)) << 8) | (((K + 2) * (K + 2)) << 16) | (((K + 3) * (K + 3)) << 24);
    constexpr uint32_t q23 = ((K + 4) * (K + 4)) | (((K + 5) * (K + 5)) << 8) | (((K + 6) * (K + 6)) << 16) | (((K + 7) * (K + 7)) << 24);
    const uint32_t b0 = bw[0], b1 = bw[1], b2 = bw[2], b3 = bw[3];
    // the eight 0/1 flags as bytes of two dwords: count, sum k and sum k^2 are three v_dot4 pairs
    const uint32_t p01 = __builtin_amdgcn_perm(m[1], m[0], 0x06040200u);
    const uint32_t p23 = __builtin_amdgcn_perm(m[3], m[2], 0x06040200u);
    cnt = udot4(p23, 0x01010101u, udot4(p01, 0x01010101u, cnt));
    kr = udot4(p23, w23, udot4(p01, w01, kr));
    s.k2 = udot4(p23, q23, udot4(p01, q01, s.k2));
    if constexpr (KB == 0) br = sdot2z(b0, kOnes); else br = sdot2(b0, kOnes, br);   // the first half starts the row's totals
    br = sdot2(b1, kOnes, br); br = sdot2(b2, kOnes, br); br = sdot2(b3, kOnes, br);
    int bk = s.bk;
    bk = sdot2(b0, (K) | ((K + 1) << 16), bk); bk = sdot2(b1, (K + 2) | ((K + 3) << 16), bk);
    bk = sdot2(b2, (K + 4) | ((K + 5) << 16), bk); bk = sdot2(b3, (K + 6) | ((K + 7) << 16), bk);
    s.bk = bk;
    if (squares) {   // wave-uniform: a scalar branch (the empty volatile statement keeps it from becoming a copy and a select)
        asm volatile("");
        int q = (int)s.bb32;
        q = sdot2(b0, b0, q); q = sdot2(b1, b1, q); q = sdot2(b2, b2, q); q = sdot2(b3, b3, q);
        s.bb32 = (uint32_t)q;
    }
}
// the row's totals into the plain and the row-weighted sums
MC_D void acc_add_row(Acc& s, int cnt, int kr, int br, int r, int r2)
{
    s.n += cnt;
    s.k1 += kr;
    s.b0 += br;
    s.rn = mad24(r, cnt, s.rn);
    s.rrn = mad24(r2, cnt, s.rrn);
    s.rk = mad24(r, kr, s.rk);
    s.rb = mad24(r, br, s.rb);
}

// strip-local sums -> frame moments, added to the table row of lab_add and subtracted from the row of lab_sub
// (either may be >= S: skipped).  All products are 24-bit or 32x32->64 (u0 < 2^13, vb < 2^12, n <= 1024, strip of
// <= 128 rows); the table is int64 modulo 2^64, so a subtraction that comes before its addition is harmless.
MC_D void sums_apply2(uint32_t lab_add, uint32_t lab_sub, uint32_t u0, uint32_t vb, int n, int k1, int k2, int rn,
                      int rrn, int rk, int b0, int bk, int rb, uint32_t bb_lo, uint32_t bb_hi, unsigned long long* tab, int S, int ts,
                      bool rot = false)
{
    const bool add = lab_add < (uint32_t)S, sub = lab_sub < (uint32_t)S;
    if (!add && !sub) return;
    const int su = __mul24((int)u0, n) + k1, sv = __mul24((int)vb, n) + rn;
    const int x = __mul24((int)u0, rn) + rk;
    unsigned long long v[kMomStride];
    v[0] = (unsigned long long)(long long)n;
    v[1] = (unsigned long long)(long long)su;
    v[2] = (unsigned long long)(long long)sv;
    v[3] = (unsigned long long)((long long)(int)u0 * (long long)(su + k1) + (long long)k2);
    v[4] = (unsigned long long)((long long)(int)vb * (long long)su + (long long)x);
    v[5] = (unsigned long long)((long long)(int)vb * (long long)(sv + rn) + (long long)rrn);
    v[6] = (unsigned long long)(long long)b0;
    v[7] = (unsigned long long)((long long)(int)u0 * (long long)b0 + (long long)bk);
    v[8] = (unsigned long long)((long long)(int)vb * (long long)b0 + (long long)rb);
    v[9] = ((unsigned long long)bb_hi << 32) | bb_lo;
    // rot (the log's drain: odd lanes): the fields in the order 5..9, 0..4.  Neighbouring records often name the same
    // two labels -- a boundary that crosses several strips in the same rows -- and neighbouring lanes then do not meet
    // on one address (same-row collisions cost 2.8 us of a 67 us launch; this takes about half of them away)
    constexpr int H = kMomStride / 2;
#pragma unroll
    for (int k = 0; k < H; ++k) {
        const unsigned long long lo = v[k], hi = v[k + H];
        v[k] = rot ? hi : lo; v[k + H] = rot ? lo : hi;
    }
    const int o0 = rot ? H : 0, o1 = rot ? -H : 0;
    if (add) {
        unsigned long long* t = tab + (size_t)lab_add * ts;
        unsigned long long *t0 = t + o0, *t1 = t + o1;
#pragma unroll
        for (int k = 0; k < H; ++k) atomicAdd(t0 + k, v[k]);
#pragma unroll
        for (int k = H; k < kMomStride; ++k) atomicAdd(t1 + k, v[k]);
    }
    if (sub) {
        unsigned long long* t = tab + (size_t)lab_sub * ts;
        unsigned long long *t0 = t + o0, *t1 = t + o1;
#pragma unroll
        for (int k = 0; k < H; ++k) __hip_atomic_fetch_sub(t0 + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ds_sub_u64: no negation
#pragma unroll
        for (int k = H; k < kMomStride; ++k) __hip_atomic_fetch_sub(t1 + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
MC_D void sums_apply(uint32_t lab, uint32_t u0, uint32_t vb, int n, int k1, int k2, int rn, int rrn, int rk,
                     int b0, int bk, int rb, uint32_t bb_lo, uint32_t bb_hi, unsigned long long* tab, int S, int ts)
{
    sums_apply2(lab, 0xFFFFFFFFu, u0, vb, n, k1, k2, rn, rrn, rk, b0, bk, rb, bb_lo, bb_hi, tab, S, ts);
}

MC_D uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// record: the label the sums are added to (s.lab), the label they are subtracted from (lab_new), the strip's
// origin (its column strip g, shifted to bit 17, and its first row vb) and the running sums, in ten dwords: an LDS store
// costs the CU about 8 cycles per dword whatever the number of lanes that write (tools/micro/lds_sparse_write.hip: two
// lanes, ds_write_b128 33 cycles, b64 17, b32 8), and two or three lanes write a record on most rows
MC_D void rec_write(uint32_t* wlog, int idx, const Acc& s, uint32_t lab_new, uint32_t g17, uint32_t vb)
{
    // two 16-byte LDS stores and an 8-byte one (the record stride and the log base are multiples of 16 bytes)
    typedef uint32_t lds_u32x4 __attribute__((ext_vector_type(4), aligned(16)));
    typedef uint32_t lds_u32x2 __attribute__((ext_vector_type(2), aligned(8)));
    __attribute__((address_space(3))) lds_u32x4* q =
        (__attribute__((address_space(3))) lds_u32x4*)wlog + idx * (kRecDwords / 4);
    // labels are 16 bits (the sentinels' low halves are 0xFFFF >= S); vb < 2^12, n <= 1024; k1 < 2^14, k2 < 2^18; rn < 2^17, g < 2^10
    q[0] = lds_u32x4{ __builtin_amdgcn_perm(lab_new, s.lab, 0x05040100u), vb | ((uint32_t)s.n << 12),
                      (uint32_t)s.k1 | ((uint32_t)s.k2 << 14), (uint32_t)s.rn | g17 };
    q[1] = lds_u32x4{ (uint32_t)s.rrn, (uint32_t)s.rk, (uint32_t)s.b0, (uint32_t)s.bk };
    *reinterpret_cast<__attribute__((address_space(3))) lds_u32x2*>(q + 2) = lds_u32x2{ (uint32_t)s.rb, s.bb32 };
}

MC_D void rec_drain(const uint32_t* wlog, int cnt, unsigned long long* tab, int S, int ts, int lane, int np)
{
    const int nact = __popcll((unsigned long long)ballot64(true));
    for (int i = lane; i < cnt; i += nact) {
        const uint4* q = reinterpret_cast<const uint4*>(__builtin_assume_aligned(wlog, 16)) + i * (kRecDwords / 4);
        const uint4 a = q[0], b = q[1];
        const uint2 c = *reinterpret_cast<const uint2*>(q + 2);
        sums_apply2(a.x & 0xFFFFu, a.x >> 16, (a.w >> 17) * (uint32_t)np, a.y & 0xFFFu, (int)(a.y >> 12), (int)(a.z & 0x3FFFu),
                    (int)(a.z >> 14), (int)(a.w & 0x1FFFFu), (int)b.x, (int)b.y, (int)b.z, (int)b.w, (int)c.x, c.y, 0u, tab, S, ts,
                    (i & 1) != 0);
    }
}

// append the accumulator of every lane in mask m (pred = this lane's bit); at most one record per lane, so a
// log of 64 always has room after a drain (wave-uniform control flow)
MC_D void rec_append(uint32_t* wlog, int& cnt, uint64_t m, bool pred, const Acc& s, uint32_t lab_new, uint32_t g17,
                     uint32_t vb, unsigned long long* tab, int S, int ts, int lane, int np)
{
    const int k = __popcll((unsigned long long)m);
    if (__builtin_expect(cnt + k > kRecCap, 0)) { rec_drain(wlog, cnt, tab, S, ts, lane, np); cnt = 0; }
    if (pred) rec_write(wlog, cnt + mbcnt(m), s, lab_new, g17, vb);
    cnt += k;
}

// work off the queued chunks: one chunk (8 NQ pixels) per lane; its pixels that match neither the left nor the
// right label are added to the table directly (runs of one label merged; normally one run per chunk)
template <int NQ>
MC_D void queue_process(const uint32_t* wq, int qcnt, const uint4* dpf, const uint4* lpf, int G,
                        unsigned long long* tab, int S, int ts, int lane)
{
    constexpr int NP = 8 * NQ;
    const int nact = __popcll((unsigned long long)ballot64(true));
    for (int base = 0; base < qcnt; base += nact) {
        const bool act = base + lane < qcnt;
        const uint32_t ent = act ? wq[base + lane] : 0u;
        const uint32_t row = ent & 0x7FFFu, g = ent >> 16;     // absolute row of the frame, strip index
        const bool wide = (ent & 0x8000u) != 0u;               // a row with disparities beyond 11 bits: its squares too
        uint32_t lw[4 * NQ], dw[4 * NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            uint4 l4 = make_uint4(0u, 0u, 0u, 0u), d4 = l4;
            if (act) { l4 = lpf[((size_t)row * G + g) * NQ + q]; d4 = dpf[((size_t)row * G + g) * NQ + q]; }
            lw[4 * q] = l4.x; lw[4 * q + 1] = l4.y; lw[4 * q + 2] = l4.z; lw[4 * q + 3] = l4.w;
            dw[4 * q] = d4.x; dw[4 * q + 1] = d4.y; dw[4 * q + 2] = d4.z; dw[4 * q + 3] = d4.w;
        }
        const uint32_t l0 = lw[0] & 0xFFFFu, l7 = lw[4 * NQ - 1] >> 16;
        uint32_t clab = kNoLabel;
        int n = 0, k1 = 0, k2 = 0, b0 = 0, bk = 0;
        uint64_t bb = 0;
#pragma unroll
        for (int k = 1; k < NP - 1; ++k) {
            const uint32_t lab = (k & 1) ? (lw[k >> 1] >> 16) : (lw[k >> 1] & 0xFFFFu);
            const int b = (int)(short)((k & 1) ? (dw[k >> 1] >> 16) : (dw[k >> 1] & 0xFFFFu));
            const bool isrest = act && lab != l0 && lab != l7;
            const bool emit = isrest && n != 0 && lab != clab;
            if (ballot64(emit)) {   // a second run of rest pixels in one chunk: rare
                if (emit) {
                    sums_apply(clab, g * (uint32_t)NP, row, n, k1, k2, 0, 0, 0, b0, bk, 0, (uint32_t)bb, (uint32_t)(bb >> 32), tab, S, ts);
                    n = 0; k1 = 0; k2 = 0; b0 = 0; bk = 0; bb = 0;
                }
            }
            if (isrest) {
                clab = lab; n += 1; k1 += k; k2 += k * k; b0 += b; bk += k * b;
                bb += (uint64_t)(uint32_t)(b * b);
            }
        }
        if (n != 0) sums_apply(clab, g * (uint32_t)NP, row, n, k1, k2, 0, 0, 0, b0, bk, 0, (uint32_t)bb, (uint32_t)(bb >> 32), tab, S, ts);
        // The strip accumulators skipped the squares of this row (they keep them in 32 bits): add sum b^2 of the
        // pixels that match the left or the right label here, one run of equal labels at a time (the pixels that
        // match neither were added with their squares above).
        if (ballot64(act && wide)) {
            uint32_t rl = kNoLabel;
            uint64_t rb2 = 0;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const uint32_t lab = (k & 1) ? (lw[k >> 1] >> 16) : (lw[k >> 1] & 0xFFFFu);
                const int b = (int)(short)((k & 1) ? (dw[k >> 1] >> 16) : (dw[k >> 1] & 0xFFFFu));
                const bool mine = act && wide && (lab == l0 || lab == l7);
                const bool emit = mine && rb2 != 0 && lab != rl;
                if (ballot64(emit)) {
                    if (emit) {
                        if (rl < (uint32_t)S) atomicAdd(tab + (size_t)rl * ts + 9, (unsigned long long)rb2);
                        rb2 = 0;
                    }
                }
                if (mine) { rl = lab; rb2 += (uint64_t)((int64_t)b * (int64_t)b); }
            }
            if (rb2 != 0 && rl < (uint32_t)S) atomicAdd(tab + (size_t)rl * ts + 9, (unsigned long long)rb2);
        }
    }
}

typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

// hand-issued 16-byte loads and counted wait: the frame's base address in an SGPR pair, a 32-bit byte offset per
// lane (one VALU add per row instead of 64-bit address arithmetic) and an immediate offset for the second quad
template <int OFF>
MC_D void gload16s(u32x4v& dst, uint32_t byte_off, const void* sbase)
{
    // nt: the planes are read once and never again -- as non-temporal loads the same access pattern streams at
    // 6.5 TB/s instead of 5.7 (option k1_debug = 1: 50 against 57.5 us per 256 frames)
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(dst) : "v"(byte_off), "s"(sbase), "n"(OFF) : "memory");
}
// the registers of a row set are tied to the wait so that no use can move above it
template <int NQ> struct RowSet { u32x4v d[NQ], l[NQ]; };
template <int N>
MC_D void wait_vm(RowSet<1>& r)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r.d[0]), "+v"(r.l[0]) : "n"(N) : "memory");
}
template <int N>
MC_D void wait_vm(RowSet<2>& r)
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r.d[0]), "+v"(r.l[0]), "+v"(r.d[1]), "+v"(r.l[1]) : "n"(N) : "memory");
}

// value the optimiser cannot see through: keeps a 0 / ~0 mask a register that is ANDed (a select costs
// 1.5x an AND on gfx950, and 5x in its VCC form)
MC_D uint32_t opaque(uint32_t x) { asm("" : "+v"(x)); return x; }

// NQ: 16-byte quads per plane, row and lane -- a lane owns a strip 8 NQ pixels wide (NQ = 2 when the width is a
// multiple of 16: the per-row work that does not depend on the pixel count -- label logic, row-weighted sums,
// evictions and their LDS stores -- is then paid once per 16 pixels).
// RAGGED: some strip of the launch has fewer than R rows (H not a multiple of bands * nsub); lanes past
// their strip then keep their labels and are masked out.  The common full-strip case pays nothing for it.
// STREAM (measurement only, option "k1_debug" = 1): the same launch shape, loads and waits with the arithmetic
// replaced by an OR of the words read -- the rate this access pattern reaches on its own (bench.py reports it
// beside the kernel's rate as roofline.stream_only_GBps).  It writes no moments.
template <int NQ, bool TABLE_LDS, bool RAGGED, bool STREAM = false, int NSET = 3>
__global__ __launch_bounds__(NQ == 2 ? 512 : 1024) void sp_moments_lr_kernel(MomentArgs a)
{
    constexpr int NP = 8 * NQ;                 // pixels per lane and row; NSET = row sets in flight
    constexpr int WAITN = 2 * NQ * (NSET - 1); // younger loads when a row is consumed
    extern __shared__ __attribute__((aligned(16))) unsigned long long lds[];
    const int frame = a.first_frame + (int)blockIdx.y;
    const int band = (int)blockIdx.x;
    const int S = a.S[frame];
    const int T = (int)blockDim.x, tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = uni(tid >> 6);
    const int W = a.W, H = a.H, G = W / NP;
    unsigned long long* gmom = a.mom + (size_t)frame * a.s_max * kMomStride;
    unsigned long long* tab = TABLE_LDS ? lds : gmom;
    constexpr int ts = TABLE_LDS ? kTabStrideLds : kMomStride;      // row stride of the table in words
    uint32_t* wbase = reinterpret_cast<uint32_t*>(lds + (TABLE_LDS ? (size_t)a.table_rows * kTabStrideLds : 0)) +
                      wave * (kRecCap * kRecDwords + kQueueCap);
    uint32_t* wlog = wbase;
    uint32_t* wq = wbase + kRecCap * kRecDwords;

    const int rows_per_band = (H + a.bands - 1) / a.bands;
    const int r0 = band * rows_per_band;
    const int r1 = min(H, r0 + rows_per_band);
    // T = nsub * G (host): sub-band p, column strip g
    const int nsub = T / G, p = tid / G, g = tid % G;
    const int R = (max(0, r1 - r0) + nsub - 1) / nsub;       // <= 128 / NQ (host): 32-bit strip-local sums
    const int rs = r0 + p * R, re = min(r1, rs + R);
    const int count = max(0, re - rs);                        // == R for every lane unless RAGGED
    const uint32_t g17 = (uint32_t)g << 17;                         // the strip's column, where a record carries it
    const size_t frame_base = ((size_t)frame * a.frame_stride) / 8;   // in uint4 units (W % 8 == 0)
    const uint4* dpf = reinterpret_cast<const uint4*>(a.disp) + frame_base;
    const uint4* lpf = reinterpret_cast<const uint4*>(a.label) + frame_base;

    Acc L, Rr;
    acc_reset(L, kFlushL);
    acc_reset(Rr, kFlushR);
    int cnt = 0, qcnt = 0;     // wave-uniform

    // NSET register sets hold the rows it .. it+NSET-1 of the strip (three: with non-temporal loads two to three rows in
    // flight measured 1.5 % faster than four, five and six slower; tools/ab).  The loads are issued by hand and
    // waited for with counted s_waitcnt (in-order return): a set is reloaded with row it+NSET as soon as its
    // words have been consumed, and a row only waits until all but the younger loads have landed,
    // so 4-6 KB (NQ = 1) or 8-12 KB (NQ = 2) per wave stay in flight under the arithmetic.  Every load is issued
    // by every wave (rows past the strip's end read the frame's first row at the lane's column, which stays in the
    // cache), otherwise the count of younger loads would not be the constant the wait relies on.
    const int cl = max(count, 1);
    const int pitch = G * 16 * NQ;                                             // bytes per row of a plane
    const uint32_t safe0 = (uint32_t)(min(rs, H - 1) * pitch + g * 16 * NQ);   // byte offset, < 2^25 per frame
    const uint32_t tail0 = (uint32_t)(g * 16 * NQ);
    auto load_row = [&](const int j, RowSet<NQ>& rr) __attribute__((always_inline)) {
        // rows past the strip's end are loaded only to keep the count of loads in flight constant: they read the
        // frame's first row at the lane's column, 1280 bytes per plane that stay in the cache (re-reading the strip's
        // last row cost 9 % more HBM traffic once the loads were non-temporal)
        const int lim = RAGGED ? cl : R;
        const uint32_t o = j < lim ? safe0 + (uint32_t)__mul24(j, pitch) : tail0;
        gload16s<0>(rr.d[0], o, dpf);
        gload16s<0>(rr.l[0], o, lpf);
        if constexpr (NQ == 2) {
            gload16s<16>(rr.d[1], o, dpf);
            gload16s<16>(rr.l[1], o, lpf);
        }
    };
    RowSet<NQ> sets[NSET];
#pragma unroll
    for (int k = 0; k < NSET; ++k) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) { sets[k].d[q] = 0; sets[k].l[q] = 0; }
    }
#pragma unroll
    for (int k = 0; k < NSET; ++k) load_row(k, sets[k]);
    // the table is cleared under the latency of the first loads
    if (TABLE_LDS) {
        for (int i = tid; i < S * kTabStrideLds; i += T) tab[i] = 0ull;
        __syncthreads();
    }
    uint32_t sink = 0;
    auto row = [&](const int it, RowSet<NQ>& rr) __attribute__((always_inline)) {
        if constexpr (STREAM) {
            wait_vm<WAITN>(rr);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                sink |= (rr.d[q].x | rr.d[q].y) | (rr.d[q].z | rr.d[q].w) | (rr.l[q].x | rr.l[q].y) | (rr.l[q].z | rr.l[q].w);
            load_row(it + NSET, rr);
            return;
        }
        wait_vm<WAITN>(rr);
        uint32_t lw[4 * NQ], dw[4 * NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            lw[4 * q] = rr.l[q].x; lw[4 * q + 1] = rr.l[q].y; lw[4 * q + 2] = rr.l[q].z; lw[4 * q + 3] = rr.l[q].w;
            dw[4 * q] = rr.d[q].x; dw[4 * q + 1] = rr.d[q].y; dw[4 * q + 2] = rr.d[q].z; dw[4 * q + 3] = rr.d[q].w;
        }
        const bool live = !RAGGED || it < count;
        // a lane that is past its strip keeps its labels
        const uint32_t l0 = live ? (lw[0] & 0xFFFFu) : L.lab;
        const uint32_t l7 = live ? (lw[4 * NQ - 1] >> 16) : l0;
        const bool evL = l0 != L.lab;
        const bool two = l7 != l0;
        const bool evR = two && l7 != Rr.lab;
        const uint32_t patL = l0 | (l0 << 16), patR = l7 | (l7 << 16);
        const uint32_t twoM = opaque((two && live) ? 0xFFFFFFFFu : 0u);
        const uint32_t liveM = RAGGED ? opaque(live ? 0xFFFFFFFFu : 0u) : 0xFFFFFFFFu;
        uint32_t mL[4 * NQ], mR[4 * NQ], bL[4 * NQ], bR[4 * NQ];
        uint32_t orb = 0;
#pragma unroll
        for (int j = 0; j < 4 * NQ; ++j) {
            mL[j] = match01(lw[j], patL);
            if (RAGGED) mL[j] &= liveM;
            mR[j] = match01(lw[j], patR) & twoM;
            bL[j] = pk_mul_lo_u16(dw[j], mL[j]);
            bR[j] = pk_mul_lo_u16(dw[j], mR[j]);
            orb |= dw[j];
        }
        // rows whose disparities all fit 11 bits (every Kinect frame) take the 32-bit sum of squares
        const uint64_t wide = ballot64((orb & 0xF800F800u) != 0u);
        // this set's words are consumed: fetch row it+NSET into it.  (No use of lw[] / dw[] may follow: the loads
        // land asynchronously in these registers, and a later use would make the compiler copy them before then.)
        load_row(it + NSET, rr);

        // the left accumulator follows the label of pixel 0: when that changes, the running sums are logged
        // with both labels (added to the old one, subtracted from the new one)
        // the right accumulator follows the label of the last pixel while that differs from pixel 0
        const uint64_t emL = ballot64(evL);
        const uint64_t emR = ballot64(two) & ballot64(l7 != Rr.lab);   // = ballot64(evR), as two compares and a scalar AND
        if (emL | emR) {     // one branch on the common path
            if (emL) rec_append(wlog, cnt, emL, evL, L, l0, g17, (uint32_t)rs, tab, S, ts, lane, NP);
            if (emR) rec_append(wlog, cnt, emR, evR, Rr, l7, g17, (uint32_t)rs, tab, S, ts, lane, NP);
        }
        L.lab = l0;
        Rr.lab = evR ? l7 : Rr.lab;
        int cL = 0, kL = 0, brL = 0, cR = 0, kR = 0, brR = 0;
        const bool squares = wide == 0;
        acc_add_half<0>(L, mL, bL, squares, cL, kL, brL);
        acc_add_half<0>(Rr, mR, bR, squares, cR, kR, brR);
        if constexpr (NQ == 2) {
            acc_add_half<8>(L, mL + 4, bL + 4, squares, cL, kL, brL);
            acc_add_half<8>(Rr, mR + 4, bR + 4, squares, cR, kR, brR);
        }
        const int r2 = it * it;       // wave-uniform: one scalar multiply
        acc_add_row(L, cL, kL, brL, it, r2);
        acc_add_row(Rr, cR, kR, brR, it, r2);
        // pixels in neither, or a row whose squares were left out: queue the chunk
        const bool partial = live && cL + cR != NP;
        const bool rest = partial || (live && !squares);
        uint64_t rm = ballot64(partial);
        if (!squares) rm = RAGGED ? ballot64(live) : ballot64(true);     // wave-uniform
        if (rm) {
            // room for this row's chunks (checked here, where the queue is written, not once per row)
            if (__builtin_expect(qcnt + __popcll((unsigned long long)rm) > kQueueCap, 0)) {
                queue_process<NQ>(wq, qcnt, dpf, lpf, G, tab, S, ts, lane);
                qcnt = 0;
            }
            if (rest) wq[qcnt + mbcnt(rm)] = (uint32_t)(rs + it) | (squares ? 0u : 0x8000u) | ((uint32_t)g << 16);
            qcnt += __popcll((unsigned long long)rm);
        }
    };
    for (int it = 0; it < R; it += NSET) {
        bool done = false;
#pragma unroll
        for (int k = 0; k < NSET; ++k) {
            if (!done) {
                row(it + k, sets[k]);
                done = it + k + 1 >= R;
            }
        }
    }
    // the last reloads are never consumed, but their registers must not be reused while they are in flight
#pragma unroll
    for (int k = 0; k < NSET; ++k) wait_vm<0>(sets[k]);
    if constexpr (STREAM) {
        if (sink == 0x12345678u) wq[lane] = sink;    // keeps the loads alive; practically never taken
        return;
    }
    // flush: queued chunks, both accumulators, the log
    if (qcnt) queue_process<NQ>(wq, qcnt, dpf, lpf, G, tab, S, ts, lane);
    {
        // the end of the strip closes both accumulators: their sums go to the label they follow, nothing is subtracted
        const bool wl = L.lab < (uint32_t)S, wr = Rr.lab < (uint32_t)S;
        rec_append(wlog, cnt, ballot64(wl), wl, L, kFlushL, g17, (uint32_t)rs, tab, S, ts, lane, NP);
        rec_append(wlog, cnt, ballot64(wr), wr, Rr, kFlushR, g17, (uint32_t)rs, tab, S, ts, lane, NP);
    }
    rec_drain(wlog, cnt, tab, S, ts, lane, NP);

    if (!TABLE_LDS) return;
    __syncthreads();
    // LDS rows are kTabStrideLds words apart, HBM rows kMomStride
    if (a.bands == 1) {
        for (int i = tid; i < S * kMomStride; i += T) { const int r = i / kMomStride; gmom[i] = tab[i + r * (kTabStrideLds - kMomStride)]; }
    } else {
        for (int i = tid; i < S * kMomStride; i += T) {
            const int r = i / kMomStride;
            unsigned long long x = tab[i + r * (kTabStrideLds - kMomStride)];
            if (x) atomicAdd(gmom + i, x);
        }
    }
}

// Generic fallback for widths that are not a multiple of 8: one pixel per
// thread-iteration, global atomics per pixel run of one lane.  Correct for any
// shape; not the measured path.
__global__ void sp_moments_generic_kernel(MomentArgs a)
{
    const int frame = a.first_frame + (int)blockIdx.y;
    const int S = a.S[frame];
    const int W = a.W, H = a.H;
    unsigned long long* gmom = a.mom + (size_t)frame * a.s_max * kMomStride;
    const size_t base = (size_t)frame * a.frame_stride;
    const int rows_per_band = (H + a.bands - 1) / a.bands;
    const int r0 = (int)blockIdx.x * rows_per_band, r1 = min(H, r0 + rows_per_band);
    for (int v = r0; v < r1; ++v) {
        // each thread owns a contiguous span of the row
        const int span = (W + (int)blockDim.x - 1) / (int)blockDim.x;
        const int ub = (int)threadIdx.x * span, ue = min(W, ub + span);
        Slot A; slot_reset(A, 0xFFFFFFFFu);
        for (int u = ub; u < ue; ++u) {
            const uint32_t l = a.label[base + (size_t)v * W + u];
            const int b = a.disp[base + (size_t)v * W + u];
            if (l >= (uint32_t)S) continue;
            if (l != A.lab) { slot_flush<false>(A, nullptr, gmom, S); slot_reset(A, l); }
            slot_add1(A, u, v, b);
        }
        slot_flush<false>(A, nullptr, gmom, S);
    }
}

// ---------------------------------------------------------------------------
// K1b: local fit.  One thread per (frame, superpixel).
// ---------------------------------------------------------------------------
__global__ void local_fit_kernel(FitArgs a)
{
    const int frame = a.first_frame + (int)blockIdx.y;
    const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int S = a.S[frame];
    if (s >= S) return;
    const long long* m = reinterpret_cast<const long long*>(a.mom) + ((size_t)frame * a.s_max + s) * kMomStride;
    double* f = a.fit + ((size_t)frame * a.s_max + s) * kFitStride;
    const long long cx = a.W / 2, cy = a.H / 2;
    const long long n = m[0];
    const long long su = m[1] - cx * n, sv = m[2] - cy * n;
    const long long suu = m[3] - 2 * cx * m[1] + cx * cx * n;
    const long long suv = m[4] - cx * m[2] - cy * m[1] + cx * cy * n;
    const long long svv = m[5] - 2 * cy * m[2] + cy * cy * n;
    const long long sb = m[6];
    const long long sub = m[7] - cx * m[6];
    const long long svb = m[8] - cy * m[6];
    const __int128 du = (__int128)n * suu - (__int128)su * su;
    const __int128 dvv = (__int128)n * svv - (__int128)sv * sv;
    // isZeroDisparity (OverSegmentation.cpp:422-432) keeps a superpixel as soon as one entry is >= 0.5; entries
    // are shorts widened through size_t (:148), so every non-zero value counts (a negative one becomes ~1.8e19):
    // "some b != 0" <=> Sbb != 0 (a sum of squares), which no mix of signs can cancel
    const bool ok = (n >= 10) && (du != 0) && (dvv != 0) && (m[9] != 0);
    a.valid[(size_t)frame * a.s_max + s] = ok ? 1 : 0;
    f[0] = (double)suu; f[1] = (double)suv; f[2] = (double)svv;
    f[3] = (double)su;  f[4] = (double)sv;  f[5] = (double)n;
    f[6] = (double)sub; f[7] = (double)svb; f[8] = (double)sb;
    double le = 0.0, ssr = 0.0;
    Vec3 x = { 0.0, 0.0, 0.0 };
    if (ok) {
        Sym3 A = { f[0], f[1], f[2], f[3], f[4], f[5] };
        Vec3 r = { f[6], f[7], f[8] }, d;
        const double quad = ldl3(A, r, x, d);
        ssr = (double)m[9] - quad;
        le = ssr / (2.0 * a.var);
    }
    f[9] = x.x; f[10] = x.y; f[11] = x.z;
    f[12] = (double)m[9]; f[13] = ssr; f[14] = 0.0; f[15] = 0.0;
    a.local_err[(size_t)frame * a.s_max + s] = le;
}

// ---------------------------------------------------------------------------
// K2: edge probabilities.  One thread per directed CSR entry.
// ---------------------------------------------------------------------------
__global__ void edge_prob_kernel(EdgeArgs a)
{
    const int frame = a.first_frame + (int)blockIdx.y;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const int S = a.S[frame];
    const uint32_t E2 = a.E2[frame];
    if (e >= E2) return;
    const uint32_t* rowp = a.rowp + (size_t)frame * (a.s_max + 1);
    const uint32_t j = a.col[(size_t)frame * a.e2_max + e];
    // row of entry e: largest i with rowp[i] <= e
    int lo = 0, hi = S;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (rowp[mid] <= e) lo = mid; else hi = mid; }
    const int i = lo;
    const uint8_t* valid = a.valid + (size_t)frame * a.s_max;
    double* qo = a.q + (size_t)frame * a.e2_max + e;
    if (j >= (uint32_t)S || (int)j == i || !valid[i] || !valid[j]) { *qo = -1.0; return; }
    const double* fi = a.fit + ((size_t)frame * a.s_max + i) * kFitStride;
    const double* fj = a.fit + ((size_t)frame * a.s_max + j) * kFitStride;
    const double log_var = det_log(a.var);
    int inf_flag = 0;
    // together {i, j}: sums of two records; the wave-order sum of two ssr values is one IEEE addition
    // whichever lanes they sit in, and a + b == b + a
    double mt[kSumStride];
#pragma unroll
    for (int k = 0; k < 9; ++k) mt[k] = fi[k] + fj[k];
    mt[9] = fi[12] + fj[12];
    Vec3 x; double err;
    const double T = segment_from_sums(mt, fi[13] + fj[13], a.var, log_var, x, err, &inf_flag);
    // separate {i}, {j}: the residual is exactly zero (same operations as the local fit)
    double mi[kSumStride], mj[kSumStride];
#pragma unroll
    for (int k = 0; k < 9; ++k) { mi[k] = fi[k]; mj[k] = fj[k]; }
    mi[9] = fi[12]; mj[9] = fj[12];
    const double pi = segment_from_sums(mi, fi[13], a.var, log_var, x, err, &inf_flag);
    const double pj = segment_from_sums(mj, fj[13], a.var, log_var, x, err, &inf_flag);
    const double Sp = pi + pj;
    double lB, lS;
    if (T > Sp) { lB = T; lS = Sp; } else { lS = T; lB = Sp; }
    const double sum = lB + det_log(1.0 + det_exp(lS - lB));
    const double ratio = det_exp(T - sum);
    if (ratio != ratio) atomicOr(a.status + frame, 1);
    if (inf_flag) atomicOr(a.status + frame, 2);
    *qo = ratio;
}

// ---------------------------------------------------------------------------
// Segment(superPixels): one wavefront evaluates one explicit member set.
// mem[i] != 0 marks members (device byte array of S entries).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void segment_prob_kernel(SegArgs a)
{
    const int lane = lane_id();
    const int S = a.S;
    const double* fit = a.fit;
    double acc[kSumStride];
#pragma unroll
    for (int k = 0; k < kSumStride; ++k) acc[k] = 0.0;
    double rpart = 0.0;
    int cnt = 0;
    for (int i = lane; i < S; i += kWave) {
        if (!a.mem[i]) continue;
        const double* f = fit + (size_t)i * kFitStride;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] += f[k];
        acc[9] += f[12];
        rpart += f[13];
        cnt++;
    }
#pragma unroll
    for (int k = 0; k < kSumStride; ++k) acc[k] = wave_sum_f64(acc[k]);  // exact integers: order-free
    const double R = wave_sum_f64(rpart);                                 // wave order by id
    cnt = wave_sum_i32(cnt);
    Vec3 x = { 0, 0, 0 };
    double p = 0.0, err = 0.0;
    int inf_flag = 0;
    if (cnt > 0) p = segment_from_sums(acc, R, a.var, det_log(a.var), x, err, &inf_flag);
    if (lane == 0) {
        a.out[0] = p; a.out[1] = err; a.out[2] = x.x; a.out[3] = x.y; a.out[4] = x.z;
        a.out[5] = inf_flag ? 1.0 : 0.0;
    }
}

// Member list -> membership bytes (the allocation-free entry point stages the list in the library's static
// scratch; mem[0..S) was cleared on the stream before).  Duplicate ids write the same 1.
__global__ void mark_members_kernel(const uint32_t* list, int n, uint8_t* mem)
{
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k < n) mem[list[k]] = 1;
}

// ---------------------------------------------------------------------------
// Device self-tests of the shared math (parity of det_log/det_exp/division/Philox).
// ---------------------------------------------------------------------------
__global__ void selftest_math_kernel(const double* x, const double* y, int n, double* ol, double* oe, double* od)
{
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    ol[i] = det_log(x[i]);
    oe[i] = det_exp(x[i]);
    od[i] = x[i] / y[i];
}
// one wave per block: eleven wave_sum_f64 and one wave_sum11_f64 of the same values
__global__ __launch_bounds__(64) void selftest_wave_sums_kernel(const double* x, double* plain, double* folded)
{
    const int lane = (int)threadIdx.x, b = (int)blockIdx.x;
    double v[11], w[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) { v[k] = x[((size_t)b * 11 + k) * 64 + lane]; w[k] = wave_sum_f64(v[k]); }
    wave_sum11_f64(v, lane);
    if (lane < 11) {
        double pw = 0.0, fv = 0.0;
#pragma unroll
        for (int k = 0; k < 11; ++k) if (lane == k) { pw = w[k]; fv = v[k]; }
        plain[(size_t)b * 11 + lane] = pw; folded[(size_t)b * 11 + lane] = fv;
    }
}
__global__ void selftest_philox_kernel(const uint32_t* ck, uint32_t* out)
{
    u32x4 c = { ck[0], ck[1], ck[2], ck[3] };
    u32x4 r = philox4x32_10(c, ck[4], ck[5]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_moments(const MomentArgs& a, int n_frames, int threads, size_t lds_bytes, int mode, hipStream_t st)
{
    // mode 3: left/right kernel with the table in LDS, 4: the same with the table in HBM, 2: generic (any width)
    dim3 grid(a.bands, n_frames);
    if (mode == 3 || mode == 4) {
        // a.lane_px: 8 or 16 pixels per lane and row (host: 16 only when W % 16 == 0)
        const int npx = a.lane_px == 16 ? 16 : 8;
        const void* fns[] = {
            reinterpret_cast<const void*>(&sp_moments_lr_kernel<1, true, false>), reinterpret_cast<const void*>(&sp_moments_lr_kernel<1, true, true>),
            reinterpret_cast<const void*>(&sp_moments_lr_kernel<2, true, false>), reinterpret_cast<const void*>(&sp_moments_lr_kernel<2, true, true>),
            reinterpret_cast<const void*>(&sp_moments_lr_kernel<1, true, true, true>), reinterpret_cast<const void*>(&sp_moments_lr_kernel<2, true, true, true>) };
        static DeviceOnce once3;
        bool* attr3 = once3.slot();
        if (!attr3 || !*attr3) {
            for (const void* f : fns) {
                hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
                if (e != hipSuccess) return e;
            }
            if (attr3) *attr3 = true;
        }
        // ragged: some strip has fewer rows than the others
        const int rpb = (a.H + a.bands - 1) / a.bands, nsub = threads / (a.W / npx);
        const bool ragged = (a.H % a.bands) != 0 || (rpb % nsub) != 0;
        const dim3 blk(threads);
        if (a.debug == 1) {
            if (npx == 16) hipLaunchKernelGGL((sp_moments_lr_kernel<2, true, true, true>), grid, blk, lds_bytes, st, a);
            else hipLaunchKernelGGL((sp_moments_lr_kernel<1, true, true, true>), grid, blk, lds_bytes, st, a);
        } else if (npx == 16) {
            if (mode == 3) {
                if (ragged) hipLaunchKernelGGL((sp_moments_lr_kernel<2, true, true>), grid, blk, lds_bytes, st, a);
                else hipLaunchKernelGGL((sp_moments_lr_kernel<2, true, false>), grid, blk, lds_bytes, st, a);
            } else {
                if (ragged) hipLaunchKernelGGL((sp_moments_lr_kernel<2, false, true>), grid, blk, lds_bytes, st, a);
                else hipLaunchKernelGGL((sp_moments_lr_kernel<2, false, false>), grid, blk, lds_bytes, st, a);
            }
        } else {
            if (mode == 3) {
                if (ragged) hipLaunchKernelGGL((sp_moments_lr_kernel<1, true, true>), grid, blk, lds_bytes, st, a);
                else hipLaunchKernelGGL((sp_moments_lr_kernel<1, true, false>), grid, blk, lds_bytes, st, a);
            } else {
                if (ragged) hipLaunchKernelGGL((sp_moments_lr_kernel<1, false, true>), grid, blk, lds_bytes, st, a);
                else hipLaunchKernelGGL((sp_moments_lr_kernel<1, false, false>), grid, blk, lds_bytes, st, a);
            }
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(sp_moments_generic_kernel, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

size_t moments_lr_lds_bytes(int table_rows, int threads)
{
    return (size_t)table_rows * kTabStrideLds * 8 + (size_t)((threads + 63) / 64) * (kRecCap * kRecDwords + kQueueCap) * 4;
}

hipError_t launch_local_fit(const FitArgs& a, int n_frames, hipStream_t st)
{
    dim3 grid((a.s_max + 127) / 128, n_frames);
    hipLaunchKernelGGL(local_fit_kernel, grid, dim3(128), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_edge_prob(const EdgeArgs& a, int n_frames, hipStream_t st)
{
    if (a.e2_max == 0) return hipSuccess;
    dim3 grid((a.e2_max + 127) / 128, n_frames);
    hipLaunchKernelGGL(edge_prob_kernel, grid, dim3(128), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_segment_prob(const SegArgs& a, hipStream_t st)
{
    hipLaunchKernelGGL(segment_prob_kernel, dim3(1), dim3(64), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_mark_members(const uint32_t* list, int n, uint8_t* mem, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(mark_members_kernel, dim3((n + 255) / 256), dim3(256), 0, st, list, n, mem);
    return hipGetLastError();
}

hipError_t launch_selftest_math(const double* x, const double* y, int n, double* ol, double* oe, double* od, hipStream_t st)
{
    hipLaunchKernelGGL(selftest_math_kernel, dim3((n + 127) / 128), dim3(128), 0, st, x, y, n, ol, oe, od);
    return hipGetLastError();
}
hipError_t launch_selftest_wave_sums(const double* x, int n_waves, double* plain, double* folded, hipStream_t st)
{
    hipLaunchKernelGGL(selftest_wave_sums_kernel, dim3(n_waves), dim3(64), 0, st, x, plain, folded);
    return hipGetLastError();
}
hipError_t launch_selftest_philox(const uint32_t* ck, uint32_t* out, hipStream_t st)
{
    hipLaunchKernelGGL(selftest_philox_kernel, dim3(1), dim3(1), 0, st, ck, out);
    return hipGetLastError();
}

}  // namespace mc
