// Device helpers shared by the PU-Flow HIP kernels (gfx950 / CDNA4 only).
//
// Register data layout used by every MLP-shaped kernel here ("channel-major chain"):
//   v_mfma_f32_16x16x4_f32 computes D(16x16) = A(16x4) * B(4x16) + C with
//     A: lane l holds A[row = l&15][k = l>>4]          (one float)
//     B: lane l holds B[k = l>>4][col = l&15]          (one float)
//     C/D: lane l holds D[row = 4*(l>>4) + r][col = l&15], r = 0..3   (float4)
//   We put OUTPUT CHANNELS on rows and POINTS/EDGES on the 16 columns, so a 16-channel block of
//   activations for 16 columns is one float4 per lane:  lane (col = l&15, q = l>>4), register r
//   <-> channel 16*cb + 4*q + r.  That is exactly the B operand the NEXT layer needs if its
//   K-step "r" consumes channels {16*cb + 4*q + r | q = 0..3}: no LDS round trip, no shuffles
//   between layers.  The host packs each weight matrix accordingly (packing.frag_pack):
//   fragment (ob, cb) is 64 lanes x float4, lane l = W[16*ob + (l&15)][16*cb + 4*(l>>4) + r].
//   The f32 MFMA is an exact k-ordered fp32 fma chain (no reduced precision).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

typedef float f4 __attribute__((ext_vector_type(4)));

#define PF_WAVE 64

__device__ __forceinline__ f4 pf_mfma(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f4 pf_splat(float v) { f4 r = {v, v, v, v}; return r; }

// leaky ReLU for 0 <= slope <= 1 as max(x, slope * x): same bits as the select form (x >= 0 ? x : slope * x)
// for every finite x including -0, one packed multiply + one max per value instead of compare + multiply + select
__device__ __forceinline__ f4 pf_lrelu(f4 v, float slope) {
    const f4 s = v * slope;
    f4 r;
    r.x = fmaxf(v.x, s.x); r.y = fmaxf(v.y, s.y); r.z = fmaxf(v.z, s.z); r.w = fmaxf(v.w, s.w);
    return r;
}

__device__ __forceinline__ f4 pf_relu(f4 v) {
    f4 r;
    r.x = fmaxf(v.x, 0.f); r.y = fmaxf(v.y, 0.f); r.z = fmaxf(v.z, 0.f); r.w = fmaxf(v.w, 0.f);
    return r;
}

__device__ __forceinline__ float pf_lrelu1(float v, float slope) { return fmaxf(v, v * slope); }

// ---- weight fragment sources -----------------------------------------------------------
// Every lane-address of a weight fragment is (wave-uniform base) + lane*16.  hipcc otherwise
// hoists one 64-bit VGPR address pair PER FRAGMENT out of the persistent tile loop (hundreds of
// VGPRs); both sources below keep it to ONE VGPR:
//   PfWLds : fragments resident in LDS, ds_read_b128 with a 16-bit immediate offset
//   PfWBuf : fragments in global memory through a buffer descriptor (SGPRs), uniform soffset
struct PfWLds {
    const f4* base;     // LDS
    int lane;
    __device__ __forceinline__ f4 load(int frag) const { return base[frag * PF_WAVE + lane]; }
};

struct PfWBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;           // lane * 16
    __device__ __forceinline__ PfWBuf(const void* p, int lane)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000)), voff(lane * 16) {}
    __device__ __forceinline__ f4 load(int frag) const {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, frag * (PF_WAVE * 16), 0);
        return __builtin_bit_cast(f4, v);
    }
};

// acc[p][acc0+ob] += W[ob][cb] * in[p][in0+cb]  for ob < OB, cb < CB.
// Fragment (ob, cb) of this matrix is `frag0 + ob*WCB + cb` of the weight source
// (WCB = number of 16-channel input blocks of the packed matrix = its row stride).
// P independent column tiles share each weight fragment (and give independent MFMA chains).
// Software pipeline: D fragments are kept in flight; a sched_barrier after every fragment's MFMAs
// stops hipcc from hoisting ALL loads to the top of the kernel.
template <int OB, int CB, int WCB, int D = 3, class WS, int P, int NIN, int NACC>
__device__ __forceinline__ void pf_mm(const WS& ws, int frag0, const f4 (&in)[P][NIN], int in0,
                                      f4 (&acc)[P][NACC], int acc0) {
    constexpr int NFRAG = OB * CB;
    constexpr int DD = D < NFRAG ? D : NFRAG;
    f4 wb[DD];
#pragma unroll
    for (int i = 0; i < DD; ++i) wb[i] = ws.load(frag0 + (i / CB) * WCB + (i % CB));
#pragma unroll
    for (int i = 0; i < NFRAG; ++i) {
        const int ob = i / CB, cb = i % CB;
        const f4 wv = wb[i % DD];
        if (i + DD < NFRAG) wb[i % DD] = ws.load(frag0 + ((i + DD) / CB) * WCB + ((i + DD) % CB));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int p = 0; p < P; ++p)
                acc[p][acc0 + ob] = pf_mfma(wv[r], in[p][in0 + cb][r], acc[p][acc0 + ob]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// ---- split-fp16 ("f16x2") path: fp32 accuracy in THREE fp16 MFMAs per 32-channel step ---------------
// x = hi + lo' * 2^-11 with hi = rne_f16(x) and lo' = rne_f16((x - hi) * 2^11): x - hi is exact in fp32 and at most
// half an fp16 ulp of x, so lo' never exceeds |x| (no overflow from the scale), is a normal fp16 number whenever x
// is (precision independent of magnitude) and hi + lo' 2^-11 carries 22+ significant bits.  A product keeps
// hi.hi in the main accumulator and hi.lo' + lo'.hi in a second one that is folded in as  acc + accx * 2^-11;
// the dropped lo.lo term is < 2^-24 relative.  gfx950's fp16 MFMA honours subnormal operands (probed), so tiny
// values degrade gracefully.  Range: |x| must stay below 65504 (fp16 max) - activations of this network are O(1..100);
// beyond it the result is inf/NaN (loud).  Used by csrc/cnf.hip; the eval kernels of the discrete path moved to the natural-scale variant below.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
constexpr float PF_LO_SCALE = 2048.f, PF_LO_INV = 1.f / 2048.f;

__device__ __forceinline__ u2 pf_split2_pair(float a, float b) {
    unsigned hb = __builtin_bit_cast(unsigned, (h2){(_Float16)a, (_Float16)b});
    asm("" : "+v"(hb));          // opaque: read hi back from the packed register instead of converting twice
    const h2 h = __builtin_bit_cast(h2, hb);
    const h2 l = {(_Float16)((a - (float)h.x) * PF_LO_SCALE), (_Float16)((b - (float)h.y) * PF_LO_SCALE)};
    return (u2){hb, __builtin_bit_cast(unsigned, l)};
}

struct PfPair2 { h8 h, l; };
__device__ __forceinline__ PfPair2 pf_pair2(f4 b0, f4 b1) {
    const u2 s0 = pf_split2_pair(b0.x, b0.y), s1 = pf_split2_pair(b0.z, b0.w);
    const u2 s2 = pf_split2_pair(b1.x, b1.y), s3 = pf_split2_pair(b1.z, b1.w);
    PfPair2 p;
    p.h = __builtin_bit_cast(h8, (u4){s0.x, s1.x, s2.x, s3.x});
    p.l = __builtin_bit_cast(h8, (u4){s0.y, s1.y, s2.y, s3.y});
    return p;
}

__device__ __forceinline__ f4 pf_mfma_f16(h8 a, h8 b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

#ifndef PF_MM2_DEPTH
#define PF_MM2_DEPTH 8
#endif
#ifndef PF_W2LDS_DEPTH
#define PF_W2LDS_DEPTH 2
#endif
// weights: [frag][2 splits hi / lo'][64 lanes] x 16 B in LDS
struct PfW2Lds {
    static constexpr int DEPTH = PF_W2LDS_DEPTH;   // fragments in flight (pf_mm2f): LDS latency
    const u4* base;
    int lane;
    __device__ __forceinline__ h8 load(int frag, int split) const {
        return __builtin_bit_cast(h8, base[(frag * 2 + split) * PF_WAVE + lane]);
    }
};

template <int DEPTH_>
struct PfW2BufD {
    static constexpr int DEPTH = DEPTH_;
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;           // lane * 16
    __device__ __forceinline__ PfW2BufD(const void* p, int lane)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000)), voff(lane * 16) {}
    __device__ __forceinline__ h8 load(int frag, int split) const {
        return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (frag * 2 + split) * (PF_WAVE * 16), 0));
    }
};

struct PfW2Buf {
    static constexpr int DEPTH = PF_MM2_DEPTH;   // L2 latency wants ~500 cycles of MFMA work in flight
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;           // lane * 16
    __device__ __forceinline__ PfW2Buf(const void* p, int lane)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000)), voff(lane * 16) {}
    __device__ __forceinline__ h8 load(int frag, int split) const {
        return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (frag * 2 + split) * (PF_WAVE * 16), 0));
    }
};

// acc[p][acc0+ob] += Wh xh ;  accx[p][acc0+ob] += Wh xl' + Wl' xh   over block PAIRS cp < CP.
// Caller folds:  value = acc + accx * PF_LO_INV.  D fragments (two 16-B reads each) are kept in flight: 2 is enough
// for LDS-resident weights, L2-resident weights behind buffer loads want ~500 cycles of cover (PF_MM2_DEPTH).
template <int OB, int CP, int WCP, int D = 2, class WS2, int P, int NIN, int NACC>
__device__ __forceinline__ void pf_mm2(const WS2& ws, int frag0, const PfPair2 (&in)[P][NIN], int in0,
                                       f4 (&acc)[P][NACC], f4 (&accx)[P][NACC], int acc0) {
    constexpr int NFRAG = OB * CP;
    constexpr int DD = D < NFRAG ? D : NFRAG;
    h8 wb[DD][2];
#pragma unroll
    for (int i = 0; i < DD; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) wb[i][s] = ws.load(frag0 + (i / CP) * WCP + (i % CP), s);
#pragma unroll
    for (int i = 0; i < NFRAG; ++i) {
        const int ob = i / CP, cp = i % CP;
        const h8 wh = wb[i % DD][0], wl = wb[i % DD][1];
        if (i + DD < NFRAG) {
            const int f = frag0 + ((i + DD) / CP) * WCP + ((i + DD) % CP);
#pragma unroll
            for (int s = 0; s < 2; ++s) wb[i % DD][s] = ws.load(f, s);
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            f4 x = accx[p][acc0 + ob];
            x = pf_mfma_f16(wh, in[p][in0 + cp].l, x);
            x = pf_mfma_f16(wl, in[p][in0 + cp].h, x);
            accx[p][acc0 + ob] = x;
            acc[p][acc0 + ob] = pf_mfma_f16(wh, in[p][in0 + cp].h, acc[p][acc0 + ob]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// NB 16-channel blocks in[p][in0 .. in0+NB) -> (NB+1)/2 split block pairs (an odd tail pairs with a zero block)
template <int NB, int P, int NIN>
__device__ __forceinline__ void pf_pairs2(const f4 (&in)[P][NIN], int in0, PfPair2 (&out)[P][(NB + 1) / 2]) {
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int c = 0; c < (NB + 1) / 2; ++c)
            out[p][c] = pf_pair2(in[p][in0 + 2 * c], 2 * c + 1 < NB ? in[p][in0 + 2 * c + 1] : pf_splat(0.f));
}

// acc[p][acc0+ob] += W[ob][cp] in[p][cp] with the cross accumulator folded in (complete split-fp16 product)
template <int OB, int CP, int WCP, class WS2, int P, int NIN, int NACC>
__device__ __forceinline__ void pf_mm2f(const WS2& ws, int frag0, const PfPair2 (&in)[P][NIN], int in0,
                                        f4 (&acc)[P][NACC], int acc0) {
    f4 accx[P][OB];
    f4 accm[P][OB];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int o = 0; o < OB; ++o) { accx[p][o] = pf_splat(0.f); accm[p][o] = acc[p][acc0 + o]; }
    pf_mm2<OB, CP, WCP, WS2::DEPTH>(ws, frag0, in, in0, accm, accx, 0);
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int o = 0; o < OB; ++o) acc[p][acc0 + o] = accm[p][o] + accx[p][o] * PF_LO_INV;
}

// ---- split-fp16 with a NATURAL-scale low half ("f16n"): one accumulator, no fold ----------------------------------
// x = hi + lo, hi = rne_f16(x), lo = rne_f16(x - hi) written straight into the packed operand by v_fma_mixlo/mixhi_f16
// (3 VALU per value pair instead of 9).  gfx950's fp16 MFMA honours subnormal operands, so a small lo keeps an ABSOLUTE
// precision of 2^-25 in the units the operand is stored in; hosts keep that floor far below fp32 rounding by storing the
// weights scaled by a power of two (packing.frag_pack_f16n: max |W'| in [2^13, 2^14)) and, where the range allows,
// the activations too.  All three product terms (hi.hi, hi.lo, lo.hi) go into ONE fp32 accumulator.
struct PfPairN { h8 h, l; };
__device__ __forceinline__ unsigned pf_pk_f16(float a, float b) {
    return __builtin_bit_cast(unsigned, (h2){(_Float16)a, (_Float16)b});          // v_cvt_pk_f16_f32 (RNE)
}
__device__ __forceinline__ PfPairN pf_pairn(f4 b0, f4 b1) {
    const unsigned h0 = pf_pk_f16(b0.x, b0.y), h1 = pf_pk_f16(b0.z, b0.w), h2_ = pf_pk_f16(b1.x, b1.y), h3 = pf_pk_f16(b1.z, b1.w);
    unsigned l0, l1, l2, l3;
    // lo = fp16(x - hi) as fma(hi, -1, x); the trailing s_nop covers VALU write -> MFMA operand read (hipcc pads nothing
    // for registers written inside an asm statement)
    asm("v_fma_mixlo_f16 %0, %4, -1.0, %8 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %5, -1.0, %10 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %2, %6, -1.0, %12 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %7, -1.0, %14 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %4, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %5, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %6, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %7, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "s_nop 1"
        : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "v"(h0), "v"(h1), "v"(h2_), "v"(h3), "v"(b0.x), "v"(b0.y), "v"(b0.z), "v"(b0.w), "v"(b1.x), "v"(b1.y), "v"(b1.z), "v"(b1.w));
    PfPairN p;
    p.h = __builtin_bit_cast(h8, (u4){h0, h1, h2_, h3});
    p.l = __builtin_bit_cast(h8, (u4){l0, l1, l2, l3});
    return p;
}

// reduce-scatter max steps: (a, b) -> lanes 0..31 get max over both halves of a, lanes 32..63 of b;  rows: the same
// between odd and even 16-lane rows.  Inline asm because the builtin's two results are mis-paired by hipcc 7.2 once they
// are bit-cast to float; the leading s_nop is the VALU write -> permlane read hazard.
__device__ __forceinline__ float pf_rsmax32(float a, float b) {
    unsigned x = __builtin_bit_cast(unsigned, a), y = __builtin_bit_cast(unsigned, b);
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return fmaxf(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y));
}
__device__ __forceinline__ float pf_rsmax16(float a, float b) {
    unsigned x = __builtin_bit_cast(unsigned, a), y = __builtin_bit_cast(unsigned, b);
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return fmaxf(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y));
}

// acc[p][o] += W[o][cp] feat[p][cp] over CP block pairs, three fp16 MFMAs per pair into the one accumulator (small terms
// first).  SWAP = false: D[channel][edge] (weights are the A operand);  SWAP = true: D[edge][channel].
#ifndef PF_MMN_CPMAJOR
#define PF_MMN_CPMAJOR 0
#endif
// Products per 32-channel step: 3 = split-fp16 with the natural-scale low half (hi*lo + lo*hi + hi*hi: fp32-grade results,
// the parity mode and the default); 1 = hi*hi only - plain fp16 operands, fp32 accumulation: the reduced-precision THROUGHPUT
// build (`libpuflow_hip_f16.so`, BASELINE configs[1] "bf16/fp16" line; judged by Chamfer distance, never the headline)
#ifndef PF_MMN_TERMS
#define PF_MMN_TERMS 3
#endif
#ifndef PF_MMN_PRIO
#define PF_MMN_PRIO 0
#endif
template <bool SWAP, int OB, int CP, int WCP, bool FENCE = true, class WS, int P, int NIN, int NACC>
__device__ __forceinline__ void pf_mmn(const WS& ws, int frag0, const PfPairN (&feat)[P][NIN], f4 (&acc)[P][NACC], int in0 = 0, int acc0 = 0) {
    constexpr int D = WS::DEPTH;                  // fragments in flight: 2 from LDS, 8 behind buffer loads (L2 latency)
    constexpr int NFRAG = OB * CP;
    constexpr int DD = D < NFRAG ? D : NFRAG;
    // fragment walk: ob-major (one accumulator's whole chain, then the next) or cp-major (accumulators alternate)
    auto OBI = [](int i) { return PF_MMN_CPMAJOR ? i % OB : i / CP; };
    auto CPI = [](int i) { return PF_MMN_CPMAJOR ? i / OB : i % CP; };
    h8 wb[DD][2];
#pragma unroll
    for (int i = 0; i < DD; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) wb[i][s] = ws.load(frag0 + OBI(i) * WCP + CPI(i), s);
#if PF_MMN_PRIO
    __builtin_amdgcn_s_setprio(PF_MMN_PRIO);      // tuning build: a wave inside an MFMA chain wins the issue arbitration
#endif
#pragma unroll
    for (int i = 0; i < NFRAG; ++i) {
        const int ob = OBI(i), cp = CPI(i);
        const h8 wh = wb[i % DD][0], wl = wb[i % DD][1];
        if (i + DD < NFRAG) {
            const int f = frag0 + OBI(i + DD) * WCP + CPI(i + DD);
#pragma unroll
            for (int s = 0; s < 2; ++s) wb[i % DD][s] = ws.load(f, s);
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            f4 x = acc[p][acc0 + ob];
            if constexpr (SWAP) {
                if constexpr (PF_MMN_TERMS == 3) {
                    x = pf_mfma_f16(feat[p][in0 + cp].l, wh, x);
                    x = pf_mfma_f16(feat[p][in0 + cp].h, wl, x);
                }
                x = pf_mfma_f16(feat[p][in0 + cp].h, wh, x);
            } else {
                if constexpr (PF_MMN_TERMS == 3) {
                    x = pf_mfma_f16(wh, feat[p][in0 + cp].l, x);
                    x = pf_mfma_f16(wl, feat[p][in0 + cp].h, x);
                }
                x = pf_mfma_f16(wh, feat[p][in0 + cp].h, x);
            }
            acc[p][acc0 + ob] = x;
        }
        if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);      // FENCE = false: the caller interleaves two streams itself
    }
#if PF_MMN_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}


// cooperative global -> LDS copy of n 16-byte words by the whole workgroup (call before a __syncthreads).  D loads per thread
// are in flight before the first LDS store: a plain load -> store loop pays one full memory latency per iteration, and at
// 88 - 152 KiB of weights per workgroup (6 - 32 iterations) that was most of what a small-batch launch of these kernels cost.
template <int D = 8, class T4>
__device__ __forceinline__ void pf_stage_lds(T4* __restrict__ dst, const T4* __restrict__ src, int n) {
    const int nt = blockDim.x;
    for (int base = threadIdx.x; base < n; base += nt * D) {
        T4 v[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int i = base + k * nt;
            v[k] = src[i < n ? i : n - 1];
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int i = base + k * nt;
            if (i < n) dst[i] = v[k];
        }
    }
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void pf_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pf_static_for<I + 1, N>(f);
    }
}

// bias / per-channel vector as accumulator initialiser: channels 16*ob + 4*q .. +3
__device__ __forceinline__ f4 pf_bias(const float* __restrict__ b, int ob, int q) {
    return *reinterpret_cast<const f4*>(b + ob * 16 + 4 * q);
}

// max over the 16 lanes of a DPP row (= the 16 columns of one MFMA tile); every lane gets the max.
__device__ __forceinline__ float pf_rowmax16(float v) {
    // row_ror:n = 0x120 + n
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false)));
    return v;
}

// XCD-aware tile order (guide T1): virtual id v -> logical tile so that the workgroups that
// share an XCD (equal blockIdx % 8 under round-robin dispatch) walk one contiguous chunk of
// tiles (= the same batch items -> their gather table stays in that XCD's L2).  Speed only.
__device__ __forceinline__ int pf_xcd_tile(int v, int chunk) { return (v & 7) * chunk + (v >> 3); }
