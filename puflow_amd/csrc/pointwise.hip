// Per-point stages that follow each EdgeConv unit, as two kernels:
//
//   pf_pq_gemm   PQ' = Wpq h + bpq        next unit's per-point EdgeConv vectors P | Q (packing.py: the exact per-point fold
//                                         of the edge feature [x_i, x_j, x_j - x_i], interpflow.py:229-232)
//   pf_cond      c   = W2 relu(W1 h + b1)                     FeatMergeUnit   (interpflow.py:251-258)
//                s,t = LinearA1D_s(c), LinearA1D_t(c)         AffineInjectorLayer nets (coupling.py:132-134,
//                                                             interpflow.py:22-43): they depend on c only, so they are
//                                                             computed once per ORIGINAL point, shared by f and the R replicas of g
//                cp  = W0[:, tdim:] c                         c-part of coupling1's first layer (interpflow.py:38-41)
//                (c itself is written only on request: W2 is folded into the first layers of the three nets)
//
// Arithmetic of both: split-fp16 products with a natural-scale low half (pf_mfma.h "f16n": three fp16 MFMAs per
// 32-channel step into one fp32 accumulator); weights arrive pre-split and pre-scaled by a power of two per matrix
// (packing.frag_pack_f16n_scaled), the kernels multiply by its inverse.
//
// pf_pq_gemm is the [2S x ODIM] x [ODIM x T] GEMM that produces 134 MB per launch at 32 x 2048 points: its roofline is
// the HBM write.  Mapping: the WEIGHTS are register-resident (a wave owns 16 RB output rows for the whole launch: RB x CP
// fragment pairs = 64 VGPRs at ODIM = 128), the POINTS stream: a round stages NT tiles of 16 points in LDS as ready-made
// B operands (every wave converts 1 / NW of the round's tiles: fp32 -> (hi, lo) once per point, not once per consumer),
// then every wave multiplies every tile by its rows.  LDS bytes per MFMA are half of what streaming the weights would
// need, no weight byte is re-read from L2 after the prologue, and the operand split costs 1/16 of the VALU work.
//
// pf_cond keeps its 120 KiB of weights in LDS for the life of a persistent workgroup (one wave = 16 points).
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"
#include "pf_mfma.h"

namespace {

__device__ __forceinline__ h8 pf_ldw(const u4* p) { return __builtin_bit_cast(h8, *p); }

// ---------------------------------------------------------------------------------------------------------------------
struct PqArgs {
    const float* h;       // [T, ODIM]
    const u4* w;          // f16n fragment image [ROWS/16][CP][hi / lo][64 lanes]
    const float* bias;    // [ROWS]   true scale, added after the rescale
    const float* scales;  // POST_SCALES: [6] = 2^-sw of the PQ matrix
    float* pq;            // [T, ROWS]
    int T, ntiles, NT, nrounds;
    float inv;            // used when scales == nullptr (pf_cnf_context)
};

template <int ODIM, int ROWS, int NW>
__global__ __launch_bounds__(NW * 64) void pq_gemm_kernel(PqArgs a) {
    constexpr int HB = ODIM / 16, CP = (HB + 1) / 2, OBT = ROWS / 16, RB = OBT / NW, NTMAX = 16;
    static_assert(OBT % NW == 0, "rows split evenly over the waves");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    __shared__ u4 bl[NTMAX][CP][2][64];
    h8 wh[RB][CP], wl[RB][CP];
    f4 bias[RB];
#pragma unroll
    for (int ob = 0; ob < RB; ++ob) {
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
            const u4* f = a.w + (size_t)(((wave * RB + ob) * CP + cp) * 2) * 64 + lane;
            wh[ob][cp] = pf_ldw(f);
            wl[ob][cp] = pf_ldw(f + 64);
        }
        bias[ob] = *reinterpret_cast<const f4*>(a.bias + (wave * RB + ob) * 16 + 4 * q);
    }
    const float inv = a.scales ? a.scales[6] : a.inv;
    for (int round = blockIdx.x; round < a.nrounds; round += gridDim.x) {
        const int tile0 = round * a.NT;
        const int nt = a.ntiles - tile0 < a.NT ? a.ntiles - tile0 : a.NT;
        // ---- stage: this wave converts tiles wave, wave + NW, ...
        for (int t = wave; t < nt; t += NW) {
            int pt = (tile0 + t) * 16 + col;
            pt = pt < a.T ? pt : a.T - 1;
            f4 hb[HB + (HB & 1)];
#pragma unroll
            for (int b = 0; b < HB; ++b) hb[b] = *reinterpret_cast<const f4*>(a.h + (size_t)pt * ODIM + b * 16 + 4 * q);
            if constexpr (HB & 1) hb[HB] = pf_splat(0.f);
#pragma unroll
            for (int cp = 0; cp < CP; ++cp) {
                const PfPairN p = pf_pairn(hb[2 * cp], hb[2 * cp + 1]);
                bl[t][cp][0][lane] = __builtin_bit_cast(u4, p.h);
                bl[t][cp][1][lane] = __builtin_bit_cast(u4, p.l);
            }
        }
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            f4 acc[RB];
#pragma unroll
            for (int ob = 0; ob < RB; ++ob) acc[ob] = pf_splat(0.f);
#pragma unroll
            for (int cp = 0; cp < CP; ++cp) {
                const h8 fh = __builtin_bit_cast(h8, bl[t][cp][0][lane]), fl = __builtin_bit_cast(h8, bl[t][cp][1][lane]);
#pragma unroll
                for (int ob = 0; ob < RB; ++ob) {
                    f4 x = acc[ob];
                    if constexpr (PF_MMN_TERMS == 3) {
                        x = pf_mfma_f16(wh[ob][cp], fl, x);
                        x = pf_mfma_f16(wl[ob][cp], fh, x);
                    }
                    x = pf_mfma_f16(wh[ob][cp], fh, x);
                    acc[ob] = x;
                }
            }
            const int pt = (tile0 + t) * 16 + col;
            if (pt < a.T) {
#pragma unroll
                for (int ob = 0; ob < RB; ++ob) {
                    f4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = fmaf(acc[ob][r], inv, bias[ob][r]);
                    *reinterpret_cast<f4*>(a.pq + (size_t)pt * ROWS + (wave * RB + ob) * 16 + 4 * q) = o;
                }
            }
        }
        __syncthreads();                                   // the next round overwrites the staged tiles
    }
}

template <int ODIM, int ROWS>
int launch_pq(PqArgs a, hipStream_t s) {
    constexpr int NW = 16;
    a.ntiles = (a.T + 15) / 16;
    // tiles per round: as many as LDS holds when there is plenty of work, fewer when that would leave CUs idle
    int nt = a.ntiles / 256;
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    a.NT = nt;
    a.nrounds = (a.ntiles + nt - 1) / nt;
    const int grid = a.nrounds < 256 ? a.nrounds : 256;    // one workgroup per CU (LDS, 16 waves)
    hipLaunchKernelGGL((pq_gemm_kernel<ODIM, ROWS, NW>), dim3(grid), dim3(NW * 64), 0, s, a);
    return pf_last_launch_status();
}

// the same kernel for the continuous model's context GEMM ctx = c Hc^T + hb ([288 x cd] x [cd x T], 75 MB written per
// block at 32 x 2048): 18 row blocks = 9 waves x 2
template <int ODIM>
int launch_ctx(PqArgs a, hipStream_t s) {
    constexpr int NW = 9;
    a.ntiles = (a.T + 15) / 16;
    int nt = a.ntiles / 256;
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    a.NT = nt;
    a.nrounds = (a.ntiles + nt - 1) / nt;
    const int grid = a.nrounds < 256 ? a.nrounds : 256;
    hipLaunchKernelGGL((pq_gemm_kernel<ODIM, 288, NW>), dim3(grid), dim3(NW * 64), 0, s, a);
    return pf_last_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
struct CondArgs {
    const float* h;                       // [T, ODIM]
    const float* w;                       // blob base
    long long off[13];                    // POST_SLOTS
    float* c;                             // [T, CDIM]   (nullable)
    float* st;                            // [T, 8]      s0 s1 s2 t0 t1 t2 - -
    float* cp;                            // [T, 64]
    int T, ntiles;
    int contiguous;                       // M1 | H1 | S2 | T2 | ST4 lie back to back in the blob (packing.py packs them so): one copy
};

__device__ __forceinline__ f4 pf_scale(f4 v, float k) { return v * k; }

// body of the conditioner stage of one unit for workgroup tile0 of tstride (a.ntiles = WAVE tiles of 16 points);  wl: LDS for
// the unit's fragment images (cond_lds_bytes<ODIM>())
template <int ODIM>
constexpr int cond_lds_bytes() { return ((ODIM / 32) * ((ODIM / 16 + 1) / 2) + 12 * ((ODIM / 32 + 1) / 2) + 20) * 2048; }

template <int ODIM, int CDIM, int NW>
__device__ __forceinline__ void cond_body(const CondArgs& a, u4* wl, int tile0, int tstride) {
    constexpr int HB = ODIM / 16, MB = ODIM / 32, CB = CDIM / 16;
    constexpr int HP = (HB + 1) / 2, MP = (MB + 1) / 2;              // block pairs (32 input channels per MFMA step)
    // LDS-resident fragment images: M1 | H1 | S2 | T2 | ST4   (2 KiB per fragment pair)
    constexpr int F_M1 = 0, F_H1 = F_M1 + MB * HP, F_S2 = F_H1 + 12 * MP, F_T2 = F_S2 + 8, F_ST4 = F_T2 + 8, F_END = F_ST4 + 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    static_assert(F_END * 2048 == cond_lds_bytes<ODIM>(), "LDS size formula");
    {
        auto stage = [&](int f0, int nf, long long off) {
            const u4* src = reinterpret_cast<const u4*>(a.w + off);
            pf_stage_lds(wl + f0 * 128, src, nf * 128);
        };
        if (a.contiguous) stage(F_M1, F_END, a.off[0]);
        else {
            stage(F_M1, MB * HP, a.off[0]); stage(F_H1, 12 * MP, a.off[3]); stage(F_S2, 8, a.off[4]); stage(F_T2, 8, a.off[6]);
            stage(F_ST4, 4, a.off[8]);
        }
        __syncthreads();
    }
    const PfW2Lds wsM1{wl + F_M1 * 128, lane}, wsH1{wl + F_H1 * 128, lane}, wsS2{wl + F_S2 * 128, lane},
        wsT2{wl + F_T2 * 128, lane}, wsST4{wl + F_ST4 * 128, lane};
    const PfW2BufD<2> wsM2(a.w + a.off[2], lane);      // only for the `cs` API output: shallow prefetch, few registers
    const float* sc = a.w + a.off[12];
    const float iM1 = sc[0], iM2 = sc[1], iH1 = sc[2], iS2 = sc[3], iT2 = sc[4], iST4 = sc[5];
    const float* b1 = a.w + a.off[1];
    const float* bS2 = a.w + a.off[5];
    const float* bT2 = a.w + a.off[7];
    const float* bST4 = a.w + a.off[9];

    // the workgroup owns the consecutive WAVE tiles (16 points each) [tile0 * per, (tile0 + 1) * per), its waves take them
    // round-robin (tstride = workgroups of this unit: every one gets the same number of wave tiles)
    const int per = (a.ntiles + tstride - 1) / tstride;
    const int wt_end = min((tile0 + 1) * per, a.ntiles);
    for (int wt = tile0 * per + wave; wt < wt_end; wt += NW) {
        const int g = wt * 16 + col;
        const bool ok = g < a.T;
        const int pt = ok ? g : a.T - 1;
        PfPairN hp[1][HP];
        {
            f4 h[HB + (HB & 1)];
#pragma unroll
            for (int b = 0; b < HB; ++b) h[b] = *reinterpret_cast<const f4*>(a.h + (size_t)pt * ODIM + b * 16 + 4 * q);
            if constexpr (HB & 1) h[HB] = pf_splat(0.f);
#pragma unroll
            for (int c = 0; c < HP; ++c) hp[0][c] = pf_pairn(h[2 * c], h[2 * c + 1]);
        }
        // ---- merge MLP: m = relu(W1 h + b1)
        PfPairN mp[1][MP];
        {
            f4 m[1][MB + (MB & 1)];
#pragma unroll
            for (int o = 0; o < MB; ++o) m[0][o] = pf_bias(b1, o, q);
            pf_mmn<false, MB, HP, HP>(wsM1, 0, hp, m);
#pragma unroll
            for (int o = 0; o < MB; ++o) m[0][o] = pf_relu(pf_scale(m[0][o], iM1));
            if constexpr (MB & 1) m[0][MB] = pf_splat(0.f);
#pragma unroll
            for (int c = 0; c < MP; ++c) mp[0][c] = pf_pairn(m[0][2 * c], m[0][2 * c + 1]);
        }
        // c = W2 m is only materialised when the caller wants `cs` (feat_extract API): everything downstream of c
        // is linear in it, so the host folded W2 into those layers (H1 := [s_W0; t_W0; c1_W0c] W2, packing.pack_plan)
        if (a.c) {
            f4 c[1][CB];
#pragma unroll
            for (int o = 0; o < CB; ++o) c[0][o] = pf_splat(0.f);
            pf_mmn<false, CB, MP, MP>(wsM2, 0, mp, c);
            if (ok)
#pragma unroll
                for (int o = 0; o < CB; ++o)
                    *reinterpret_cast<f4*>(a.c + (size_t)pt * CDIM + o * 16 + 4 * q) = pf_scale(c[0][o], iM2);
        }
        if (!a.st) continue;                               // conditioning features only (the continuous model: pf_cond_all with st = cp = NULL)
        // ---- coupling1 c-part (rows 128..191 of H1), stored raw
        {
            f4 acc[1][4];
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[0][o] = pf_splat(0.f);
            pf_mmn<false, 4, MP, MP>(wsH1, 8 * MP, mp, acc);
            if (ok)
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    *reinterpret_cast<f4*>(a.cp + (size_t)pt * 64 + o * 16 + 4 * q) = pf_scale(acc[0][o], iH1);
        }
        // ---- injector nets: hidden1 -> hidden2 for s (pairs 0..1) and t (pairs 2..3)
        PfPairN h2p[1][4];
        pf_static_for<0, 2>([&](auto nc) {
            constexpr int net = decltype(nc)::value;
            f4 h1[1][4];
#pragma unroll
            for (int o = 0; o < 4; ++o) h1[0][o] = pf_splat(0.f);
            pf_mmn<false, 4, MP, MP>(wsH1, (4 * net) * MP, mp, h1);
            PfPairN h1p[1][2];
            h1p[0][0] = pf_pairn(pf_lrelu(pf_scale(h1[0][0], iH1), 0.01f), pf_lrelu(pf_scale(h1[0][1], iH1), 0.01f));
            h1p[0][1] = pf_pairn(pf_lrelu(pf_scale(h1[0][2], iH1), 0.01f), pf_lrelu(pf_scale(h1[0][3], iH1), 0.01f));
            f4 h2[1][4];
#pragma unroll
            for (int o = 0; o < 4; ++o) h2[0][o] = pf_bias(net == 0 ? bS2 : bT2, o, q);
            if constexpr (net == 0) pf_mmn<false, 4, 2, 2>(wsS2, 0, h1p, h2); else pf_mmn<false, 4, 2, 2>(wsT2, 0, h1p, h2);
            const float i2 = net == 0 ? iS2 : iT2;
            h2p[0][2 * net + 0] = pf_pairn(pf_lrelu(pf_scale(h2[0][0], i2), 0.01f), pf_lrelu(pf_scale(h2[0][1], i2), 0.01f));
            h2p[0][2 * net + 1] = pf_pairn(pf_lrelu(pf_scale(h2[0][2], i2), 0.01f), pf_lrelu(pf_scale(h2[0][3], i2), 0.01f));
        });
        {
            f4 acc[1][1];
            acc[0][0] = pf_bias(bST4, 0, q);
            pf_mmn<false, 1, 4, 4>(wsST4, 0, h2p, acc);
            if (ok && q < 2) *reinterpret_cast<f4*>(a.st + (size_t)pt * 8 + 4 * q) = pf_scale(acc[0][0], iST4);
        }
    }
}

#ifndef PF_COND_NW
#define PF_COND_NW 12
#endif
constexpr int COND_NW = PF_COND_NW;

template <int ODIM, int CDIM>
__global__ __launch_bounds__(COND_NW * 64) void cond_kernel(CondArgs a) {
    extern __shared__ u4 cond_lds[];
    cond_body<ODIM, CDIM, COND_NW>(a, cond_lds, blockIdx.x, gridDim.x);
}

// All six units in ONE launch: workgroup -> (unit, slot) through `first[]` (units 0 / 1 are cheaper and get fewer workgroups).
struct CondAllArgs { CondArgs u[6]; int first[7]; };

__global__ __launch_bounds__(COND_NW * 64) void cond_all_kernel(CondAllArgs g) {
    extern __shared__ u4 cond_lds[];
    const int b = blockIdx.x;
    int unit = 0;
#pragma unroll
    for (int i = 1; i < 6; ++i) unit += b >= g.first[i] ? 1 : 0;
    const int slot = b - g.first[unit], n = g.first[unit + 1] - g.first[unit];
    if (unit == 0) cond_body<32, 32, COND_NW>(g.u[0], cond_lds, slot, n);
    else if (unit == 1) cond_body<64, 64, COND_NW>(g.u[1], cond_lds, slot, n);
    else {
        // the four 128-channel units share ONE inlined body: their argument records are selected with scalar moves
        // (unit is workgroup-uniform), not by indexing the kernel-argument array (that would spill it to scratch)
        CondArgs a = g.u[2];
        if (unit == 3) a = g.u[3];
        if (unit == 4) a = g.u[4];
        if (unit == 5) a = g.u[5];
        cond_body<128, 128, COND_NW>(a, cond_lds, slot, n);
    }
}

template <class KERNEL>
void cond_allow_lds(KERNEL k, int bytes) {
    pf_allow_lds(reinterpret_cast<const void*>(k), (size_t)bytes);
}

// the five LDS-resident matrices of a unit back to back in the blob, in LDS order?  (fragment pair = 512 floats)
inline int cond_contiguous(const long long* off, int odim) {
    const int HB = odim / 16, MB = odim / 32, HP = (HB + 1) / 2, MP = (MB + 1) / 2;
    long long e = off[0] + (long long)MB * HP * 512;
    if (off[3] != e) return 0;
    e += 12ll * MP * 512;
    if (off[4] != e) return 0;
    e += 8 * 512;
    if (off[6] != e) return 0;
    e += 8 * 512;
    return off[8] == e ? 1 : 0;
}

template <int ODIM, int CDIM>
int launch_cond(CondArgs a, hipStream_t s) {
    constexpr int NW = COND_NW, lds = cond_lds_bytes<ODIM>();
    a.contiguous = cond_contiguous(a.off, ODIM);
    a.ntiles = (a.T + 15) / 16;                            // wave tiles
    const int wgt = (a.ntiles + NW - 1) / NW;
    int per_cu = (160 * 1024) / lds;                       // persistent workgroups that fit one CU's LDS
    const int by_waves = 32 / NW;
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu < 1) per_cu = 1;
    const int cap = 256 * per_cu;
    const int grid = wgt < cap ? wgt : cap;
    cond_allow_lds(cond_kernel<ODIM, CDIM>, lds);
    hipLaunchKernelGGL((cond_kernel<ODIM, CDIM>), dim3(grid), dim3(NW * 64), lds, s, a);
    return pf_last_launch_status();
}

}  // namespace

// unit: 0..4 selects (ODIM, ROWS) = (32, 256) (64, 512) (128, 512) x 3.  w: blob base; off[13]: POST_SLOTS.
extern "C" int pf_pq_gemm(int unit, const float* h, const float* w, const long long* off, float* pq_next, int T, void* stream) {
    if (!h || !w || !off || !pq_next) return PF_ERR_NULL;
    if (T <= 0) return PF_ERR_SHAPE;
    PqArgs a{};
    a.h = h; a.w = reinterpret_cast<const u4*>(w + off[10]); a.bias = w + off[11]; a.scales = w + off[12]; a.pq = pq_next; a.T = T;
    hipStream_t s = (hipStream_t)stream;
    switch (unit) {
        case 0: return launch_pq<32, 256>(a, s);
        case 1: return launch_pq<64, 512>(a, s);
        case 2: case 3: case 4: return launch_pq<128, 512>(a, s);
        default: return PF_ERR_UNSUPPORTED;
    }
}

// ctx [T, 288] = c [T, cd] Hc^T + hb with Hc as an f16n fragment image (packing.pack_cnf_context; the products carry its
// power-of-two scale, inv_scale takes it out) - modules/continuous/diffeq_layers.py:72-86: everything a ConcatSquash layer
// takes from the context, for all three layers of a block in one GEMM.  cd: 32, 64 or 128.
extern "C" int pf_cnf_context(const float* c, int cd, const float* hc_image, const float* hb, float inv_scale, float* ctx, int T,
                              void* stream) {
    if (!c || !hc_image || !hb || !ctx) return PF_ERR_NULL;
    if (T <= 0) return PF_ERR_SHAPE;
    PqArgs a{};
    a.h = c; a.w = reinterpret_cast<const u4*>(hc_image); a.bias = hb; a.scales = nullptr; a.inv = inv_scale; a.pq = ctx; a.T = T;
    hipStream_t s = (hipStream_t)stream;
    switch (cd) {
        case 32: return launch_ctx<32>(a, s);
        case 64: return launch_ctx<64>(a, s);
        case 128: return launch_ctx<128>(a, s);
        default: return PF_ERR_UNSUPPORTED;
    }
}

// unit: 0..5 selects (ODIM, CDIM) = (32,32) (64,64) (128,128) x 4.
extern "C" int pf_cond(int unit, const float* h, const float* w, const long long* off, float* c, float* st, float* cp,
                       int T, void* stream) {
    if (!h || !w || !off || !st || !cp) return PF_ERR_NULL;
    if (T <= 0) return PF_ERR_SHAPE;
    CondArgs a{};
    a.h = h; a.w = w; a.c = c; a.st = st; a.cp = cp; a.T = T;
    for (int i = 0; i < 13; ++i) a.off[i] = off[i];
    hipStream_t s = (hipStream_t)stream;
    switch (unit) {
        case 0: return launch_cond<32, 32>(a, s);
        case 1: return launch_cond<64, 64>(a, s);
        case 2: case 3: case 4: case 5: return launch_cond<128, 128>(a, s);
        default: return PF_ERR_UNSUPPORTED;
    }
}

// The conditioner stages of all six units in one launch.  h[u]: unit u's EdgeConv output [T, odim_u]; c[u] nullable (all or
// none); st [6][T][8], cp [6][T][64] unit-major as the flow kernels read them; off: 6 x 13 offsets (POST_SLOTS per unit).
extern "C" int pf_cond_all(const float* const* h, const float* w, const long long* off, float* const* c, float* st, float* cp,
                           int T, void* stream) {
    if (!h || !w || !off || (!st != !cp) || (!st && !c)) return PF_ERR_NULL;      // st = cp = NULL: only the conditioning features c
    if (T <= 0) return PF_ERR_SHAPE;
    CondAllArgs g{};
    const int ntiles = (T + 15) / 16;                      // wave tiles
    const int wgt = (ntiles + COND_NW - 1) / COND_NW;      // workgroups that give every wave one tile
    // workgroups per unit: units 0 / 1 carry about half the work of a 128-channel unit; one workgroup per CU (LDS).  Small
    // batches (a 128-channel unit fits one round of 50 workgroups): the narrow units take two rounds of their cheaper body
    static const int share_big[6] = {32, 32, 48, 48, 48, 48}, share_small[6] = {24, 32, 50, 50, 50, 50};
    const int* share = wgt <= 50 ? share_small : share_big;
    g.first[0] = 0;
    for (int u = 0; u < 6; ++u) {
        if (!h[u]) return PF_ERR_NULL;
        CondArgs& a = g.u[u];
        a.h = h[u]; a.w = w; a.c = c ? c[u] : nullptr; a.st = st ? st + (size_t)u * T * 8 : nullptr; a.cp = cp ? cp + (size_t)u * T * 64 : nullptr; a.T = T;
        a.ntiles = ntiles;
        for (int i = 0; i < 13; ++i) a.off[i] = off[u * 13 + i];
        a.contiguous = cond_contiguous(a.off, u == 0 ? 32 : (u == 1 ? 64 : 128));
        const int n = wgt < share[u] ? wgt : share[u];
        g.first[u + 1] = g.first[u] + n;
    }
    constexpr int lds = cond_lds_bytes<128>();
    cond_allow_lds(cond_all_kernel, lds);
    hipLaunchKernelGGL(cond_all_kernel, dim3(g.first[6]), dim3(COND_NW * 64), lds, (hipStream_t)stream, g);
    return pf_last_launch_status();
}

// Both stages of unit `unit` (the round-1 entry point, kept for callers that want one call per unit).
extern "C" int pf_post(int unit, const float* h, const float* w, const long long* off, float* c, float* st, float* cp,
                       float* pq_next, int T, void* stream) {
    if (unit < 5) {
        if (!pq_next) return PF_ERR_NULL;
        const int rc = pf_pq_gemm(unit, h, w, off, pq_next, T, stream);
        if (rc != PF_OK) return rc;
    }
    return pf_cond(unit, h, w, off, c, st, cp, T, stream);
}
