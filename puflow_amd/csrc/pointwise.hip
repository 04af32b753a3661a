// Per-point stage that follows each EdgeConv unit (one launch per unit):
//   c   = W2 relu(W1 h + b1)                         FeatMergeUnit   (interpflow.py:251-258)
//   (c itself is written only on request: W2 is folded into the first layers of the three nets below)
//   s,t = LinearA1D_s(c), LinearA1D_t(c)             AffineInjectorLayer nets (coupling.py:132-134,
//                                                    interpflow.py:22-43) - depend on c only, so they
//                                                    are computed once per ORIGINAL point and shared by
//                                                    f and by the R replicas of g
//   cp  = W0[:, tdim:] c                             c-part of coupling1's first layer (interpflow.py:38-41)
//   PQ' = Wpq h + bpq                                next unit's per-point EdgeConv vectors (packing.py)
// MFMA columns = 16 points; all intermediates stay in registers (pf_mfma.h layout).  Arithmetic: split-fp16
// products (pf_mfma.h "f16x2": three fp16 MFMAs per 32-channel step, fp32-class accuracy); weights arrive
// pre-split from the host (packing.frag_pack_f16x2) and stream through buffer loads (they stay in L1/L2).
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"
#include "pf_mfma.h"

#ifndef PF_POST_P
#define PF_POST_P 2
#endif
#ifndef PF_POST_NW
#define PF_POST_NW 4
#endif

namespace {

struct PostArgs {
    const float* h;                       // [T, ODIM]
    const f4* wM1; const float* b1;       // [ODIM/2, ODIM]
    const f4* wM2;                        // [CDIM, ODIM/2]
    const f4* wH1;                        // [192, ODIM/2]  rows: (s_W0 | t_W0 | c1_W0c) W2  (merge conv2 folded in)
    const f4* wS2; const float* bS2;      // [64, 64]
    const f4* wT2; const float* bT2;      // [64, 64]
    const f4* wST4; const float* bST4;    // [16, 128]    rows 0-2: s_W4 on cols 0-63, rows 3-5: t_W4 on cols 64-127
    const f4* wPQ; const float* bPQ;      // [2*SNEXT, ODIM]  (unused when SNEXT == 0)
    float* c;                             // [T, CDIM]   (nullable)
    float* st;                            // [T, 8]      s0 s1 s2 t0 t1 t2 - -
    float* cp;                            // [T, 64]
    float* pq;                            // [T, 2*SNEXT]
    int T, ntiles;
};

template <int ODIM, int CDIM, int SNEXT, int P, int NW>
__global__ __launch_bounds__(NW * 64) void post_kernel(PostArgs a) {
    constexpr int HB = ODIM / 16, MB = ODIM / 32, CB = CDIM / 16;
    constexpr int HP = (HB + 1) / 2, MP = (MB + 1) / 2;              // block pairs (32 input channels per MFMA step)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    const PfW2Buf wsPQ(a.wPQ, lane), wsM1(a.wM1, lane), wsM2(a.wM2, lane), wsH1(a.wH1, lane), wsS2(a.wS2, lane),
        wsT2(a.wT2, lane), wsST4(a.wST4, lane);

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int pt0 = (tile * NW + wave) * P * 16;
        int pt[P];
        bool ok[P];
        PfPair2 hp[P][HP];
        {
            f4 h[P][HB];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int g = pt0 + p * 16 + col;
                ok[p] = g < a.T;
                pt[p] = ok[p] ? g : a.T - 1;
#pragma unroll
                for (int b = 0; b < HB; ++b)
                    h[p][b] = *reinterpret_cast<const f4*>(a.h + (size_t)pt[p] * ODIM + b * 16 + 4 * q);
            }
            pf_pairs2<HB>(h, 0, hp);
        }

        // ---- next unit's P|Q vectors
        if constexpr (SNEXT > 0) {
            pf_static_for<0, (2 * SNEXT) / 32>([&](auto cc) {
                constexpr int ob0 = decltype(cc)::value * 2;
#ifdef PF_POST_SYNC
                if constexpr (decltype(cc)::value % PF_POST_SYNC == 0) __syncthreads();   // waves stream the weights in step -> L1 hits
#endif
                f4 acc[P][2];
#pragma unroll
                for (int o = 0; o < 2; ++o)
#pragma unroll
                    for (int p = 0; p < P; ++p) acc[p][o] = pf_bias(a.bPQ, ob0 + o, q);
                pf_mm2f<2, HP, HP>(wsPQ, ob0 * HP, hp, 0, acc, 0);
#pragma unroll
                for (int o = 0; o < 2; ++o)
#pragma unroll
                    for (int p = 0; p < P; ++p)
                        if (ok[p])
                            *reinterpret_cast<f4*>(a.pq + (size_t)pt[p] * (2 * SNEXT) + (ob0 + o) * 16 + 4 * q) = acc[p][o];
            });
        }

        // ---- merge MLP
        PfPair2 mp[P][MP];
        {
            f4 m[P][MB];
#pragma unroll
            for (int o = 0; o < MB; ++o)
#pragma unroll
                for (int p = 0; p < P; ++p) m[p][o] = pf_bias(a.b1, o, q);
            pf_mm2f<MB, HP, HP>(wsM1, 0, hp, 0, m, 0);
#pragma unroll
            for (int o = 0; o < MB; ++o)
#pragma unroll
                for (int p = 0; p < P; ++p) m[p][o] = pf_relu(m[p][o]);
            pf_pairs2<MB>(m, 0, mp);
        }

        // c = W2 m is only materialised when the caller wants `cs` (feat_extract API): everything downstream of c
        // is linear in it, so the host folded W2 into those layers (H1 := [s_W0; t_W0; c1_W0c] W2, packing.pack_plan)
        if (a.c) {
            f4 c[P][CB];
#pragma unroll
            for (int o = 0; o < CB; ++o)
#pragma unroll
                for (int p = 0; p < P; ++p) c[p][o] = pf_splat(0.f);
            pf_mm2f<CB, MP, MP>(wsM2, 0, mp, 0, c, 0);
#pragma unroll
            for (int o = 0; o < CB; ++o)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    if (ok[p]) *reinterpret_cast<f4*>(a.c + (size_t)pt[p] * CDIM + o * 16 + 4 * q) = c[p][o];
        }

        // ---- coupling1 c-part (rows 128..191 of H1), stored raw
        {
            f4 acc[P][4];
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p][o] = pf_splat(0.f);
            pf_mm2f<4, MP, MP>(wsH1, 8 * MP, mp, 0, acc, 0);
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    if (ok[p]) *reinterpret_cast<f4*>(a.cp + (size_t)pt[p] * 64 + o * 16 + 4 * q) = acc[p][o];
        }

        // ---- injector nets: hidden1 -> hidden2 for s (pairs 0..1) and t (pairs 2..3)
        PfPair2 h2p[P][4];
        pf_static_for<0, 2>([&](auto nc) {
            constexpr int net = decltype(nc)::value;
            f4 h1[P][4];
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int p = 0; p < P; ++p) h1[p][o] = pf_splat(0.f);
            pf_mm2f<4, MP, MP>(wsH1, (4 * net) * MP, mp, 0, h1, 0);
            f4 h2[P][4];
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    h1[p][o] = pf_lrelu(h1[p][o], 0.01f);
                    h2[p][o] = pf_bias(net == 0 ? a.bS2 : a.bT2, o, q);
                }
            PfPair2 h1p[P][2];
            pf_pairs2<4>(h1, 0, h1p);
            if constexpr (net == 0) pf_mm2f<4, 2, 2>(wsS2, 0, h1p, 0, h2, 0); else pf_mm2f<4, 2, 2>(wsT2, 0, h1p, 0, h2, 0);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                h2p[p][2 * net + 0] = pf_pair2(pf_lrelu(h2[p][0], 0.01f), pf_lrelu(h2[p][1], 0.01f));
                h2p[p][2 * net + 1] = pf_pair2(pf_lrelu(h2[p][2], 0.01f), pf_lrelu(h2[p][3], 0.01f));
            }
        });
        {
            f4 acc[P][1];
#pragma unroll
            for (int p = 0; p < P; ++p) acc[p][0] = pf_bias(a.bST4, 0, q);
            pf_mm2f<1, 4, 4>(wsST4, 0, h2p, 0, acc, 0);
#pragma unroll
            for (int p = 0; p < P; ++p)
                if (ok[p] && q < 2) *reinterpret_cast<f4*>(a.st + (size_t)pt[p] * 8 + 4 * q) = acc[p][0];
        }
    }
}

template <int ODIM, int CDIM, int SNEXT>
int launch(PostArgs a, hipStream_t s) {
    constexpr int P = PF_POST_P, NW = PF_POST_NW;
    a.ntiles = (a.T + NW * P * 16 - 1) / (NW * P * 16);
    int grid = a.ntiles < 2048 ? a.ntiles : 2048;
    hipLaunchKernelGGL((post_kernel<ODIM, CDIM, SNEXT, P, NW>), dim3(grid), dim3(NW * 64), 0, s, a);
    return pf_last_launch_status();
}

}  // namespace

// unit: 0..5 selects (ODIM, CDIM, SNEXT) = (32,32,128) (64,64,256) (128,128,256)x3 (128,128,0).
// w: blob base; off[12]: float offsets of M1, b1, M2, H1, S2, bS2, T2, bT2, ST4, bST4, PQ, bPQ.
extern "C" int pf_post(int unit, const float* h, const float* w, const long long* off, float* c, float* st, float* cp,
                       float* pq_next, int T, void* stream) {
    if (!h || !w || !off || !st || !cp) return PF_ERR_NULL;
    if (T <= 0) return PF_ERR_SHAPE;
    PostArgs a{};
    a.h = h;
    a.wM1 = reinterpret_cast<const f4*>(w + off[0]); a.b1 = w + off[1];
    a.wM2 = reinterpret_cast<const f4*>(w + off[2]);
    a.wH1 = reinterpret_cast<const f4*>(w + off[3]);
    a.wS2 = reinterpret_cast<const f4*>(w + off[4]); a.bS2 = w + off[5];
    a.wT2 = reinterpret_cast<const f4*>(w + off[6]); a.bT2 = w + off[7];
    a.wST4 = reinterpret_cast<const f4*>(w + off[8]); a.bST4 = w + off[9];
    a.wPQ = reinterpret_cast<const f4*>(w + off[10]); a.bPQ = w + off[11];
    a.c = c; a.st = st; a.cp = cp; a.pq = pq_next; a.T = T;
    hipStream_t s = (hipStream_t)stream;
    if (unit < 5 && !pq_next) return PF_ERR_NULL;
    switch (unit) {
        case 0: return launch<32, 32, 128>(a, s);
        case 1: return launch<64, 64, 256>(a, s);
        case 2: case 3: case 4: return launch<128, 128, 256>(a, s);
        case 5: return launch<128, 128, 0>(a, s);
        default: return PF_ERR_UNSUPPORTED;
    }
}
