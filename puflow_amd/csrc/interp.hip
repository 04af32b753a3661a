// Interpolation module, fully fused: kNN-8 context features (DistanceEncoder MLP || EdgeConv
// without pooling) -> WeightEstimationUnit -> softmax over the 8 neighbours of the first R
// channels -> weighted sum of the neighbours' latents.  Replaces InterpolationModule.forward and
// its sub-modules (modules/discrete/interpflow.py:85-186) in eval mode.
//
// Mapping: one MFMA column tile = 2 points x 8 neighbours (col = 8*ps + k).  The 256-channel
// context never exists, not even in registers: both producers end in a linear layer (DistanceEncoder's
// last conv, the EdgeConv's conv_out) that feeds the weight unit's first conv with no nonlinearity in
// between (interpflow.py:134,144), so the host composes them (packing.fold_state_dict):
//     w1 = (W0a W6) d2 + (W0b Gout) feat + [W0b (PA x_i + QB x_j + pb) + b0 + W0a b6]
// Every term that is affine in the edge's raw inputs e = (x_i, x_j, |x_i - x_j|, 1) - the distance encoder's first
// layer, the EdgeConv pre-activations, the bracket above - is one MFMA step against an 8-column "edge table"
// (e occupies 8 of the 32 k-slots).  Only the first R rows of the last weight conv are computed (interpflow.py:180).
// Arithmetic: split-fp16 products with a natural-scale low half (pf_mfma.h "f16n": one accumulator, no fold), fp32-class
// accuracy; 91 fragment pairs = 273 fp16 MFMAs per tile where the unfolded f32 formulation needs 1216.  Matrices that
// share an accumulator share a power-of-two scale (packing.INTERP_SCALES); the kernel multiplies by its inverse.
//
// Weight blob (float offsets in `off[]`, see puflow_amd/packing.py::INTERP_SLOTS); matrices are f16x2 fragment images:
//   0 dtab [64 x e]        1 d_W3 [64 x 64]     2 d_b3 [64] f32      3 (W0a W6) [128 x 64]    4 scales [6] f32
//   5 ectab [128 x e]      6 ec G1..G7 (16 pair fragments; layer t starts at fragment floor(t/2) * ceil(t/2))
//   7 (W0b Gout) [128 x 128]   8 w1tab [128 x e] (bracket above)   9 w_W3 [64 x 128]  10 w_b3 [64] f32
//   11 w_W6 [16 x 64] (rows 0..3 replicated per q group: the R <= 4 fast path)   12 w_b6 [16] f32
//   13 w_W6 [32 x 64] all r_max = 32 rows (R > 4: WeightEstimationUnit supports up to 32, interpflow.py:142)   14 w_b6 [32] f32
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"
#include "pf_mfma.h"

// tuned on MI355X with tools/tune_variants.py (P = column tiles per wave, NW = waves per workgroup)
#ifndef PF_INTERP_P
#define PF_INTERP_P 1
#endif
#ifndef PF_INTERP_NW
#define PF_INTERP_NW 12
#endif

namespace {

struct InterpArgs {
    const float* xyz;    // [T,3]
    const float* z;      // [T,3]
    const int* idx;      // [T,16]  (first 8 used)
    const float* w;
    long long off[15];
    float* u;            // [T*R,3]  row = n*R + r
    float* aw;           // non-null (R <= 4 only): write the softmax weights [T][8 neighbours][4 rows] INSTEAD of u (z unused):
                         // the weighted latent sum then runs inside flow g (pf_flow_inv_interp), and this kernel no longer waits for z
    int T, N, ntiles, R;     // ntiles: WAVE tiles of 2 P points; R = upsampling ratio actually written (1..4; BIG: 5..32)
    int per;                 // wave tiles per workgroup
    int contiguous;          // the seven LDS-resident matrices lie back to back in the blob, in LDS order: one copy
};

template <int P, int NW, bool BIG = false>     // BIG: upsampling ratios 5..32 (all 32 rows of the last weight conv)
__global__ __launch_bounds__(NW * 64) void interp_kernel(InterpArgs a) {
    constexpr int R = 4;                 // rows computed (W6 is packed with 4 rows per q group); a.R <= 4 are stored
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    const int ps = col >> 3, k = col & 7;
    // everything but the 64 KiB (W0b Gout) image lives in LDS for the whole persistent workgroup: 76 fragments x 2 KiB
    // = 152 KiB (edge tables, d_W3, the 7-step growth chain, (W0a W6), w_W3); (W0b Gout) streams from L2, 4 fragments ahead
    constexpr int L_DT = 0, L_ET = 4, L_WT = 12, L_D3 = 20, L_EC = 28, L_D6 = 44, L_W3 = 60, L_END = 76;
    __shared__ u4 wl[L_END * 128];
    {
        auto stage = [&](int f0, int nf, long long off) {
            const u4* src = reinterpret_cast<const u4*>(a.w + off);
            pf_stage_lds(wl + f0 * 128, src, nf * 128);
        };
        if (a.contiguous) stage(L_DT, L_END, a.off[0]);
        else {
            stage(L_DT, 4, a.off[0]); stage(L_ET, 8, a.off[5]); stage(L_WT, 8, a.off[8]); stage(L_D3, 8, a.off[1]);
            stage(L_EC, 16, a.off[6]); stage(L_D6, 16, a.off[3]); stage(L_W3, 16, a.off[9]);
        }
        __syncthreads();
    }
    const PfW2Lds wsDT{wl + L_DT * 128, lane}, wsET{wl + L_ET * 128, lane}, wsWT{wl + L_WT * 128, lane},
        wsD3{wl + L_D3 * 128, lane}, wsEC{wl + L_EC * 128, lane}, wsD6{wl + L_D6 * 128, lane}, wsW3{wl + L_W3 * 128, lane};
    const PfW2BufD<4> wsW0(a.w + a.off[7], lane);
    const PfW2BufD<2> wsW6(a.w + a.off[BIG ? 13 : 11], lane);

    const float* sc = a.w + a.off[4];
    const float iDT = sc[0], iD3 = sc[1], iW1 = sc[2], iEC = sc[3], iW3 = sc[4], iW6 = sc[5];

    // a workgroup owns `per` consecutive WAVE tiles (2 P points each), its waves take them round-robin: every CU gets the same
    // number of wave tiles (4 x 2048 points: 16 per CU = one full round + four waves, instead of two full rounds on a third
    // of the CUs with whole-workgroup tiles)
    const int wt_end = min((int)(blockIdx.x + 1) * a.per, a.ntiles);
    for (int wt = blockIdx.x * a.per + wave; wt < wt_end; wt += NW) {
        const int pt0 = wt * P * 2;
        int gi[P], gj[P];
        bool ok[P];
        PfPairN e[P][1];                 // raw edge inputs (x_i, x_j, |x_i - x_j|, 1) in k-slots 0..7 (lanes q = 0)
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int g = pt0 + p * 2 + ps;
            ok[p] = g < a.T;
            gi[p] = ok[p] ? g : a.T - 1;
            gj[p] = (gi[p] / a.N) * a.N + a.idx[(size_t)gi[p] * 16 + k];
            float xi[3], xj[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                xi[c] = a.xyz[(size_t)gi[p] * 3 + c];
                xj[c] = a.xyz[(size_t)gj[p] * 3 + c];
            }
            const float v0 = xi[0] - xj[0], v1 = xi[1] - xj[1], v2 = xi[2] - xj[2];
            const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(v0, v0), __fmul_rn(v1, v1)), __fmul_rn(v2, v2)));
            const f4 z4 = pf_splat(0.f);
            const f4 e0 = {xi[0], xi[1], xi[2], xj[0]}, e1 = {xj[1], xj[2], nrm, 1.f};
            e[p][0] = pf_pairn(q == 0 ? e0 : z4, q == 0 ? e1 : z4);
        }

        // ---- EdgeConv growth features (C=3, g=16, 8 convs) on the same 8 neighbours; conv_out is folded into w1
        // (first: its 8-layer dependent chain sets the tile's latency and needs the fewest live registers)
        f4 w1[P][8];
        {
            PfPairN fp[P][4];
            f4 last[P];
            {
                f4 acc[P][1];
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p][0] = pf_splat(0.f);
                pf_mmn<false, 1, 1, 1>(wsET, 0, e, acc);                 // pre-activation from the raw inputs (edge table)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    last[p] = pf_lrelu(acc[p][0] * iEC, 0.05f);
                    fp[p][0] = pf_pairn(last[p], pf_splat(0.f));
                }
            }
            pf_static_for<1, 8>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int CPT = (t + 1) / 2;                     // input pairs of growth layer t
                constexpr int F0 = (t / 2) * ((t + 1) / 2);          // fragments of layers 1..t-1: sum ceil(s/2)
                f4 acc[P][1];
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p][0] = pf_splat(0.f);
                pf_mmn<false, 1, 1, 1>(wsET, t, e, acc);
                pf_mmn<false, 1, CPT, CPT>(wsEC, F0, fp, acc);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const f4 f = pf_lrelu(acc[p][0] * iEC, 0.05f);
                    if constexpr (t % 2 == 1) fp[p][t / 2] = pf_pairn(last[p], f);
                    else fp[p][t / 2] = pf_pairn(f, pf_splat(0.f));
                    last[p] = f;
                }
            });
            // w1 = [b0 + W0a b6 + W0b.(edge table)] e + (W0b Gout) feat
#pragma unroll
            for (int ob = 0; ob < 8; ++ob)
#pragma unroll
                for (int p = 0; p < P; ++p) w1[p][ob] = pf_splat(0.f);
            pf_mmn<false, 4, 1, 1>(wsWT, 0, e, w1, 0, 0);
            pf_mmn<false, 4, 1, 1>(wsWT, 4, e, w1, 0, 4);
            pf_mmn<false, 4, 4, 4>(wsW0, 0, fp, w1, 0, 0);            // two chunks of 4 output blocks
            pf_mmn<false, 4, 4, 4>(wsW0, 16, fp, w1, 0, 4);
        }

        // ---- distance encoder 10 -> 64 -> 64, then its (folded) contribution to the weight unit's first layer
        {
            f4 d[P][4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int p = 0; p < P; ++p) d[p][cb] = pf_splat(0.f);
            pf_mmn<false, 4, 1, 1>(wsDT, 0, e, d);
            PfPairN dp[P][2];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                dp[p][0] = pf_pairn(pf_lrelu(d[p][0] * iDT, 0.01f), pf_lrelu(d[p][1] * iDT, 0.01f));
                dp[p][1] = pf_pairn(pf_lrelu(d[p][2] * iDT, 0.01f), pf_lrelu(d[p][3] * iDT, 0.01f));
            }
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int p = 0; p < P; ++p) d[p][cb] = pf_bias(a.w + a.off[2], cb, q);
            pf_mmn<false, 4, 2, 2>(wsD3, 0, dp, d);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                dp[p][0] = pf_pairn(pf_lrelu(d[p][0] * iD3, 0.01f), pf_lrelu(d[p][1] * iD3, 0.01f));
                dp[p][1] = pf_pairn(pf_lrelu(d[p][2] * iD3, 0.01f), pf_lrelu(d[p][3] * iD3, 0.01f));
            }
            // w1 += (W0a W6) d2
            pf_mmn<false, 4, 2, 2>(wsD6, 0, dp, w1, 0, 0);
            pf_mmn<false, 4, 2, 2>(wsD6, 8, dp, w1, 0, 4);
        }

        // ---- rest of the weight unit: 128 -> 64 -> R
        f4 w3[P][BIG ? 2 : 1];
        {
            PfPairN w1p[P][4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    w1p[p][c] = pf_pairn(pf_lrelu(w1[p][2 * c] * iW1, 0.01f), pf_lrelu(w1[p][2 * c + 1] * iW1, 0.01f));
            f4 w2[P][4];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int p = 0; p < P; ++p) w2[p][ob] = pf_bias(a.w + a.off[10], ob, q);
            pf_mmn<false, 4, 4, 4>(wsW3, 0, w1p, w2);
            PfPairN w2p[P][2];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                w2p[p][0] = pf_pairn(pf_lrelu(w2[p][0] * iW3, 0.01f), pf_lrelu(w2[p][1] * iW3, 0.01f));
                w2p[p][1] = pf_pairn(pf_lrelu(w2[p][2] * iW3, 0.01f), pf_lrelu(w2[p][3] * iW3, 0.01f));
            }
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int ob = 0; ob < (BIG ? 2 : 1); ++ob) w3[p][ob] = pf_bias(a.w + a.off[BIG ? 14 : 12], ob, q);
            pf_mmn<false, (BIG ? 2 : 1), 2, 2>(wsW6, 0, w2p, w3);
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int ob = 0; ob < (BIG ? 2 : 1); ++ob) w3[p][ob] = w3[p][ob] * iW6;
        }

        // ---- softmax over the 8 neighbours (lanes k = 0..7 of the point), then weighted latent sum
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if constexpr (BIG) {
                // lane (k, q) owns rows 16 ob + 4 q + r of the weight tensor; every lane sums all three latent channels
                float zj[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) zj[c] = a.z[(size_t)gj[p] * 3 + c];
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * ob + 4 * q + r;
                        const float x = w3[p][ob][r];
                        float m = x;
                        m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
                        const float ex = expf(x - m);
                        float sm = ex;
                        sm += __shfl_xor(sm, 1); sm += __shfl_xor(sm, 2); sm += __shfl_xor(sm, 4);
                        const float av = ex / sm;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            float s = av * zj[c];
                            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
                            if (ok[p] && k == 0 && row < a.R) a.u[((size_t)gi[p] * a.R + row) * 3 + c] = s;
                        }
                    }
            } else {
                float av[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float x = w3[p][0][r];
                    float m = x;
                    m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
                    const float ex = expf(x - m);
                    float s = ex;
                    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
                    av[r] = ex / s;
                }
                if (a.aw) {                                                      // weights only (uniform branch)
                    if (ok[p] && q == 0) {
                        f4 o = {av[0], av[1], av[2], av[3]};
                        *reinterpret_cast<f4*>(a.aw + ((size_t)gi[p] * 8 + k) * 4) = o;
                    }
                    continue;
                }
                const float zj = a.z[(size_t)gj[p] * 3 + (q < 3 ? q : 0)];      // lane q handles latent channel q
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float s = av[r] * zj;
                    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
                    if (ok[p] && k == 0 && q < 3 && r < a.R) a.u[((size_t)gi[p] * a.R + r) * 3 + q] = s;
                }
            }
        }
    }
}

}  // namespace

static int interp_launch(const float* xyz, const float* z, const int* idx16, const float* w, const long long* off,
                         float* u_out, float* aw_out, int B, int N, int R, void* stream) {
    if (B <= 0 || N < 8 || (long long)B * N > (1ll << 28)) return PF_ERR_SHAPE;
    if (R < 1 || R > 32) return PF_ERR_UNSUPPORTED;          // r_max of WeightEstimationUnit (interpflow.py:142)
    constexpr int NWB = 8;                                     // the R > 4 variant needs more registers: 2 waves per SIMD
    InterpArgs a{};
    a.xyz = xyz; a.z = z; a.idx = idx16; a.w = w; a.u = u_out; a.aw = aw_out; a.T = B * N; a.N = N; a.R = R;
    for (int i = 0; i < 15; ++i) a.off[i] = off[i];
    a.contiguous = off[5] == off[0] + 4 * 512 && off[8] == off[5] + 8 * 512 && off[1] == off[8] + 8 * 512 &&
                   off[6] == off[1] + 8 * 512 && off[3] == off[6] + 16 * 512 && off[9] == off[3] + 16 * 512;
    // launch shape: one column tile per wave x 12 waves.  (2, 8) - two tiles share each weight fragment read, half the LDS / L2
    // weight traffic per tile - measures 169 -> 160 us in a back-to-back loop of this kernel alone (tools/tune_interp.py) and
    // changes nothing inside the step (191 us either way): the kernel is bound by its dependent MFMA -> VALU -> MFMA chains
    // (four per SIMD in both shapes), not by weight traffic.  The same holds for cond_all_kernel at (8, 2): 83 us either way.
    auto go = [&](auto pc, auto nwc, auto bigc) {
        constexpr int P = decltype(pc)::value, NW = decltype(nwc)::value;
        constexpr bool BIG = decltype(bigc)::value;
        a.ntiles = (a.T + P * 2 - 1) / (P * 2);
        const int wgt = (a.ntiles + NW - 1) / NW;
        const int grid = wgt < 256 ? wgt : 256;               // persistent: 152 KiB of LDS = one workgroup per CU
        a.per = (a.ntiles + grid - 1) / grid;
        hipLaunchKernelGGL((interp_kernel<P, NW, BIG>), dim3(grid), dim3(NW * 64), 0, (hipStream_t)stream, a);
    };
    if (R > 4) go(std::integral_constant<int, 1>{}, std::integral_constant<int, NWB>{}, std::true_type{});
    else go(std::integral_constant<int, PF_INTERP_P>{}, std::integral_constant<int, PF_INTERP_NW>{}, std::false_type{});
    return pf_last_launch_status();
}

extern "C" int pf_interp(const float* xyz, const float* z, const int* idx16, const float* w, const long long* off,
                         float* u_out, int B, int N, int R, void* stream) {
    if (!xyz || !z || !idx16 || !w || !off || !u_out) return PF_ERR_NULL;
    return interp_launch(xyz, z, idx16, w, off, u_out, nullptr, B, N, R, stream);
}

// The interpolation weights alone (everything of pf_interp but the weighted latent sum): aw [B*N][8][4] = softmax over the 8
// neighbours of the first 4 rows of the weight unit's output (interpflow.py:180).  A function of xyz and the neighbour lists
// only - it can run beside the feature extractor / flow f chain; pf_flow_inv_interp consumes it.
extern "C" int pf_interp_weights(const float* xyz, const int* idx16, const float* w, const long long* off, float* aw_out,
                                 int B, int N, void* stream) {
    if (!xyz || !idx16 || !w || !off || !aw_out) return PF_ERR_NULL;
    return interp_launch(xyz, nullptr, idx16, w, off, nullptr, aw_out, B, N, 4, stream);
}

