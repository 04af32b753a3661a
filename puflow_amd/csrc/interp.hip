// Interpolation module, fully fused: kNN-8 context features (DistanceEncoder MLP || EdgeConv
// without pooling) -> WeightEstimationUnit -> softmax over the 8 neighbours of the first R
// channels -> weighted sum of the neighbours' latents.  Replaces InterpolationModule.forward and
// its sub-modules (modules/discrete/interpflow.py:85-186) in eval mode.
//
// Mapping: one MFMA column tile = 2 points x 8 neighbours (col = 8*ps + k).  The 256-channel
// context never exists, not even in registers: both producers end in a linear layer (DistanceEncoder's
// last conv, the EdgeConv's conv_out) that feeds the weight unit's first conv with no nonlinearity in
// between (interpflow.py:134,144), so the host composes them (packing.fold_state_dict):
//     w1 = [b0 + W0a b6] + (W0a W6) d2 + (W0b Gout) feat + W0b (PA x_i + QB x_j + pb)
// -> 704 MFMAs per tile instead of 1216.  Only the first R rows of the last weight conv are
// computed (interpflow.py:180).
//
// Weight blob (float offsets in `off[]`, see puflow_amd/packing.py::INTERP_SLOTS):
//   0 dtab [64][8] (PA(3) QB(3) wn b0)   1 d_W3 frags [4x4]  2 d_b3 [64]  3 (W0a W6) frags [8x4]  4 b0 + W0a b6 [128]
//   5 ectab [256][8] (PA(3) QB(3) pb 0; rows 0..127 used)    6 ec G1..G7 frags   7 (W0b Gout) frags [8x8]
//   8 w1tab [128][8] = W0b.(edge table rows 128..255)        9 w_W3 frags [4x8]  10 w_b3 [64]
//   11 w_W6 frags [1x4] (R rows replicated per q group)      12 w_b6 [16]
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"
#include "pf_mfma.h"

// tuned on MI355X with tools/tune_variants.py (P = column tiles per wave, NW = waves per workgroup)
#ifndef PF_INTERP_P
#define PF_INTERP_P 2
#endif
#ifndef PF_INTERP_NW
#define PF_INTERP_NW 8
#endif

namespace {

struct InterpArgs {
    const float* xyz;    // [T,3]
    const float* z;      // [T,3]
    const int* idx;      // [T,16]  (first 8 used)
    const float* w;
    long long off[13];
    float* u;            // [T*R,3]  row = n*R + r
    int T, N, ntiles, R;     // R = upsampling ratio actually written (1..4)
};

template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void interp_kernel(InterpArgs a) {
    constexpr int R = 4;                 // rows computed (W6 is packed with 4 rows per q group); a.R <= 4 are stored
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    const int ps = col >> 3, k = col & 7;
    const float* dtab = a.w + a.off[0];
    const float* ectab = a.w + a.off[5];
    const PfWBuf wsD3(a.w + a.off[1], lane), wsD6(a.w + a.off[3], lane), wsEC(a.w + a.off[6], lane),
        wsW0(a.w + a.off[7], lane), wsW3(a.w + a.off[9], lane), wsW6(a.w + a.off[11], lane);

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int pt0 = (tile * NW + wave) * P * 2;
        int gi[P], gj[P];
        bool ok[P];
        float xi[P][3], xj[P][3], nrm[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int g = pt0 + p * 2 + ps;
            ok[p] = g < a.T;
            gi[p] = ok[p] ? g : a.T - 1;
            gj[p] = (gi[p] / a.N) * a.N + a.idx[(size_t)gi[p] * 16 + k];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                xi[p][c] = a.xyz[(size_t)gi[p] * 3 + c];
                xj[p][c] = a.xyz[(size_t)gj[p] * 3 + c];
            }
            const float v0 = xi[p][0] - xj[p][0], v1 = xi[p][1] - xj[p][1], v2 = xi[p][2] - xj[p][2];
            nrm[p] = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(v0, v0), __fmul_rn(v1, v1)), __fmul_rn(v2, v2)));
        }
        // rows [off, off+4) of a [*, 8] table: b + PA.xi + QB.xj (+ wn*|xi-xj| when NRM)
        auto tabrow = [&](const float* tab, int p, int off, bool use_nrm) -> f4 {
            f4 r;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f4* t = reinterpret_cast<const f4*>(tab + (size_t)(off + kk) * 8);
                const f4 t0 = t[0], t1 = t[1];
                float s = use_nrm ? fmaf(t1.z, nrm[p], t1.w) : t1.z;
                s = fmaf(t0.x, xi[p][0], s); s = fmaf(t0.y, xi[p][1], s); s = fmaf(t0.z, xi[p][2], s);
                s = fmaf(t0.w, xj[p][0], s); s = fmaf(t1.x, xj[p][1], s); s = fmaf(t1.y, xj[p][2], s);
                r[kk] = s;
            }
            return r;
        };

        // ---- distance encoder 10 -> 64 -> 64, then its (folded) contribution to the weight unit's first layer
        f4 w1[P][8];
        {
            f4 d1[P][4], d2[P][4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    d1[p][cb] = pf_lrelu(tabrow(dtab, p, cb * 16 + 4 * q, true), 0.01f);
                    d2[p][cb] = pf_bias(a.w + a.off[2], cb, q);
                }
            pf_mm<4, 4, 4>(wsD3, 0, d1, 0, d2, 0);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int p = 0; p < P; ++p) d2[p][cb] = pf_lrelu(d2[p][cb], 0.01f);
            // w1 = (b0 + W0a b6) + W0b.(edge table) + (W0a W6) d2
#pragma unroll
            for (int ob = 0; ob < 8; ++ob)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    w1[p][ob] = pf_bias(a.w + a.off[4], ob, q) + tabrow(a.w + a.off[8], p, ob * 16 + 4 * q, false);
            pf_mm<8, 4, 4>(wsD6, 0, d2, 0, w1, 0);
        }

        // ---- EdgeConv growth features (C=3, g=16, 8 convs) on the same 8 neighbours; conv_out is folded into w1
        {
            f4 feat[P][8];
#pragma unroll
            for (int p = 0; p < P; ++p) feat[p][0] = pf_lrelu(tabrow(ectab, p, 4 * q, false), 0.05f);
            pf_static_for<1, 8>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                f4 acc[P][1];
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p][0] = tabrow(ectab, p, 16 * t + 4 * q, false);
                pf_mm<1, t, t>(wsEC, t * (t - 1) / 2, feat, 0, acc, 0);
#pragma unroll
                for (int p = 0; p < P; ++p) feat[p][t] = pf_lrelu(acc[p][0], 0.05f);
            });
            pf_mm<8, 8, 8>(wsW0, 0, feat, 0, w1, 0);              // w1 += (W0b Gout) feat
        }

        // ---- rest of the weight unit: 128 -> 64 -> R
        f4 w3[P][1];
        {
            f4 w2[P][4];
#pragma unroll
            for (int ob = 0; ob < 8; ++ob)
#pragma unroll
                for (int p = 0; p < P; ++p) w1[p][ob] = pf_lrelu(w1[p][ob], 0.01f);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int p = 0; p < P; ++p) w2[p][ob] = pf_bias(a.w + a.off[10], ob, q);
            pf_mm<4, 8, 8>(wsW3, 0, w1, 0, w2, 0);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int p = 0; p < P; ++p) w2[p][ob] = pf_lrelu(w2[p][ob], 0.01f);
#pragma unroll
            for (int p = 0; p < P; ++p) w3[p][0] = *reinterpret_cast<const f4*>(a.w + a.off[12] + 4 * q);
            pf_mm<1, 4, 4>(wsW6, 0, w2, 0, w3, 0);
        }

        // ---- softmax over the 8 neighbours (lanes k = 0..7 of the point), then weighted latent sum
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float av[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float x = w3[p][0][r];
                float m = x;
                m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
                const float e = expf(x - m);
                float s = e;
                s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
                av[r] = e / s;
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

}  // namespace

extern "C" int pf_interp(const float* xyz, const float* z, const int* idx16, const float* w, const long long* off,
                         float* u_out, int B, int N, int R, void* stream) {
    if (!xyz || !z || !idx16 || !w || !off || !u_out) return PF_ERR_NULL;
    if (B <= 0 || N < 8 || (long long)B * N > (1ll << 28)) return PF_ERR_SHAPE;
    if (R < 1 || R > 4) return PF_ERR_UNSUPPORTED;
    constexpr int P = PF_INTERP_P, NW = PF_INTERP_NW;
    InterpArgs a{};
    a.xyz = xyz; a.z = z; a.idx = idx16; a.w = w; a.u = u_out; a.T = B * N; a.N = N; a.R = R;
    for (int i = 0; i < 13; ++i) a.off[i] = off[i];
    a.ntiles = (a.T + NW * P * 2 - 1) / (NW * P * 2);
    const int grid = a.ntiles < 4096 ? a.ntiles : 4096;
    hipLaunchKernelGGL((interp_kernel<P, NW>), dim3(grid), dim3(NW * 64), 0, (hipStream_t)stream, a);
    return pf_last_launch_status();
}
