// All flow blocks of one direction of the TRAINING step in one launch (forward) and four (backward).
//
// Reference: modules/discrete/interpflow.py:46-82 (FlowBlock.forward / .inverse), :302-321 (PointInterpFlow.f / .g),
// modules/flows/normalize.py:28-54 (ActNorm), permutate.py:77-124 (reverse permutation, invertible 3x3 linear),
// coupling.py:55-137 (additive coupling with the LinearA1D conditioner of interpflow.py:22-43, conditional affine injector).
//
// csrc/train_flow.hip + csrc/train_mlp.hip run a block as 3-4 launches forward and ~10 backward, one autograd node each;
// at 32 x 256 points every one of them is a 4-25 us kernel on a serial chain: ~190 launches and 1.4 ms of a 7.6 ms step for
// work that is ~0.1 ms of arithmetic.  Nothing in the chain couples two points, so here a wave keeps its 16 rows in registers
// through ALL blocks:
//
//   forward   per block: [f] ActNorm + 3x3 linear -> conditioner MLP (64 -> 64 -> 64 -> 1|2, f32 MFMA 16x16x4, channel-major
//             tiles as in train_mlp.hip; 32, 64 or 128 conditioning channels) -> coupling shift -> reverse -> injector;  [g] the exact inverses in reverse order with
//             the conditioning rows shared by the R replicas of a point.  The block's weights are staged in LDS (<= 58 KB) between
//             two workgroup barriers; what the backward needs is stored once (block input, y / v, h1, h2, o).
//   backward  the same walk in the opposite order: injector / coupling / MLP chain / affine backward per row, dz1, dz2, do and
//             dc, ds, dt stored, the 15 parameter-gradient sums of (logs, bias, matrix) left per wave tile;
//             then ONE reduction launch for the 3x3 parameters of all blocks (incl. the W^-1 -> W and log-det terms) and the
//             weight gradients of all conditioner nets as one batched split-K launch + reduction (pf_mlp_train_dw_batch).
//
// Sums are fixed-order (per-tile partials, then strided + tree): results do not depend on scheduling.
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"
#include "pf_mfma.h"

namespace {

constexpr int FC_NW = 4;                         // waves per workgroup, one 16-row tile each
constexpr int FC_LD = 68;                        // LDS row stride of a 64-wide matrix
// forward image of one block's conditioner (floats)
// (W0c [64][cc + 4] sized for cc = 128, W1 [64][68], W2 [16][68], b1, b2, Wx [64][4])
constexpr int FC_CCMAX = 128;
constexpr int FF_W0 = 0, FF_W1 = 64 * (FC_CCMAX + 4), FF_W2 = FF_W1 + 64 * FC_LD, FF_B1 = FF_W2 + 16 * FC_LD, FF_B2 = FF_B1 + 64,
              FF_WX = FF_B2 + 16, FF_FLOATS = FF_WX + 64 * 4;
// backward image: Wt0[u][c] = W0[c][td + u] ([cc][68]), Wt1[u][c] = W1[c][u], Wt2[u][c] = W2[c][u] (c < 16), Wx
constexpr int FB_T0 = 0, FB_T1 = FC_CCMAX * FC_LD, FB_T2 = FB_T1 + 64 * FC_LD, FB_WX = FB_T2 + 64 * 20, FB_FLOATS = FB_WX + 64 * 4;
constexpr float FC_SLOPE = 0.01f;

__device__ __forceinline__ f4 fc_mfma4(f4 a, f4 b, f4 c) {
    c = pf_mfma(a.x, b.x, c); c = pf_mfma(a.y, b.y, c); c = pf_mfma(a.z, b.z, c); c = pf_mfma(a.w, b.w, c);
    return c;
}
__device__ __forceinline__ f4 fc_lrelu(f4 z) {
    f4 r;
    r.x = fmaxf(z.x, z.x * FC_SLOPE); r.y = fmaxf(z.y, z.y * FC_SLOPE); r.z = fmaxf(z.z, z.z * FC_SLOPE); r.w = fmaxf(z.w, z.w * FC_SLOPE);
    return r;
}

// The LDS image of a block's conditioner is written ONCE per call by a small pack kernel (flowchain_pack_kernel) into global
// memory in exactly the layout the chain kernels use; a workgroup then stages a block as one linear float4 copy, fetched into
// registers while the previous block is still being computed (the strided, transposing reads of the parameter tensors cost
// seven dependent memory round trips per block when every workgroup did them itself: 100 us per launch instead of ~30).
__device__ __forceinline__ void fc_image_fwd(float* img, const PfFlowChain& a, int i) {
    const int td = a.td[i], cc = a.cc[i], wo = 3 - td, ld0 = cc + 4;
    for (int k = blockIdx.z * blockDim.x + threadIdx.x; k < FF_FLOATS; k += gridDim.z * blockDim.x) {
        float v = 0.f;
        if (k < FF_W1) {
            const int c = k / ld0, u = k - c * ld0;
            if (c < 64 && u < cc) v = a.w0[i][(size_t)c * (td + cc) + td + u];
        } else if (k < FF_W2) {
            const int c = (k - FF_W1) / FC_LD, u = (k - FF_W1) % FC_LD;
            if (u < 64) v = a.w2[i][c * 64 + u];
        } else if (k < FF_B1) {
            const int c = (k - FF_W2) / FC_LD, u = (k - FF_W2) % FC_LD;
            if (c < wo && u < 64) v = a.w4[i][c * 64 + u];
        } else if (k < FF_B2) v = a.b2[i][k - FF_B1];
        else if (k < FF_WX) { if (k - FF_B2 < wo) v = a.b4[i][k - FF_B2]; }
        else {
            const int c = (k - FF_WX) >> 2, j = (k - FF_WX) & 3;
            if (j < td) v = a.w0[i][(size_t)c * (td + cc) + j];
        }
        img[k] = v;
    }
}
__device__ __forceinline__ void fc_image_bwd(float* img, const PfFlowChain& a, int i) {
    const int td = a.td[i], cc = a.cc[i], wo = 3 - td;
    for (int k = blockIdx.z * blockDim.x + threadIdx.x; k < FB_FLOATS; k += gridDim.z * blockDim.x) {
        float v = 0.f;
        if (k < FB_T1) {
            const int u = k / FC_LD, c = k % FC_LD;
            if (u < cc && c < 64) v = a.w0[i][(size_t)c * (td + cc) + td + u];
        } else if (k < FB_T2) {
            const int u = (k - FB_T1) / FC_LD, c = (k - FB_T1) % FC_LD;
            if (c < 64) v = a.w2[i][c * 64 + u];
        } else if (k < FB_WX) {
            const int u = (k - FB_T2) / 20, c = (k - FB_T2) % 20;
            if (c < wo) v = a.w4[i][c * 64 + u];
        } else {
            const int c = (k - FB_WX) >> 2, j = (k - FB_WX) & 3;
            if (j < td) v = a.w0[i][(size_t)c * (td + cc) + j];
        }
        img[k] = v;
    }
}
__global__ __launch_bounds__(256) void flowchain_pack_kernel(PfFlowChain a) {
    const int i = blockIdx.x;
    if (blockIdx.y == 0) fc_image_fwd(a.img + (size_t)i * FF_FLOATS, a, i);
    else fc_image_bwd(a.img + (size_t)a.nb * FF_FLOATS + (size_t)i * FB_FLOATS, a, i);
}

// image -> registers -> LDS: N4 float4 per image, FC_PRE per thread
constexpr int FC_PRE = 15;
static_assert(FF_FLOATS % 4 == 0 && FB_FLOATS % 4 == 0 && FF_FLOATS <= FC_PRE * 4 * 64 * FC_NW && FB_FLOATS <= FC_PRE * 4 * 64 * FC_NW, "image");
template <int FLOATS>
__device__ __forceinline__ void fc_fetch(f4 (&pre)[FC_PRE], const float* __restrict__ img) {
#pragma unroll
    for (int n = 0; n < FC_PRE; ++n) {
        const int k = threadIdx.x + 64 * FC_NW * n;
        pre[n] = k < FLOATS / 4 ? reinterpret_cast<const f4*>(img)[k] : pf_splat(0.f);
    }
}
template <int FLOATS>
__device__ __forceinline__ void fc_commit(const f4 (&pre)[FC_PRE], float* lds) {
#pragma unroll
    for (int n = 0; n < FC_PRE; ++n) {
        const int k = threadIdx.x + 64 * FC_NW * n;
        if (k < FLOATS / 4) reinterpret_cast<f4*>(lds)[k] = pre[n];
    }
}

// 3x3 parameters of every block in LDS: M (W, or W^-1 for the inverse direction), e^{+-logs}, bias; f also leaves ld[i]
struct FcPrm { float M[9], el[3], b[3]; };
__device__ __forceinline__ void fc_inv3(const float* W, float* Wi, float& det) {
    const float a = W[0], b = W[1], c = W[2], d = W[3], e = W[4], f = W[5], g = W[6], h = W[7], i = W[8];
    const float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    det = a * c00 + b * c01 + c * c02;
    const float r = 1.f / det;
    Wi[0] = c00 * r; Wi[1] = (c * h - b * i) * r; Wi[2] = (b * f - c * e) * r;
    Wi[3] = c01 * r; Wi[4] = (a * i - c * g) * r; Wi[5] = (c * d - a * f) * r;
    Wi[6] = c02 * r; Wi[7] = (b * g - a * h) * r; Wi[8] = (a * e - b * d) * r;
}
__device__ __forceinline__ void fc_params(float (*prm)[16], const PfFlowChain& a, bool write_ld) {
    if ((int)threadIdx.x < a.nb) {
        const int i = threadIdx.x;
        float W[9], Wi[9], det;
#pragma unroll
        for (int k = 0; k < 9; ++k) W[k] = a.W[i][k];
        fc_inv3(W, Wi, det);
#pragma unroll
        for (int k = 0; k < 9; ++k) prm[i][k] = a.inv ? Wi[k] : W[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            prm[i][9 + k] = expf(a.inv ? -a.logs[i][k] : a.logs[i][k]);
            prm[i][12 + k] = a.bias[i][k];
        }
        const float ldv = (a.logs[i][0] + a.logs[i][1] + a.logs[i][2] + logf(fabsf(det))) * a.n_ld;
        prm[i][15] = ldv;
        if (write_ld && a.ld) a.ld[i] = ldv;
    }
}

// conditioner forward on one 16-row tile: xv = leading coordinates (zero beyond td), crow = this row's conditioning features.
// h1, h2 stored; o0, o1 = the 3 - td outputs, broadcast to the four lanes of a row
__device__ __forceinline__ void fc_mlp_fwd(const float* lds, const float (&xv)[3], const float* __restrict__ crow, int cc, float* h1p,
                                           float* h2p, bool valid, int col, int q, float& o0, float& o1) {
    f4 cin[8], act[4], nxt[4];
    const int ld0 = cc + 4;
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
        cin[cb] = pf_splat(0.f);
        if (cb * 16 < cc) cin[cb] = *reinterpret_cast<const f4*>(crow + cb * 16 + 4 * q);
    }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        f4 acc = pf_splat(0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f4 wx = *reinterpret_cast<const f4*>(lds + FF_WX + (ob * 16 + 4 * q + r) * 4);
            acc[r] += wx.x * xv[0] + wx.y * xv[1] + wx.z * xv[2];
        }
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
            if (cb * 16 < cc)
                acc = fc_mfma4(*reinterpret_cast<const f4*>(lds + FF_W0 + (ob * 16 + col) * ld0 + cb * 16 + 4 * q), cin[cb], acc);
        acc = fc_lrelu(acc);
        if (valid) *reinterpret_cast<f4*>(h1p + ob * 16 + 4 * q) = acc;
        nxt[ob] = acc;
    }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        f4 acc = *reinterpret_cast<const f4*>(lds + FF_B1 + ob * 16 + 4 * q);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
            acc = fc_mfma4(*reinterpret_cast<const f4*>(lds + FF_W1 + (ob * 16 + col) * FC_LD + cb * 16 + 4 * q), nxt[cb], acc);
        acc = fc_lrelu(acc);
        if (valid) *reinterpret_cast<f4*>(h2p + ob * 16 + 4 * q) = acc;
        act[ob] = acc;
    }
    f4 acc = *reinterpret_cast<const f4*>(lds + FF_B2 + 4 * q);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
        acc = fc_mfma4(*reinterpret_cast<const f4*>(lds + FF_W2 + col * FC_LD + cb * 16 + 4 * q), act[cb], acc);
    o0 = __shfl(acc[0], col);                      // rows 0, 1 of the output tile live in the q = 0 lanes
    o1 = __shfl(acc[1], col);
}

// conditioner backward chain on one tile: do0, do1 = gradient of the 3 - td outputs; dz2, dz1 stored; dc summed over the R
// replicas of a conditioning row; sj = gradient of the leading coordinates
__device__ __forceinline__ void fc_mlp_bwd(const float* lds, float do0, float do1, int td, const float* __restrict__ h1p,
                                           const float* __restrict__ h2p, float* dz1p, float* dz2p, float* dcp, int cc, int R,
                                           bool valid, int col, int q, float (&sj)[3], float* dz1sp) {
    f4 gl = pf_splat(0.f);
    if (q == 0) { gl[0] = do0; gl[1] = td == 1 ? do1 : 0.f; }
    f4 g2[4], g1[4];
#pragma unroll
    for (int ub = 0; ub < 4; ++ub) {
        f4 acc = fc_mfma4(*reinterpret_cast<const f4*>(lds + FB_T2 + (ub * 16 + col) * 20 + 4 * q), gl, pf_splat(0.f));
        const f4 hv = *reinterpret_cast<const f4*>(h2p + ub * 16 + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] *= hv[r] > 0.f ? 1.f : FC_SLOPE;
        if (valid) *reinterpret_cast<f4*>(dz2p + ub * 16 + 4 * q) = acc;
        g2[ub] = acc;
    }
#pragma unroll
    for (int ub = 0; ub < 4; ++ub) {
        f4 acc = pf_splat(0.f);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
            acc = fc_mfma4(*reinterpret_cast<const f4*>(lds + FB_T1 + (ub * 16 + col) * FC_LD + cb * 16 + 4 * q), g2[cb], acc);
        const f4 hv = *reinterpret_cast<const f4*>(h1p + ub * 16 + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] *= hv[r] > 0.f ? 1.f : FC_SLOPE;
        if (valid) *reinterpret_cast<f4*>(dz1p + ub * 16 + 4 * q) = acc;
        g1[ub] = acc;
        if (dz1sp) {                                      // the same gradient summed over the R rows of this conditioning row (wave-uniform)
            f4 sm = acc;
#pragma unroll
            for (int m = 1; m < 16; m <<= 1)
                if (m < R) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sm[r] += __shfl_xor(sm[r], m);
                }
            if (valid && (col & (R - 1)) == 0) *reinterpret_cast<f4*>(dz1sp + ub * 16 + 4 * q) = sm;
        }
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f4 wx = *reinterpret_cast<const f4*>(lds + FB_WX + (cb * 16 + 4 * q + r) * 4);
            s0 = fmaf(g1[cb][r], wx.x, s0); s1 = fmaf(g1[cb][r], wx.y, s1); s2 = fmaf(g1[cb][r], wx.z, s2);
        }
    s0 += __shfl_xor(s0, 16); s0 += __shfl_xor(s0, 32);
    s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
    sj[0] = s0; sj[1] = s1; sj[2] = s2;
#pragma unroll
    for (int ub = 0; ub < 8; ++ub) {
        if (ub * 16 >= cc) break;
        f4 acc = pf_splat(0.f);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
            acc = fc_mfma4(*reinterpret_cast<const f4*>(lds + FB_T0 + (ub * 16 + col) * FC_LD + cb * 16 + 4 * q), g1[cb], acc);
#pragma unroll
        for (int m = 1; m < 16; m <<= 1)
            if (m < R) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += __shfl_xor(acc[r], m);
            }
        if (valid && (col & (R - 1)) == 0) *reinterpret_cast<f4*>(dcp + ub * 16 + 4 * q) = acc;
    }
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(64 * FC_NW) void flowchain_fwd_kernel(PfFlowChain a) {
    extern __shared__ float lds[];
    __shared__ float prm[PF_FLOWCHAIN_MAXB][16];
    __shared__ int last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
    const int ntiles = (a.rows + 15) / 16;
    const int tile = blockIdx.x * FC_NW + wave;
    const int p0 = tile * 16 + col;
    const bool valid = tile < ntiles && p0 < a.rows;
    const int pr = valid ? p0 : a.rows - 1;
    const int pt = pr / a.R;
    const size_t rows = (size_t)a.rows;
    fc_params(prm, a, blockIdx.x == 0);
    float p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = a.x[(size_t)pr * 3 + c];
    f4 pre[FC_PRE];
    fc_fetch<FF_FLOATS>(pre, a.img + (size_t)(a.inv ? a.nb - 1 : 0) * FF_FLOATS);
    for (int k = 0; k < a.nb; ++k) {
        const int i = a.inv ? a.nb - 1 - k : k;
        const int td = a.td[i];
        __syncthreads();
        fc_commit<FF_FLOATS>(pre, lds);
        __syncthreads();
        if (k + 1 < a.nb) fc_fetch<FF_FLOATS>(pre, a.img + (size_t)(a.inv ? i - 1 : i + 1) * FF_FLOATS);
        float M[9], el[3], b[3];
#pragma unroll
        for (int j = 0; j < 9; ++j) M[j] = prm[i][j];
#pragma unroll
        for (int j = 0; j < 3; ++j) { el[j] = prm[i][9 + j]; b[j] = prm[i][12 + j]; }
        const size_t slab = (size_t)i * rows + pr;
        float s[3], t[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { s[c] = a.s[i][(size_t)pt * 3 + c]; t[c] = a.t[i][(size_t)pt * 3 + c]; }
        if (valid && q == 0)
#pragma unroll
            for (int c = 0; c < 3; ++c) a.pin[slab * 3 + c] = p[c];
        float m[3];                                     // f: y = W (p e^logs + bias);  g: v = reverse(p e^s + t)
        if (!a.inv) {
            float tt[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) tt[c] = fmaf(p[c], el[c], b[c]);
#pragma unroll
            for (int r = 0; r < 3; ++r) m[r] = M[r * 3] * tt[0] + M[r * 3 + 1] * tt[1] + M[r * 3 + 2] * tt[2];
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) m[2 - c] = fmaf(p[c], expf(s[c]), t[c]);
        }
        if (valid && q == 0)
#pragma unroll
            for (int c = 0; c < 3; ++c) a.mid[slab * 3 + c] = m[c];
        float xv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) xv[j] = j < td ? m[j] : 0.f;
        float o0, o1;
        fc_mlp_fwd(lds, xv, a.c[i] + (size_t)pt * a.cc[i], a.cc[i], a.h1 + slab * 64, a.h2 + slab * 64, valid, col, q, o0, o1);
        const float ov[3] = {0.f, td == 1 ? o0 : 0.f, td == 1 ? o1 : o0};      // shift of coordinate c (zero for c < td)
        if (!a.inv) {
            float h[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) h[c] = m[c] - ov[c];
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = (h[2 - c] - t[c]) * expf(-s[c]);
            float sv = (valid && q == 0) ? (s[0] + s[1]) + s[2] : 0.f;
#pragma unroll
            for (int w = 1; w < 64; w <<= 1) sv += __shfl_xor(sv, w);
            if (lane == 0 && tile < ntiles)
                __hip_atomic_store(a.part + (size_t)i * ntiles + tile, sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (valid && q == 0) { a.o[slab * 2] = o0; a.o[slab * 2 + 1] = o1; }
            float mi[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) mi[c] = m[c] + ov[c];
#pragma unroll
            for (int r = 0; r < 3; ++r) p[r] = (M[r * 3] * mi[0] + M[r * 3 + 1] * mi[1] + M[r * 3 + 2] * mi[2] - b[r]) * el[r];
        }
    }
    if (valid && q == 0)
#pragma unroll
        for (int c = 0; c < 3; ++c) a.out[(size_t)p0 * 3 + c] = p[c];
    if (a.inv) return;
    if (a.logp) {                                       // standard-normal log-density of z (probs.py:73-75), per wave tile
        float gv = (valid && q == 0) ? -0.5f * ((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2] + 3.f * 1.8378770664093453f) : 0.f;
#pragma unroll
        for (int w = 1; w < 64; w <<= 1) gv += __shfl_xor(gv, w);
        if (lane == 0 && tile < ntiles)
            __hip_atomic_store(a.part + (size_t)a.nb * ntiles + tile, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // sum(s) of every block: the workgroup that arrives last adds the per-tile sums in a fixed order
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(a.counter, 1u) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!last) return;
    {
        __shared__ float tot[PF_FLOWCHAIN_MAXB + 1];
        const int i = threadIdx.x >> 5, sub = threadIdx.x & 31;
        const int nrow = a.nb + (a.logp ? 1 : 0);          // row nb: the log-density partials (nb <= 7 when logp is asked for)
        float sv = 0.f;
        if (i < nrow)
            for (int w = sub; w < ntiles; w += 32)
                sv += __hip_atomic_load(a.part + (size_t)i * ntiles + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int w = 1; w < 32; w <<= 1) sv += __shfl_xor(sv, w);
        if (i < a.nb && sub == 0) a.ssum[i] = sv;
        if (i < nrow && sub == 0) tot[i] = sv;
        __syncthreads();
        if (threadIdx.x == 0) {
            *a.counter = 0u;
            if (a.logp) {                                 // -mean_b(log N(z_b) + sum_i (ld_i - sum(s_i)[b]))  (interpflow.py:327-337)
                float lds_ = 0.f, ss = 0.f;
                for (int k2 = 0; k2 < a.nb; ++k2) { lds_ += prm[k2][15]; ss += tot[k2]; }
                a.logp[0] = -(tot[a.nb] / (float)a.Bsz + lds_ - ss / (float)a.Bsz);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward, chain
__global__ __launch_bounds__(64 * FC_NW) void flowchain_bwd_kernel(PfFlowChain a) {
    extern __shared__ float lds[];
    __shared__ float prm[PF_FLOWCHAIN_MAXB][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
    const int ntiles = (a.rows + 15) / 16;
    const int tile = blockIdx.x * FC_NW + wave;
    const int p0 = tile * 16 + col;
    const bool valid = tile < ntiles && p0 < a.rows;
    const int pr = valid ? p0 : a.rows - 1;
    const int pt = pr / a.R;
    const size_t rows = (size_t)a.rows;
    fc_params(prm, a, false);
    float g[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = (valid && a.dout) ? a.dout[(size_t)pr * 3 + c] : 0.f;
    const float glp = (!a.inv && a.dlogp) ? a.dlogp[0] / (float)a.Bsz : 0.f;       // d logp / d z = z / B,  d logp / d ssum_i = 1 / B
    if (!a.inv && a.dlogp && valid)
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] = fmaf(glp, a.out[(size_t)pr * 3 + c], g[c]);
    const float* bimg = a.img + (size_t)a.nb * FF_FLOATS;
    f4 pre[FC_PRE];
    fc_fetch<FB_FLOATS>(pre, bimg + (size_t)(a.inv ? 0 : a.nb - 1) * FB_FLOATS);
    for (int k = 0; k < a.nb; ++k) {
        const int i = a.inv ? k : a.nb - 1 - k;         // the forward walk backwards
        const int td = a.td[i], wo = 3 - td;
        __syncthreads();
        fc_commit<FB_FLOATS>(pre, lds);
        __syncthreads();
        if (k + 1 < a.nb) fc_fetch<FB_FLOATS>(pre, bimg + (size_t)(a.inv ? i + 1 : i - 1) * FB_FLOATS);
        float M[9], el[3], b[3];
#pragma unroll
        for (int j = 0; j < 9; ++j) M[j] = prm[i][j];
#pragma unroll
        for (int j = 0; j < 3; ++j) { el[j] = prm[i][9 + j]; b[j] = prm[i][12 + j]; }
        const size_t slab = (size_t)i * rows + pr;
        float s[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) s[c] = a.s[i][(size_t)pt * 3 + c];
        float acc[15];
        float dmid[3], do0, do1;
        if (!a.inv) {
            // injector + reverse + coupling:  out = (reverse([y_head, y_tail - o]) - t) e^-s
            const float* outp = i == a.nb - 1 ? a.out + (size_t)pr * 3 : a.pin + ((size_t)(i + 1) * rows + pr) * 3;
            const float gs = a.dlogp ? glp : (a.dssum ? a.dssum[i] : 0.f);
            float dv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dv[c] = g[c] * expf(-s[c]);
                if (valid && q == 0) {
                    a.ds[i][(size_t)p0 * 3 + c] = gs - g[c] * outp[c];
                    a.dt[i][(size_t)p0 * 3 + c] = -dv[c];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) dmid[c] = dv[2 - c];
            do0 = td == 1 ? -dmid[1] : -dmid[2];
            do1 = td == 1 ? -dmid[2] : 0.f;
        } else {
            // u' = (Winv [v_head, v_tail + o] - bias) e^-logs
            float v[3], mi[3], mm[3], dm[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = a.mid[slab * 3 + c];
            const float o0 = a.o[slab * 2], o1 = a.o[slab * 2 + 1];
            const float ov[3] = {0.f, td == 1 ? o0 : 0.f, td == 1 ? o1 : o0};
#pragma unroll
            for (int c = 0; c < 3; ++c) mi[c] = v[c] + ov[c];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                mm[r] = M[r * 3] * mi[0] + M[r * 3 + 1] * mi[1] + M[r * 3 + 2] * mi[2] - b[r];
                dm[r] = g[r] * el[r];
                acc[r] = -dm[r] * mm[r];
                acc[3 + r] = -dm[r];
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[6 + r * 3 + j] = dm[r] * mi[j];
#pragma unroll
            for (int j = 0; j < 3; ++j) dmid[j] = M[j] * dm[0] + M[3 + j] * dm[1] + M[6 + j] * dm[2];
            do0 = td == 1 ? dmid[1] : dmid[2];
            do1 = td == 1 ? dmid[2] : 0.f;
        }
        if (valid && q == 0) {
            float* dp = a.dob + (size_t)i * rows * 2 + (size_t)p0 * wo;
            dp[0] = do0;
            if (wo == 2) dp[1] = do1;
        }
        float sj[3];
        fc_mlp_bwd(lds, do0, do1, td, a.h1 + slab * 64, a.h2 + slab * 64, a.dz1 + slab * 64, a.dz2 + slab * 64,
                   a.dc[i] + (size_t)pt * a.cc[i], a.cc[i], a.R, valid, col, q, sj,
                   (a.dz1s && a.R > 1) ? a.dz1s + ((size_t)i * (rows / a.R) + pt) * 64 : nullptr);
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < td) dmid[j] += sj[j];
        if (!a.inv) {
            // y = W (x e^logs + bias)
            float x[3], tt[3], dtt[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { x[c] = a.pin[slab * 3 + c]; tt[c] = fmaf(x[c], el[c], b[c]); }
#pragma unroll
            for (int j = 0; j < 3; ++j) dtt[j] = M[j] * dmid[0] + M[3 + j] * dmid[1] + M[6 + j] * dmid[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                g[c] = dtt[c] * el[c];
                acc[c] = dtt[c] * x[c] * el[c];
                acc[3 + c] = dtt[c];
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[6 + r * 3 + j] = dmid[r] * tt[j];
        } else {
            // v = reverse(u e^s + t), s, t of the original point: ds, dt summed over its R rows (adjacent columns of the tile)
            float u[3], as[3], at[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                u[c] = a.pin[slab * 3 + c];
                const float e = expf(s[c]), gg = dmid[2 - c];
                g[c] = gg * e;
                as[c] = gg * u[c] * e;
                at[c] = gg;
            }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1)
                if (m < a.R) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) { as[c] += __shfl_xor(as[c], m); at[c] += __shfl_xor(at[c], m); }
                }
            if (valid && q == 0 && (col & (a.R - 1)) == 0)
#pragma unroll
                for (int c = 0; c < 3; ++c) { a.ds[i][(size_t)pt * 3 + c] = as[c]; a.dt[i][(size_t)pt * 3 + c] = at[c]; }
        }
        // the 15 parameter-gradient sums of this tile (rows are replicated over the four q groups: q = 0 counts)
#pragma unroll
        for (int n = 0; n < 15; ++n) {
            float sv = (valid && q == 0) ? acc[n] : 0.f;
            sv += __shfl_xor(sv, 1); sv += __shfl_xor(sv, 2); sv += __shfl_xor(sv, 4); sv += __shfl_xor(sv, 8);
            acc[n] = sv;
        }
        if (lane == 0 && tile < ntiles) {
            float* pp = a.part + ((size_t)i * ntiles + tile) * 16;
#pragma unroll
            for (int n = 0; n < 15; ++n) pp[n] = acc[n];
        }
    }
    if (a.dx && valid && q == 0)
#pragma unroll
        for (int c = 0; c < 3; ++c) a.dx[(size_t)p0 * 3 + c] = g[c];
}

// per block: sum of the tile partials -> dlogs, dbias and the matrix gradient; f adds the log-det terms (ld = (sum logs +
// log|det W|) n), g turns the gradient of W^-1 into the gradient of W (dW = -W^-T dWinv W^-T)
__global__ __launch_bounds__(256) void flowchain_param_kernel(PfFlowChain a) {
    __shared__ float fs[16][16];
    __shared__ float sm[16];
    const int i = blockIdx.x;
    const int ntiles = (a.rows + 15) / 16;
    const int vi = threadIdx.x & 15, sub = threadIdx.x >> 4;
    const float* pp = a.part + (size_t)i * ntiles * 16;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = sub;
    for (; w + 48 < ntiles; w += 64) {
        s0 += pp[(size_t)w * 16 + vi]; s1 += pp[(size_t)(w + 16) * 16 + vi];
        s2 += pp[(size_t)(w + 32) * 16 + vi]; s3 += pp[(size_t)(w + 48) * 16 + vi];
    }
    for (; w < ntiles; w += 16) s0 += pp[(size_t)w * 16 + vi];
    fs[sub][vi] = vi < 15 ? (s0 + s1) + (s2 + s3) : 0.f;
    __syncthreads();
    if (threadIdx.x < 16) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += fs[k][threadIdx.x];
        sm[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float W[9], Wi[9], det;
    for (int k = 0; k < 9; ++k) W[k] = a.W[i][k];
    fc_inv3(W, Wi, det);
    if (!a.inv) {
        const float gl = a.dlogp ? -a.dlogp[0] * a.n_ld : (a.dld ? a.dld[i] * a.n_ld : 0.f);       // d logp / d ld_i = -1
        for (int c = 0; c < 3; ++c) { a.dlogs[i][c] = sm[c] + gl; a.dbias[i][c] = sm[3 + c]; }
        for (int r = 0; r < 3; ++r)
            for (int j = 0; j < 3; ++j) a.dW[i][r * 3 + j] = sm[6 + r * 3 + j] + gl * Wi[j * 3 + r];
    } else {
        for (int c = 0; c < 3; ++c) { a.dlogs[i][c] = sm[c]; a.dbias[i][c] = sm[3 + c]; }
        float t[9];                                      // t = Winv^T dWinv
        for (int r = 0; r < 3; ++r)
            for (int j = 0; j < 3; ++j) {
                float s = 0.f;
                for (int k = 0; k < 3; ++k) s += Wi[k * 3 + r] * sm[6 + k * 3 + j];
                t[r * 3 + j] = s;
            }
        for (int r = 0; r < 3; ++r)
            for (int j = 0; j < 3; ++j) {
                float s = 0.f;
                for (int k = 0; k < 3; ++k) s += t[r * 3 + k] * Wi[j * 3 + k];
                a.dW[i][r * 3 + j] = -s;
            }
    }
}

int fc_check(const PfFlowChain* a, bool bwd) {
    if (!a) return PF_ERR_NULL;
    if (a->nb < 1 || a->nb > PF_FLOWCHAIN_MAXB || a->rows <= 0) return PF_ERR_SHAPE;
    if (a->R != 1 && a->R != 2 && a->R != 4 && a->R != 8 && a->R != 16) return PF_ERR_UNSUPPORTED;
    if (a->rows % a->R != 0) return PF_ERR_SHAPE;
    if (!a->x || !a->pin || !a->mid || !a->h1 || !a->h2 || !a->out || !a->img) return PF_ERR_NULL;
    if (a->inv ? !a->o : (!bwd && (!a->ssum || !a->part || !a->counter))) return PF_ERR_NULL;
    if (!a->inv && (a->logp || a->dlogp) && (a->Bsz < 1 || a->nb >= PF_FLOWCHAIN_MAXB)) return PF_ERR_SHAPE;
    for (int i = 0; i < a->nb; ++i) {
        if (a->td[i] != 1 && a->td[i] != 2) return PF_ERR_UNSUPPORTED;
        if (a->cc[i] != 32 && a->cc[i] != 64 && a->cc[i] != 128) return PF_ERR_UNSUPPORTED;
        if (!a->c[i] || !a->s[i] || !a->t[i] || !a->logs[i] || !a->bias[i] || !a->W[i] || !a->w0[i] || !a->w2[i] || !a->b2[i] ||
            !a->w4[i] || !a->b4[i])
            return PF_ERR_NULL;
        if (bwd && (!a->dc[i] || !a->ds[i] || !a->dt[i] || !a->dlogs[i] || !a->dbias[i] || !a->dW[i] || !a->dw0[i] || !a->dw2[i] ||
                    !a->db2[i] || !a->dw4[i] || !a->db4[i]))
            return PF_ERR_NULL;
    }
    if (bwd && ((!a->dout && !a->dlogp) || !a->dz1 || !a->dz2 || !a->dob || !a->part || !a->ws || !a->dev_descs)) return PF_ERR_NULL;
    return PF_OK;
}

#ifndef PF_FC_CHUNK_BIG
#define PF_FC_CHUNK_BIG 512
#endif
#ifndef PF_FC_CHUNK
#define PF_FC_CHUNK 256
#endif
void fc_desc(const PfFlowChain* a, int i, PfMlpTrain* d) {
    *d = PfMlpTrain{};
    const size_t slab = (size_t)i * a->rows;
    d->rows = a->rows; d->nl = 3; d->td = a->td[i]; d->ldy = 3; d->cc = a->cc[i]; d->cdiv = a->R;
    d->width[0] = 64; d->width[1] = 64; d->width[2] = 3 - a->td[i];
    d->slope[0] = d->slope[1] = FC_SLOPE;
    d->chunk = a->rows >= 32768 ? PF_FC_CHUNK_BIG : PF_FC_CHUNK;            // nb networks x 3 layers in one launch: long split-K chunks (tools/time_mlpdw.py: 129 vs 141 us, 55 vs 57 us)
    d->y = a->mid ? a->mid + slab * 3 : nullptr;
    d->c = a->c[i];
    d->W[0] = a->w0[i]; d->W[1] = a->w2[i]; d->W[2] = a->w4[i];
    d->h[0] = a->h1 ? a->h1 + slab * 64 : nullptr; d->h[1] = a->h2 ? a->h2 + slab * 64 : nullptr;
    d->dout = a->dob ? a->dob + slab * 2 : nullptr;
    d->dz[0] = a->dz1 ? a->dz1 + slab * 64 : nullptr; d->dz[1] = a->dz2 ? a->dz2 + slab * 64 : nullptr;
    if (a->dz1s && a->R > 1 && a->cc[i] % 16 == 0) {        // layer 0's conditioning columns over rows / R summed rows (PF_MLP_DW_DZSUM)
        d->dc = a->dz1s + (size_t)i * (a->rows / a->R) * 64;
        d->flags = PF_MLP_DW_DZSUM;
    }
    d->dW[0] = a->dw0[i]; d->dW[1] = a->dw2[i]; d->dW[2] = a->dw4[i];
    d->db[1] = a->db2[i]; d->db[2] = a->db4[i];
}

}  // namespace

// floats of split-K scratch (`ws`) the backward needs
extern "C" long long pf_flowchain_ws_floats(const PfFlowChain* a) {
    if (!a || a->nb < 1 || a->nb > PF_FLOWCHAIN_MAXB) return -1;
    long long tot = 0;
    for (int i = 0; i < a->nb; ++i) {
        PfMlpTrain d;
        fc_desc(a, i, &d);
        d.y = a->x;                                     // sizing only: any non-null row pointer
        const long long n = pf_mlp_train_ws_floats(&d);
        if (n < 0) return -1;
        tot += n;
    }
    return tot;
}
// floats of `part`: forward (f) nb * tiles, backward nb * tiles * 16
extern "C" long long pf_flowchain_part_floats(const PfFlowChain* a) {
    if (!a || a->rows <= 0) return -1;
    return (long long)a->nb * ((a->rows + 15) / 16) * 16;
}

// floats of `img`: the packed conditioner weights of every block (forward and backward layout), written by pf_flowchain_fwd
// and read again by pf_flowchain_bwd
extern "C" long long pf_flowchain_img_floats(const PfFlowChain* a) {
    if (!a || a->nb < 1 || a->nb > PF_FLOWCHAIN_MAXB) return -1;
    return (long long)a->nb * (FF_FLOATS + FB_FLOATS);
}

extern "C" int pf_flowchain_fwd(const PfFlowChain* a, void* stream) {
    int st = fc_check(a, false);
    if (st) return st;
    const int ntiles = (a->rows + 15) / 16;
    const size_t lds = FF_FLOATS * sizeof(float);
    if (!a->img_ready) hipLaunchKernelGGL(flowchain_pack_kernel, dim3(a->nb, 2, 14), dim3(256), 0, (hipStream_t)stream, *a);
    hipLaunchKernelGGL(flowchain_fwd_kernel, dim3((ntiles + FC_NW - 1) / FC_NW), dim3(64 * FC_NW), lds, (hipStream_t)stream, *a);
    return pf_last_launch_status();
}

extern "C" int pf_flowchain_bwd(const PfFlowChain* a, void* stream) {
    int st = fc_check(a, true);
    if (st) return st;
    if (a->ws_floats < pf_flowchain_ws_floats(a)) return PF_ERR_WORKSPACE;
    const int ntiles = (a->rows + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = FB_FLOATS * sizeof(float);
    hipLaunchKernelGGL(flowchain_bwd_kernel, dim3((ntiles + FC_NW - 1) / FC_NW), dim3(64 * FC_NW), lds, s, *a);
    hipLaunchKernelGGL(flowchain_param_kernel, dim3(a->nb), dim3(256), 0, s, *a);
    PfMlpTrain descs[PF_FLOWCHAIN_MAXB];
    long long off = 0;
    for (int i = 0; i < a->nb; ++i) {
        fc_desc(a, i, &descs[i]);
        const long long n = pf_mlp_train_ws_floats(&descs[i]);
        if (n < 0) return PF_ERR_UNSUPPORTED;
        descs[i].ws = a->ws + off;
        descs[i].ws_floats = n;
        off += n;
    }
    st = pf_mlp_train_dw_batch(descs, a->nb, a->dev_descs, stream);
    if (st) return st;
    return pf_last_launch_status();
}
