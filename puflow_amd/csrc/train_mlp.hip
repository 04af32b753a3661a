// Point-wise MLPs of the TRAINING step (no BatchNorm) as one launch forward and three backward.
//
// Reference: modules/discrete/interpflow.py:22-43 (LinearA1D: Linear(no bias) - LeakyReLU - Linear - LeakyReLU - Linear, the
// conditioner of every coupling / injector layer, interpflow.py:46-82) and interpflow.py:251-258 (FeatMergeUnit:
// Linear - ReLU - Linear).  The un-fused path ran each of the ~30 evaluations per step as 6 launches forward and ~20
// backward on [8192..32768, 64] activations - all launch latency.  Here:
//
//   forward   one kernel: the 2 or 3 layers chained in registers ("channel-major": output channels on the MFMA rows, 16 points
//             on the columns, so a layer's accumulator tile IS the next layer's B operand - csrc/pf_mfma.h), hidden activations
//             stored once for the backward
//   backward  chain   dz_l = (W_l^T dz_{l+1}) * act'(h_l) layer by layer in registers, stored for the weight gradients;
//                     input gradients: dy[:, :td] and dc (summed over the `cdiv` replicas of a conditioning row)
//             dw      every layer's dW = dz^T a and db = sum dz in ONE split-K launch (blockIdx.y = layer; a block of rows
//                     staged through LDS, K dimension = rows), partial sums per row chunk
//             reduce  partial sums -> dW, db
//
// The first layer's input is cat[y[:, :td], c[row / cdiv]]: td <= 3 leading columns of a [rows, ldy] tensor (the coupling's
// untouched coordinates) and cc conditioning channels shared by cdiv consecutive rows (the x`upratio` replicas in the
// inverse pass) - neither the concatenation nor the replicated conditioning tensor exists in memory.
// All products are v_mfma_f32_16x16x4_f32 (fp32 fma chains).
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"
#include "pf_mfma.h"

namespace {

constexpr int MLP_GRID = 512;
#ifndef PF_MLP_DW_WAVES
#define PF_MLP_DW_WAVES 4
#endif
#ifndef PF_MLP_RG
#define PF_MLP_RG 16
#endif
constexpr int MLP_EB = 32, MLP_DW_WAVES = PF_MLP_DW_WAVES, MLP_SLOTS = (36 + MLP_DW_WAVES - 1) / MLP_DW_WAVES;   // 36 = 4 x 9 tiles of the widest layer
// rows per split-K chunk (multiple of MLP_EB): the descriptor's choice (batched launches have networks x layers of parallelism
// already and want long chunks: fewer partial sums to write and add), else sized so that one network fills the chip
__host__ __device__ inline int mlp_chunk(const PfMlpTrain& p) { return p.chunk > 0 ? p.chunk : (p.rows > 16384 ? 128 : 64); }

__device__ __forceinline__ f4 mfma4(f4 a, f4 b, f4 c) {
    c = pf_mfma(a.x, b.x, c); c = pf_mfma(a.y, b.y, c); c = pf_mfma(a.z, b.z, c); c = pf_mfma(a.w, b.w, c);
    return c;
}
__device__ __forceinline__ f4 lrelu4(f4 z, float s) {
    f4 r;
    r.x = fmaxf(z.x, z.x * s); r.y = fmaxf(z.y, z.y * s); r.z = fmaxf(z.z, z.z * s); r.w = fmaxf(z.w, z.w * s);
    return r;
}
__host__ __device__ inline int up16(int v) { return (v + 15) & ~15; }

struct MlpShape {
    int wi[3], wo[3], wi16[3], wo16[3];   // MFMA part of layer l: wi[0] = cc
    int in[3];                             // row length of W[l]: in[0] = td + cc
};
__host__ __device__ inline MlpShape mlp_shape(const PfMlpTrain& p) {
    MlpShape s{};
#pragma unroll
    for (int l = 0; l < 3; ++l) {                       // unrolled with constant indices: the arrays stay in registers
        if (l < p.nl) {
            s.wi[l] = l == 0 ? p.cc : p.width[l > 0 ? l - 1 : 0];
            s.wo[l] = p.width[l];
            s.wi16[l] = up16(s.wi[l]); s.wo16[l] = up16(s.wo[l]);
            s.in[l] = l == 0 ? p.td + p.cc : p.width[l > 0 ? l - 1 : 0];
        }
    }
    return s;
}
template <typename T>
__host__ __device__ inline T sel3(const T (&a)[3], int l) { return l == 0 ? a[0] : (l == 1 ? a[1] : a[2]); }
template <typename T>
__host__ __device__ inline T sel2(const T (&a)[2], int l) { return l == 0 ? a[0] : a[1]; }

// ------------------------------------------------------------------------------------------------ forward
template <int NL>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(PfMlpTrain p0, const PfMlpTrain* descs) {
    extern __shared__ float lds[];
    const PfMlpTrain p = descs ? descs[blockIdx.z] : p0;             // batched launch: one network per blockIdx.z
    const int ntiles = (p.rows + 15) / 16;
    const MlpShape sh = mlp_shape(p);
    float* Wl[NL];
    float* bl[NL];
    float* ptr = lds;
#pragma unroll
    for (int l = 0; l < NL; ++l) { Wl[l] = ptr; ptr += sh.wo16[l] * (sh.wi16[l] + 4); bl[l] = ptr; ptr += sh.wo16[l]; }
    float* Wx = ptr;                                     // [wo16[0]][4]: the td leading columns of W[0]
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int ld = sh.wi16[l] + 4, off = l == 0 ? p.td : 0;
        const int sft = 31 - __clz(sh.wi16[l]);                          // wi16 is a power of two (16 .. 128)
        // 8 loads in flight per thread, then 8 LDS stores: a load -> store loop pays a full memory latency per element
        for (int i0 = threadIdx.x; i0 < sh.wo16[l] * sh.wi16[l]; i0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + 256 * k, c = i >> sft, u = i & (sh.wi16[l] - 1);
                v[k] = (i < sh.wo16[l] * sh.wi16[l] && c < sh.wo[l] && u < sh.wi[l]) ? p.W[l][(size_t)c * sh.in[l] + off + u] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + 256 * k, c = i >> sft, u = i & (sh.wi16[l] - 1);
                if (i < sh.wo16[l] * sh.wi16[l]) Wl[l][c * ld + u] = v[k];
            }
        }
        for (int i = threadIdx.x; i < sh.wo16[l]; i += 256) bl[l][i] = (p.b[l] && i < sh.wo[l]) ? p.b[l][i] : 0.f;
    }
    for (int i = threadIdx.x; i < sh.wo16[0] * 4; i += 256) {
        const int c = i >> 2, j = i & 3;
        Wx[i] = (c < sh.wo[0] && j < p.td) ? p.W[0][(size_t)c * sh.in[0] + j] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int p0 = tile * 16 + col;
        const bool valid = p0 < p.rows;
        const int pr = valid ? p0 : p.rows - 1;
        f4 act[8];
        const float* crow = p.c + (size_t)(pr / p.cdiv) * p.cc;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            act[cb] = pf_splat(0.f);
            if (cb * 16 < sh.wi16[0]) act[cb] = *reinterpret_cast<const f4*>(crow + cb * 16 + 4 * q);
        }
        float xv[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < p.td) xv[j] = p.y[(size_t)pr * p.ldy + j];
        pf_static_for<0, NL>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            const int ld = sh.wi16[l] + 4;
            f4 nxt[8];
#pragma unroll
            for (int ob = 0; ob < 8; ++ob) {
                nxt[ob] = pf_splat(0.f);
                if (ob * 16 < sh.wo16[l]) {
                    f4 acc = *reinterpret_cast<const f4*>(bl[l] + ob * 16 + 4 * q);
                    if (l == 0 && p.td > 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const f4 wx = *reinterpret_cast<const f4*>(Wx + (ob * 16 + 4 * q + r) * 4);
                            acc[r] += wx.x * xv[0] + wx.y * xv[1] + wx.z * xv[2];
                        }
                    }
#pragma unroll
                    for (int cb = 0; cb < 8; ++cb)
                        if (cb * 16 < sh.wi16[l])
                            acc = mfma4(*reinterpret_cast<const f4*>(Wl[l] + (ob * 16 + col) * ld + cb * 16 + 4 * q), act[cb], acc);
                    if (l < NL - 1) {
                        constexpr int lh = l < 2 ? l : 1;
                        acc = lrelu4(acc, p.slope[lh]);
                        if (valid) *reinterpret_cast<f4*>(p.h[lh] + (size_t)p0 * sh.wo[l] + ob * 16 + 4 * q) = acc;
                        nxt[ob] = acc;
                    } else if (valid) {
                        if ((sh.wo[l] & 15) == 0) *reinterpret_cast<f4*>(p.out + (size_t)p0 * sh.wo[l] + ob * 16 + 4 * q) = acc;
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int ch = ob * 16 + 4 * q + r;
                                if (ch < sh.wo[l]) p.out[(size_t)p0 * sh.wo[l] + ch] = acc[r];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int ob = 0; ob < 8; ++ob) act[ob] = nxt[ob];
        });
    }
}

// ------------------------------------------------------------------------------------------------ backward, chain
template <int NL>
__global__ __launch_bounds__(256) void mlp_bwd_kernel(PfMlpTrain p0, const PfMlpTrain* descs) {
    extern __shared__ float lds[];
    const PfMlpTrain p = descs ? descs[blockIdx.z] : p0;
    const int ntiles = (p.rows + 15) / 16;
    const MlpShape sh = mlp_shape(p);
    float* Wt[NL];                                       // Wt[l][u][c] = W[l][c][off + u]
    float* ptr = lds;
#pragma unroll
    for (int l = 0; l < NL; ++l) { Wt[l] = ptr; ptr += sh.wi16[l] * (sh.wo16[l] + 4); }
    float* Wx = ptr;                                     // [wo16[0]][4]
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int ld = sh.wo16[l] + 4, off = l == 0 ? p.td : 0;
        const int sft = 31 - __clz(sh.wi16[l]);                          // wi16 is a power of two (16 .. 128)
        for (int i0 = threadIdx.x; i0 < sh.wo16[l] * sh.wi16[l]; i0 += 256 * 8) {   // consecutive threads: consecutive u (contiguous in W)
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + 256 * k, c = i >> sft, u = i & (sh.wi16[l] - 1);
                v[k] = (i < sh.wo16[l] * sh.wi16[l] && c < sh.wo[l] && u < sh.wi[l]) ? p.W[l][(size_t)c * sh.in[l] + off + u] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + 256 * k, c = i >> sft, u = i & (sh.wi16[l] - 1);
                if (i < sh.wo16[l] * sh.wi16[l]) Wt[l][u * ld + c] = v[k];
            }
        }
    }
    for (int i = threadIdx.x; i < sh.wo16[0] * 4; i += 256) {
        const int c = i >> 2, j = i & 3;
        Wx[i] = (c < sh.wo[0] && j < p.td) ? p.W[0][(size_t)c * sh.in[0] + j] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int p0 = tile * 16 + col;
        const bool valid = p0 < p.rows;
        const int pr = valid ? p0 : p.rows - 1;
        f4 g[8];
        {
            const int w = sh.wo[NL - 1];
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                g[cb] = pf_splat(0.f);
                if (cb * 16 < sh.wo16[NL - 1] && valid) {
                    const int ch = cb * 16 + 4 * q;
                    if ((w & 3) == 0) {
                        if (ch < w) g[cb] = *reinterpret_cast<const f4*>(p.dout + (size_t)p0 * w + ch);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (ch + r < w) g[cb][r] = p.dout[(size_t)p0 * w + ch + r];
                    }
                }
            }
        }
        pf_static_for<0, NL>([&](auto lc) {
            constexpr int l = NL - 1 - decltype(lc)::value;
            const int ld = sh.wo16[l] + 4;
            if (l == 0 && p.td > 0) {                     // dy[:, j] = sum_c dz1[c] W0[c][j]
                float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int cb = 0; cb < 8; ++cb)
                    if (cb * 16 < sh.wo16[0])
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const f4 wx = *reinterpret_cast<const f4*>(Wx + (cb * 16 + 4 * q + r) * 4);
                            s0 = fmaf(g[cb][r], wx.x, s0); s1 = fmaf(g[cb][r], wx.y, s1); s2 = fmaf(g[cb][r], wx.z, s2);
                        }
                s0 += __shfl_xor(s0, 16); s0 += __shfl_xor(s0, 32);
                s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
                if (q == 0 && valid && p.dy) {
                    const float sv[3] = {s0, s1, s2};
                    for (int j = 0; j < p.ldy; ++j) p.dy[(size_t)p0 * p.ldy + j] = (j < p.td && j < 3) ? sv[j < 3 ? j : 0] : 0.f;
                }
            }
            f4 nxt[8];
#pragma unroll
            for (int ub = 0; ub < 8; ++ub) {
                nxt[ub] = pf_splat(0.f);
                if (ub * 16 < sh.wi16[l]) {
                    f4 acc = pf_splat(0.f);
#pragma unroll
                    for (int cb = 0; cb < 8; ++cb)
                        if (cb * 16 < sh.wo16[l])
                            acc = mfma4(*reinterpret_cast<const f4*>(Wt[l] + (ub * 16 + col) * ld + cb * 16 + 4 * q), g[cb], acc);
                    if (l > 0) {
                        constexpr int lm = l > 0 ? l - 1 : 0;
                        const f4 hv = *reinterpret_cast<const f4*>(p.h[lm] + (size_t)pr * sh.wi[l] + ub * 16 + 4 * q);
                        const float sl = p.slope[lm];
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] *= hv[r] > 0.f ? 1.f : sl;
                        if (valid) *reinterpret_cast<f4*>(p.dz[lm] + (size_t)p0 * sh.wi[l] + ub * 16 + 4 * q) = acc;
                        nxt[ub] = acc;
                    } else if (p.dc) {                    // sum over the cdiv replicas of a conditioning row: adjacent columns
#pragma unroll
                        for (int m = 1; m < 16; m <<= 1)
                            if (m < p.cdiv) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[r] += __shfl_xor(acc[r], m);
                            }
                        if (valid && (col % p.cdiv) == 0)
                            *reinterpret_cast<f4*>(p.dc + (size_t)(p0 / p.cdiv) * p.cc + ub * 16 + 4 * q) = acc;
                    }
                }
            }
#pragma unroll
            for (int ub = 0; ub < 8; ++ub) g[ub] = nxt[ub];
        });
    }
}

// ------------------------------------------------------------------------------------------------ backward, weights
// layer l = blockIdx.y: part[chunk][off_l + c * wb16 + u] = sum over the chunk's rows of dz_{l+1}[row, c] * a_l[row, u];
// a_0 = [c[row / cdiv] (cc) | y[row, :td]], a_l = h[l-1]; bias partial sums behind the layer's weight block.
struct MlpDwLayout {
    int off[3], boff[3], wb16[3], total;
};
__host__ __device__ inline MlpDwLayout mlp_dw_layout(const PfMlpTrain& p, const MlpShape& sh) {
    MlpDwLayout L{};
    int o = 0;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        if (l < p.nl) {
            L.wb16[l] = l == 0 ? up16(p.cc + p.td) : sh.wi16[l];
            L.off[l] = o; o += sh.wo16[l] * L.wb16[l];
            L.boff[l] = o; o += sh.wo16[l];
        }
    }
    L.total = o;
    return L;
}

// MLP_DW_WAVES waves share one staged block (as csrc/train_fused.hip ec_dw_kernel: more waves per SIMD keep the matrix pipe fed
// while others sit in the load -> LDS -> barrier phase)
__global__ __launch_bounds__(64 * MLP_DW_WAVES) void mlp_dw_kernel(PfMlpTrain p0, const PfMlpTrain* descs) {
    constexpr int NTH = 64 * MLP_DW_WAVES;
    extern __shared__ float lds[];
    const PfMlpTrain p = descs ? descs[blockIdx.z] : p0;
    const int chunk = mlp_chunk(p);
    if ((int)blockIdx.x * chunk >= p.rows || (int)blockIdx.y >= p.nl) return;
    float* part = p.ws;
    const MlpShape sh = mlp_shape(p);
    const MlpDwLayout L = mlp_dw_layout(p, sh);
    const int l = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, q = lane >> 4;
    const int RA = sel3(sh.wo16, l), RB = sel3(L.wb16, l);
    const int lda = RA + 16, ldb = RB + 16;
    float* As = lds;
    float* Bs = lds + MLP_EB * lda;
    const float* asrc = l == p.nl - 1 ? p.dout : sel2(p.dz, l);
    const float* hsrc = l > 0 ? sel2(p.h, l - 1) : nullptr;
    const int wa = sel3(sh.wo, l), wil = sel3(sh.wi, l);
    const int offl = sel3(L.off, l), boffl = sel3(L.boff, l);
    const int NT = RB / 16, NRT = RA / 16;
    int rts[MLP_SLOTS], cts[MLP_SLOTS];
    bool val[MLP_SLOTS];
#pragma unroll
    for (int s = 0; s < MLP_SLOTS; ++s) {
        const int id = wave + MLP_DW_WAVES * s;
        val[s] = id < NRT * NT;
        rts[s] = val[s] ? id / NT : 0; cts[s] = val[s] ? id % NT : 0;
    }
    f4 acc[MLP_SLOTS];
#pragma unroll
    for (int s = 0; s < MLP_SLOTS; ++s) acc[s] = pf_splat(0.f);
    const int r_lo = blockIdx.x * chunk, r_hi = min(p.rows, r_lo + chunk);
    float bsum = 0.f;
    // staging in float4 units, thread t owns units t, t + 256, ...: (row, column) of each unit fixed for the whole kernel.
    // The next block's units are fetched into registers while the current block is multiplied (the products are short:
    // without this every block paid a full memory latency between two barriers)
    constexpr int UA = (MLP_EB * 32 + NTH - 1) / NTH, UB = (MLP_EB * 36 + NTH - 1) / NTH;     // float4 units of a 128- / 144-wide block per thread
    const int ra4 = RA / 4, rb4 = RB / 4;
    int elA[UA], cA[UA], elB[UB], cB[UB];
#pragma unroll
    for (int n = 0; n < UA; ++n) { const int k = threadIdx.x + NTH * n; elA[n] = k / ra4; cA[n] = (k - elA[n] * ra4) * 4; }
#pragma unroll
    for (int n = 0; n < UB; ++n) { const int k = threadIdx.x + NTH * n; elB[n] = k / rb4; cB[n] = (k - elB[n] * rb4) * 4; }
    const int dsh = 31 - __clz(p.cdiv);                 // cdiv is a power of two
    f4 ra[UA], rbv[UB];
    auto fetch = [&](int rb) {
#pragma unroll
        for (int n = 0; n < UA; ++n) {
            f4 v = pf_splat(0.f);
            const int r = rb + elA[n], c = cA[n];
            if (elA[n] < MLP_EB && r < r_hi && c < wa) {
                if ((wa & 3) == 0) v = *reinterpret_cast<const f4*>(asrc + (size_t)r * wa + c);
                else {
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        if (c + w < wa) v[w] = asrc[(size_t)r * wa + c + w];
                }
            }
            ra[n] = v;
        }
#pragma unroll
        for (int n = 0; n < UB; ++n) {
            f4 v = pf_splat(0.f);
            const int r = rb + elB[n], u = cB[n];
            if (elB[n] < MLP_EB && r < r_hi) {
                if (l > 0) { if (u < wil) v = *reinterpret_cast<const f4*>(hsrc + (size_t)r * wil + u); }
                else if (u + 3 < p.cc) v = *reinterpret_cast<const f4*>(p.c + (size_t)(r >> dsh) * p.cc + u);
                else {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const int uu = u + w;
                        if (uu < p.cc) v[w] = p.c[(size_t)(r >> dsh) * p.cc + uu];
                        else if (uu < p.cc + p.td) v[w] = p.y[(size_t)r * p.ldy + uu - p.cc];
                    }
                }
            }
            rbv[n] = v;
        }
    };
    fetch(r_lo);
    for (int rb = r_lo; rb < r_hi; rb += MLP_EB) {
        __syncthreads();
#pragma unroll
        for (int n = 0; n < UA; ++n)
            if (elA[n] < MLP_EB) *reinterpret_cast<f4*>(As + elA[n] * lda + cA[n]) = ra[n];
#pragma unroll
        for (int n = 0; n < UB; ++n)
            if (elB[n] < MLP_EB) *reinterpret_cast<f4*>(Bs + elB[n] * ldb + cB[n]) = rbv[n];
        __syncthreads();
        if (rb + MLP_EB < r_hi) fetch(rb + MLP_EB);
        if (threadIdx.x < RA)
#pragma unroll 8
            for (int el = 0; el < MLP_EB; ++el) bsum += As[el * lda + threadIdx.x];
#pragma unroll
        for (int ks = 0; ks < MLP_EB / 4; ++ks) {
            const float* ar = As + (4 * ks + q) * lda + row;
            const float* br = Bs + (4 * ks + q) * ldb + row;
#pragma unroll
            for (int s = 0; s < MLP_SLOTS; ++s)
                if (val[s]) acc[s] = pf_mfma(ar[rts[s] * 16], br[cts[s] * 16], acc[s]);
        }
    }
    float* out = part + (size_t)blockIdx.x * L.total;
    if (threadIdx.x < RA) out[boffl + threadIdx.x] = bsum;
#pragma unroll
    for (int s = 0; s < MLP_SLOTS; ++s)
        if (val[s])
#pragma unroll
            for (int r = 0; r < 4; ++r) out[offl + (rts[s] * 16 + 4 * q + r) * RB + cts[s] * 16 + row] = acc[s][r];
}

// partial sums -> dW[l] [wo, in_l] (column j < td of layer 0 sits behind the cc conditioning columns in the partials), db[l]
constexpr int MLP_RG = PF_MLP_RG;           // groups of 64 threads that share the chunk range of an output element
__global__ __launch_bounds__(64 * MLP_RG) void mlp_dw_reduce_kernel(PfMlpTrain p0, const PfMlpTrain* descs) {
    const PfMlpTrain p = descs ? descs[blockIdx.z] : p0;
    const float* part = p.ws;
    const int nchunk = (p.rows + mlp_chunk(p) - 1) / mlp_chunk(p);
    const MlpShape sh = mlp_shape(p);
    const MlpDwLayout L = mlp_dw_layout(p, sh);
    int cnt[3] = {0, 0, 0};
#pragma unroll
    for (int l = 0; l < 3; ++l)
        if (l < p.nl) cnt[l] = sh.wo[l] * (sh.in[l] + 1);
    const int total = cnt[0] + cnt[1] + cnt[2];
    __shared__ double shr[MLP_RG][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    const bool ok = i < total;
    int l = 0, c = 0, j = 0, inl = 1, src = 0;
    if (ok) {
        l = i < cnt[0] ? 0 : (i < cnt[0] + cnt[1] ? 1 : 2);
        const int rem = i - (l > 0 ? cnt[0] : 0) - (l > 1 ? cnt[1] : 0);
        inl = sel3(sh.in, l);
        c = rem / (inl + 1); j = rem % (inl + 1);
        if (j == inl) src = sel3(L.boff, l) + c;
        else {
            const int u = l == 0 ? (j < p.td ? p.cc + j : j - p.td) : j;
            src = sel3(L.off, l) + c * sel3(L.wb16, l) + u;
        }
    }
    double s = 0.0;
    if (ok)
    {
        int k = ty;
        for (; k + 3 * MLP_RG < nchunk; k += 4 * MLP_RG) {            // four loads in flight per thread
            const float v0 = part[(size_t)k * L.total + src], v1 = part[(size_t)(k + MLP_RG) * L.total + src];
            const float v2 = part[(size_t)(k + 2 * MLP_RG) * L.total + src], v3 = part[(size_t)(k + 3 * MLP_RG) * L.total + src];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; k < nchunk; k += MLP_RG) s += (double)part[(size_t)k * L.total + src];
    }
    shr[ty][tx] = s;
    __syncthreads();
    if (ty != 0 || !ok) return;
    s = 0.0;
#pragma unroll
    for (int k = 0; k < MLP_RG; ++k) s += shr[k][tx];
    float* db = sel3(p.db, l);
    if (j == inl) { if (db) db[c] = (float)s; }
    else sel3(p.dW, l)[(size_t)c * inl + j] = (float)s;
}

// descriptors of a batched launch travel as kernel arguments of this copy kernel (capturable in a hipGraph, no host copy)
constexpr int MLP_BATCH_MAX = 16;
struct MlpBatch { PfMlpTrain p[MLP_BATCH_MAX]; };
__global__ __launch_bounds__(256) void mlp_desc_upload_kernel(MlpBatch b, unsigned* dst, int nwords) {
    const unsigned* src = reinterpret_cast<const unsigned*>(&b);
    for (int i = threadIdx.x; i < nwords; i += 256) dst[i] = src[i];
}

template <typename KERNEL>
void allow_lds(KERNEL k, size_t bytes) {
    if (bytes > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int mlp_check(const PfMlpTrain* p) {
    if (!p) return PF_ERR_NULL;
    if (p->rows <= 0 || (p->nl != 2 && p->nl != 3) || p->td < 0 || p->td > 3 || p->cdiv < 1) return PF_ERR_SHAPE;
    if (p->cc != 16 && p->cc != 32 && p->cc != 64 && p->cc != 128) return PF_ERR_UNSUPPORTED;
    if (p->cdiv != 1 && p->cdiv != 2 && p->cdiv != 4 && p->cdiv != 8 && p->cdiv != 16) return PF_ERR_UNSUPPORTED;
    if (p->rows % p->cdiv != 0) return PF_ERR_SHAPE;
    if (p->chunk < 0 || p->chunk % MLP_EB != 0) return PF_ERR_SHAPE;
    for (int l = 0; l < p->nl; ++l) {
        if (p->width[l] < 1 || p->width[l] > 128) return PF_ERR_UNSUPPORTED;
        if (l < p->nl - 1 && p->width[l] != 16 && p->width[l] != 32 && p->width[l] != 64 && p->width[l] != 128) return PF_ERR_UNSUPPORTED;
        if (!p->W[l]) return PF_ERR_NULL;
    }
    if (p->td > 0 && (!p->y || p->ldy < p->td)) return PF_ERR_NULL;
    if (!p->c) return PF_ERR_NULL;
    return PF_OK;
}

}  // namespace

extern "C" long long pf_mlp_train_ws_floats(const PfMlpTrain* p) {
    if (mlp_check(p) != PF_OK) return -1;
    const MlpShape sh = mlp_shape(*p);
    const MlpDwLayout L = mlp_dw_layout(*p, sh);
    const int chunk = mlp_chunk(*p);
    const long long nchunk = (p->rows + chunk - 1) / chunk;
    return nchunk * L.total;
}

extern "C" int pf_mlp_train_fwd(const PfMlpTrain* p, void* stream) {
    int st = mlp_check(p);
    if (st) return st;
    if (!p->out) return PF_ERR_NULL;
    for (int l = 0; l < p->nl - 1; ++l)
        if (!p->h[l]) return PF_ERR_NULL;
    const MlpShape sh = mlp_shape(*p);
    size_t lds = 0;
    for (int l = 0; l < p->nl; ++l) lds += (size_t)sh.wo16[l] * (sh.wi16[l] + 4) + sh.wo16[l];
    lds = (lds + (size_t)sh.wo16[0] * 4) * sizeof(float);
    const int ntiles = (p->rows + 15) / 16;
    const int grid = (ntiles + 3) / 4 < MLP_GRID ? (ntiles + 3) / 4 : MLP_GRID;
    hipStream_t s = (hipStream_t)stream;
    if (p->nl == 2) { allow_lds(mlp_fwd_kernel<2>, lds); hipLaunchKernelGGL(mlp_fwd_kernel<2>, dim3(grid), dim3(256), lds, s, *p, (const PfMlpTrain*)nullptr); }
    else { allow_lds(mlp_fwd_kernel<3>, lds); hipLaunchKernelGGL(mlp_fwd_kernel<3>, dim3(grid), dim3(256), lds, s, *p, (const PfMlpTrain*)nullptr); }
    return pf_last_launch_status();
}

extern "C" int pf_mlp_train_bwd(const PfMlpTrain* p, void* stream) {
    int st = mlp_check(p);
    if (st) return st;
    if (!p->dout || !p->ws) return PF_ERR_NULL;
    for (int l = 0; l < p->nl - 1; ++l)
        if (!p->h[l] || !p->dz[l]) return PF_ERR_NULL;
    for (int l = 0; l < p->nl; ++l)
        if (!p->dW[l]) return PF_ERR_NULL;
    if (p->ws_floats < pf_mlp_train_ws_floats(p)) return PF_ERR_WORKSPACE;
    const MlpShape sh = mlp_shape(*p);
    const MlpDwLayout L = mlp_dw_layout(*p, sh);
    hipStream_t s = (hipStream_t)stream;
    {
        size_t lds = 0;
        for (int l = 0; l < p->nl; ++l) lds += (size_t)sh.wi16[l] * (sh.wo16[l] + 4);
        lds = (lds + (size_t)sh.wo16[0] * 4) * sizeof(float);
        const int ntiles = (p->rows + 15) / 16;
        const int grid = (ntiles + 3) / 4 < MLP_GRID ? (ntiles + 3) / 4 : MLP_GRID;
        if (p->nl == 2) { allow_lds(mlp_bwd_kernel<2>, lds); hipLaunchKernelGGL(mlp_bwd_kernel<2>, dim3(grid), dim3(256), lds, s, *p, (const PfMlpTrain*)nullptr); }
        else { allow_lds(mlp_bwd_kernel<3>, lds); hipLaunchKernelGGL(mlp_bwd_kernel<3>, dim3(grid), dim3(256), lds, s, *p, (const PfMlpTrain*)nullptr); }
    }
    const int chunk = mlp_chunk(*p);
    const int nchunk = (p->rows + chunk - 1) / chunk;
    {
        int ramax = 0, rbmax = 0;
        for (int l = 0; l < p->nl; ++l) { ramax = ramax > sh.wo16[l] ? ramax : sh.wo16[l]; rbmax = rbmax > L.wb16[l] ? rbmax : L.wb16[l]; }
        const size_t lds = sizeof(float) * (size_t)MLP_EB * ((ramax + 16) + (rbmax + 16));
        hipLaunchKernelGGL(mlp_dw_kernel, dim3(nchunk, p->nl), dim3(64 * MLP_DW_WAVES), lds, s, *p, (const PfMlpTrain*)nullptr);
    }
    int total = 0;
    for (int l = 0; l < p->nl; ++l) total += sh.wo[l] * (sh.in[l] + 1);
    hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3((total + 63) / 64), dim3(64 * MLP_RG), 0, s, *p, (const PfMlpTrain*)nullptr);
    return pf_last_launch_status();
}

// ---- batched launches: n <= 16 networks of the same depth (e.g. the scale / shift conditioners of all six flow blocks, which
// depend only on the conditioning features) in ONE launch per kernel, blockIdx.z = network.  dev_descs: n * sizeof(PfMlpTrain)
// bytes of device scratch (the descriptors are copied there by a kernel whose arguments they are).
extern "C" int pf_mlp_train_fwd_batch(const PfMlpTrain* descs, int n, void* dev_descs, void* stream) {
    if (!descs || !dev_descs) return PF_ERR_NULL;
    if (n < 1 || n > MLP_BATCH_MAX) return PF_ERR_SHAPE;
    size_t lds = 0;
    int gmax = 1;
    MlpBatch b{};
    for (int k = 0; k < n; ++k) {
        const PfMlpTrain* p = descs + k;
        int st = mlp_check(p);
        if (st) return st;
        if (p->nl != descs[0].nl || !p->out) return PF_ERR_SHAPE;
        for (int l = 0; l < p->nl - 1; ++l)
            if (!p->h[l]) return PF_ERR_NULL;
        const MlpShape sh = mlp_shape(*p);
        size_t w = 0;
        for (int l = 0; l < p->nl; ++l) w += (size_t)sh.wo16[l] * (sh.wi16[l] + 4) + sh.wo16[l];
        w = (w + (size_t)sh.wo16[0] * 4) * sizeof(float);
        lds = w > lds ? w : lds;
        const int ntiles = (p->rows + 15) / 16;
        const int g = (ntiles + 3) / 4 < MLP_GRID ? (ntiles + 3) / 4 : MLP_GRID;
        gmax = g > gmax ? g : gmax;
        b.p[k] = *p;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mlp_desc_upload_kernel, dim3(1), dim3(256), 0, s, b, (unsigned*)dev_descs, (int)(n * sizeof(PfMlpTrain) / 4));
    const PfMlpTrain* dd = (const PfMlpTrain*)dev_descs;
    if (descs[0].nl == 2) { allow_lds(mlp_fwd_kernel<2>, lds); hipLaunchKernelGGL(mlp_fwd_kernel<2>, dim3(gmax, 1, n), dim3(256), lds, s, b.p[0], dd); }
    else { allow_lds(mlp_fwd_kernel<3>, lds); hipLaunchKernelGGL(mlp_fwd_kernel<3>, dim3(gmax, 1, n), dim3(256), lds, s, b.p[0], dd); }
    return pf_last_launch_status();
}

extern "C" int pf_mlp_train_bwd_batch(const PfMlpTrain* descs, int n, void* dev_descs, void* stream) {
    if (!descs || !dev_descs) return PF_ERR_NULL;
    if (n < 1 || n > MLP_BATCH_MAX) return PF_ERR_SHAPE;
    size_t lds_b = 0, lds_w = 0;
    int gmax = 1, cmax = 1, tmax = 1;
    MlpBatch b{};
    for (int k = 0; k < n; ++k) {
        const PfMlpTrain* p = descs + k;
        int st = mlp_check(p);
        if (st) return st;
        if (p->nl != descs[0].nl || !p->dout || !p->ws) return PF_ERR_NULL;
        for (int l = 0; l < p->nl - 1; ++l)
            if (!p->h[l] || !p->dz[l]) return PF_ERR_NULL;
        for (int l = 0; l < p->nl; ++l)
            if (!p->dW[l]) return PF_ERR_NULL;
        if (p->ws_floats < pf_mlp_train_ws_floats(p)) return PF_ERR_WORKSPACE;
        const MlpShape sh = mlp_shape(*p);
        const MlpDwLayout L = mlp_dw_layout(*p, sh);
        size_t w = 0;
        for (int l = 0; l < p->nl; ++l) w += (size_t)sh.wi16[l] * (sh.wo16[l] + 4);
        w = (w + (size_t)sh.wo16[0] * 4) * sizeof(float);
        lds_b = w > lds_b ? w : lds_b;
        int ramax = 0, rbmax = 0, total = 0;
        for (int l = 0; l < p->nl; ++l) {
            ramax = ramax > sh.wo16[l] ? ramax : sh.wo16[l]; rbmax = rbmax > L.wb16[l] ? rbmax : L.wb16[l];
            total += sh.wo[l] * (sh.in[l] + 1);
        }
        const size_t w2 = sizeof(float) * (size_t)MLP_EB * ((ramax + 16) + (rbmax + 16));
        lds_w = w2 > lds_w ? w2 : lds_w;
        const int ntiles = (p->rows + 15) / 16;
        const int g = (ntiles + 3) / 4 < MLP_GRID ? (ntiles + 3) / 4 : MLP_GRID;
        gmax = g > gmax ? g : gmax;
        const int nchunk = (p->rows + mlp_chunk(*p) - 1) / mlp_chunk(*p);
        cmax = nchunk > cmax ? nchunk : cmax;
        tmax = total > tmax ? total : tmax;
        b.p[k] = *p;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mlp_desc_upload_kernel, dim3(1), dim3(256), 0, s, b, (unsigned*)dev_descs, (int)(n * sizeof(PfMlpTrain) / 4));
    const PfMlpTrain* dd = (const PfMlpTrain*)dev_descs;
    if (descs[0].nl == 2) { allow_lds(mlp_bwd_kernel<2>, lds_b); hipLaunchKernelGGL(mlp_bwd_kernel<2>, dim3(gmax, 1, n), dim3(256), lds_b, s, b.p[0], dd); }
    else { allow_lds(mlp_bwd_kernel<3>, lds_b); hipLaunchKernelGGL(mlp_bwd_kernel<3>, dim3(gmax, 1, n), dim3(256), lds_b, s, b.p[0], dd); }
    hipLaunchKernelGGL(mlp_dw_kernel, dim3(cmax, descs[0].nl, n), dim3(64 * MLP_DW_WAVES), lds_w, s, b.p[0], dd);
    hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3((tmax + 63) / 64, 1, n), dim3(64 * MLP_RG), 0, s, b.p[0], dd);
    return pf_last_launch_status();
}

// weight gradients only: dz / h / dout of every network are already in memory (csrc/train_flowchain.hip writes them from its
// own chain kernel); split-K launch + reduction over all n networks
extern "C" int pf_mlp_train_dw_batch(const PfMlpTrain* descs, int n, void* dev_descs, void* stream) {
    if (!descs || !dev_descs) return PF_ERR_NULL;
    if (n < 1 || n > MLP_BATCH_MAX) return PF_ERR_SHAPE;
    size_t lds_w = 0;
    int cmax = 1, tmax = 1;
    MlpBatch b{};
    for (int k = 0; k < n; ++k) {
        const PfMlpTrain* p = descs + k;
        int st = mlp_check(p);
        if (st) return st;
        if (p->nl != descs[0].nl || !p->dout || !p->ws) return PF_ERR_NULL;
        for (int l = 0; l < p->nl - 1; ++l)
            if (!p->h[l] || !p->dz[l]) return PF_ERR_NULL;
        for (int l = 0; l < p->nl; ++l)
            if (!p->dW[l]) return PF_ERR_NULL;
        if (p->ws_floats < pf_mlp_train_ws_floats(p)) return PF_ERR_WORKSPACE;
        const MlpShape sh = mlp_shape(*p);
        const MlpDwLayout L = mlp_dw_layout(*p, sh);
        int ramax = 0, rbmax = 0, total = 0;
        for (int l = 0; l < p->nl; ++l) {
            ramax = ramax > sh.wo16[l] ? ramax : sh.wo16[l]; rbmax = rbmax > L.wb16[l] ? rbmax : L.wb16[l];
            total += sh.wo[l] * (sh.in[l] + 1);
        }
        const size_t w2 = sizeof(float) * (size_t)MLP_EB * ((ramax + 16) + (rbmax + 16));
        lds_w = w2 > lds_w ? w2 : lds_w;
        const int nchunk = (p->rows + mlp_chunk(*p) - 1) / mlp_chunk(*p);
        cmax = nchunk > cmax ? nchunk : cmax;
        tmax = total > tmax ? total : tmax;
        b.p[k] = *p;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mlp_desc_upload_kernel, dim3(1), dim3(256), 0, s, b, (unsigned*)dev_descs, (int)(n * sizeof(PfMlpTrain) / 4));
    const PfMlpTrain* dd = (const PfMlpTrain*)dev_descs;
    hipLaunchKernelGGL(mlp_dw_kernel, dim3(cmax, descs[0].nl, n), dim3(64 * MLP_DW_WAVES), lds_w, s, b.p[0], dd);
    hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3((tmax + 63) / 64, 1, n), dim3(64 * MLP_RG), 0, s, b.p[0], dd);
    return pf_last_launch_status();
}
