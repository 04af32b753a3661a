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

#ifndef PF_MLP_GRID
#define PF_MLP_GRID 512
#endif
constexpr int MLP_GRID = PF_MLP_GRID;
#ifndef PF_MLP_DW_WAVES
#define PF_MLP_DW_WAVES 4
#endif
#ifndef PF_MLP_RG
#define PF_MLP_RG 16
#endif
#ifndef PF_DW_ABL
#define PF_DW_ABL 0            // timing-only ablations of mlp_dw_kernel (tools/time_mlpdw.py): 1 no MFMA block, 2 no loads in the loop, 4 no LDS stores
#endif
#ifndef PF_MLP_STAGE
#define PF_MLP_STAGE 16
#endif
constexpr int MLP_STAGE = PF_MLP_STAGE;   // weight elements a thread has in flight while a workgroup stages a layer's weights into LDS
constexpr int MLP_LD = 144;            // LDS row stride of the weight-gradient kernel's staged blocks: >= 128 / 144 columns, = 16 (mod 32) floats
#ifndef PF_MLP_EB
#define PF_MLP_EB 32
#endif
constexpr int MLP_EB = PF_MLP_EB, MLP_DW_WAVES = PF_MLP_DW_WAVES, MLP_SLOTS = (36 + MLP_DW_WAVES - 1) / MLP_DW_WAVES;   // 36 = 4 x 9 tiles of the widest layer
// rows per split-K chunk (multiple of MLP_EB): the descriptor's choice (batched launches have networks x layers of parallelism
// already and want long chunks: fewer partial sums to write and add), else sized so that one network fills the chip
#ifndef PF_MLP_CHUNK_BIG
#define PF_MLP_CHUNK_BIG 128
#endif
__host__ __device__ inline int mlp_chunk(const PfMlpTrain& p) { return p.chunk > 0 ? p.chunk : (p.rows > 16384 ? PF_MLP_CHUNK_BIG : 64); }

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
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* rflp(T* ptr) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// A descriptor that reaches a kernel through `descs ? descs[blockIdx.z] : p0` carries GENERIC pointers (only the pointer members
// of a by-value kernel argument are known to be global): every access through them was a flat_load / flat_store, which counts
// on the LDS counter as well as the memory counter - each `s_waitcnt lgkmcnt` in front of an LDS-fed MFMA then also waited for
// the global prefetch in flight.  MlpG = the descriptor's pointers typed as what they are (address space 1); a cast to
// address space 1 and back is folded away by hipcc, only an access THROUGH the typed pointer becomes global_load / global_store.
#define PF_G __attribute__((address_space(1)))
typedef const float PF_G* gcp;
typedef float PF_G* gp;
typedef const f4 PF_G* gc4p;
typedef f4 PF_G* g4p;
struct MlpG {
    gcp y, c, W[3], b[3], dout;
    gp h[2], out, dz[2], dy, dc, dW[3], db[3], ws;
};
__device__ __forceinline__ MlpG mlp_glob(const PfMlpTrain& p) {
    MlpG g;
    g.y = (gcp)p.y; g.c = (gcp)p.c; g.dout = (gcp)p.dout; g.out = (gp)p.out; g.dy = (gp)p.dy; g.dc = (gp)p.dc; g.ws = (gp)p.ws;
#pragma unroll
    for (int i = 0; i < 3; ++i) { g.W[i] = (gcp)p.W[i]; g.b[i] = (gcp)p.b[i]; g.dW[i] = (gp)p.dW[i]; g.db[i] = (gp)p.db[i]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { g.h[i] = (gp)p.h[i]; g.dz[i] = (gp)p.dz[i]; }
    return g;
}

// the descriptor of this workgroup's network with wave-uniform shapes and global pointers (see above): with the fields as they
// come out of the vector loads hipcc treats every shape derived from them as lane-dependent - exec-masked regions around each
// tile's read + MFMA, one LDS read waited for per MFMA (mlp_dw_kernel: ~7 us per 32-row block of the widest layer)
// Up to 16 descriptors travel as ONE kernel argument (16 x 248 B = 3968 B of the 4 KB): network blockIdx.z of a batched launch,
// entry 0 of a single one.  (Round 5: they used to be copied into device memory by a launch of their own in front of every
// batched call - six launches per training step - and came back through vector loads.)
constexpr int MLP_BATCH_MAX = 16;
struct MlpBatch { PfMlpTrain p[MLP_BATCH_MAX]; };
static_assert(sizeof(MlpBatch) <= 4032, "kernel argument block");
__device__ __forceinline__ PfMlpTrain mlp_desc(const MlpBatch& bb) {
    PfMlpTrain p = bb.p[blockIdx.z];
    p.rows = rfl(p.rows); p.nl = rfl(p.nl); p.td = rfl(p.td); p.cc = rfl(p.cc); p.cdiv = rfl(p.cdiv); p.ldy = rfl(p.ldy); p.chunk = rfl(p.chunk);
#pragma unroll
    for (int i = 0; i < 3; ++i) { p.width[i] = rfl(p.width[i]); p.W[i] = rflp(p.W[i]); p.b[i] = rflp(p.b[i]); p.dW[i] = rflp(p.dW[i]); p.db[i] = rflp(p.db[i]); }
#pragma unroll
    for (int i = 0; i < 2; ++i) { p.h[i] = rflp(p.h[i]); p.dz[i] = rflp(p.dz[i]); }
    p.y = rflp(p.y); p.c = rflp(p.c); p.out = rflp(p.out); p.dout = rflp(p.dout); p.dy = rflp(p.dy); p.dc = rflp(p.dc); p.ws = rflp(p.ws);
    return p;
}

template <int NL>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(MlpBatch bb) {
    extern __shared__ float lds[];
    const PfMlpTrain p = mlp_desc(bb);
    const MlpG G = mlp_glob(p);
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
        for (int i0 = threadIdx.x; i0 < sh.wo16[l] * sh.wi16[l]; i0 += 256 * MLP_STAGE) {
            float v[MLP_STAGE];
#pragma unroll
            for (int k = 0; k < MLP_STAGE; ++k) {
                const int i = i0 + 256 * k, c = i >> sft, u = i & (sh.wi16[l] - 1);
                v[k] = (i < sh.wo16[l] * sh.wi16[l] && c < sh.wo[l] && u < sh.wi[l]) ? G.W[l][(size_t)c * sh.in[l] + off + u] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < MLP_STAGE; ++k) {
                const int i = i0 + 256 * k, c = i >> sft, u = i & (sh.wi16[l] - 1);
                if (i < sh.wo16[l] * sh.wi16[l]) Wl[l][c * ld + u] = v[k];
            }
        }
        for (int i = threadIdx.x; i < sh.wo16[l]; i += 256) bl[l][i] = (G.b[l] && i < sh.wo[l]) ? G.b[l][i] : 0.f;
    }
    for (int i = threadIdx.x; i < sh.wo16[0] * 4; i += 256) {
        const int c = i >> 2, j = i & 3;
        Wx[i] = (c < sh.wo[0] && j < p.td) ? G.W[0][(size_t)c * sh.in[0] + j] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int p0 = tile * 16 + col;
        const bool valid = p0 < p.rows;
        const int pr = valid ? p0 : p.rows - 1;
        f4 act[8];
        gcp crow = G.c + (size_t)(pr / p.cdiv) * p.cc;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            act[cb] = pf_splat(0.f);
            if (cb * 16 < sh.wi16[0]) act[cb] = *(gc4p)(crow + cb * 16 + 4 * q);
        }
        float xv[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < p.td) xv[j] = G.y[(size_t)pr * p.ldy + j];
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
                        if (valid) *(g4p)(G.h[lh] + (size_t)p0 * sh.wo[l] + ob * 16 + 4 * q) = acc;
                        nxt[ob] = acc;
                    } else if (valid) {
                        if ((sh.wo[l] & 15) == 0) *(g4p)(G.out + (size_t)p0 * sh.wo[l] + ob * 16 + 4 * q) = acc;
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int ch = ob * 16 + 4 * q + r;
                                if (ch < sh.wo[l]) G.out[(size_t)p0 * sh.wo[l] + ch] = acc[r];
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
__global__ __launch_bounds__(256) void mlp_bwd_kernel(MlpBatch bb) {
    extern __shared__ float lds[];
    const PfMlpTrain p = mlp_desc(bb);
    const MlpG G = mlp_glob(p);
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
        for (int i0 = threadIdx.x; i0 < sh.wo16[l] * sh.wi16[l]; i0 += 256 * MLP_STAGE) {   // consecutive threads: consecutive u (contiguous in W)
            float v[MLP_STAGE];
#pragma unroll
            for (int k = 0; k < MLP_STAGE; ++k) {
                const int i = i0 + 256 * k, c = i >> sft, u = i & (sh.wi16[l] - 1);
                v[k] = (i < sh.wo16[l] * sh.wi16[l] && c < sh.wo[l] && u < sh.wi[l]) ? G.W[l][(size_t)c * sh.in[l] + off + u] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < MLP_STAGE; ++k) {
                const int i = i0 + 256 * k, c = i >> sft, u = i & (sh.wi16[l] - 1);
                if (i < sh.wo16[l] * sh.wi16[l]) Wt[l][u * ld + c] = v[k];
            }
        }
    }
    for (int i = threadIdx.x; i < sh.wo16[0] * 4; i += 256) {
        const int c = i >> 2, j = i & 3;
        Wx[i] = (c < sh.wo[0] && j < p.td) ? G.W[0][(size_t)c * sh.in[0] + j] : 0.f;
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
                        if (ch < w) g[cb] = *(gc4p)(G.dout + (size_t)p0 * w + ch);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (ch + r < w) g[cb][r] = G.dout[(size_t)p0 * w + ch + r];
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
                if (q == 0 && valid && G.dy) {
                    const float sv[3] = {s0, s1, s2};
                    for (int j = 0; j < p.ldy; ++j) G.dy[(size_t)p0 * p.ldy + j] = (j < p.td && j < 3) ? sv[j < 3 ? j : 0] : 0.f;
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
                        const f4 hv = *(gc4p)(G.h[lm] + (size_t)pr * sh.wi[l] + ub * 16 + 4 * q);
                        const float sl = p.slope[lm];
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] *= hv[r] > 0.f ? 1.f : sl;
                        if (valid) *(g4p)(G.dz[lm] + (size_t)p0 * sh.wi[l] + ub * 16 + 4 * q) = acc;
                        nxt[ub] = acc;
                    } else if (G.dc) {                    // sum over the cdiv replicas of a conditioning row: adjacent columns
#pragma unroll
                        for (int m = 1; m < 16; m <<= 1)
                            if (m < p.cdiv) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[r] += __shfl_xor(acc[r], m);
                            }
                        if (valid && (col % p.cdiv) == 0)
                            *(g4p)(G.dc + (size_t)(p0 / p.cdiv) * p.cc + ub * 16 + 4 * q) = acc;
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

// one staged 32-row block of the split-K weight-gradient product for the NP output tiles of a wave (tile s = row tile rts[s],
// column tile cts[s]); SAME: all in row tile rts[0]
template <int NP, bool SAME>
__device__ __forceinline__ void dw_block(const float* ar0, const float* br0, const int (&rts)[MLP_SLOTS],
                                         const int (&cts)[MLP_SLOTS], f4 (&acc)[MLP_SLOTS]) {
#pragma unroll
    for (int ks = 0; ks < MLP_EB / 4; ++ks) {
        const float* ar = ar0 + 4 * ks * MLP_LD;
        const float* br = br0 + 4 * ks * MLP_LD;
        float a[NP], b[NP];
        a[0] = ar[rts[0] * 16];
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            b[s] = br[cts[s] * 16];
            if (s > 0) a[s] = SAME ? a[0] : ar[rts[s] * 16];
        }
#pragma unroll
        for (int s = 0; s < NP; ++s) acc[s] = pf_mfma(a[s], b[s], acc[s]);
    }
}

// Products with a NARROW side on the vector ALU: S[w][n] = sum over rows [r_lo, r_hi) of wide[row, w] nar[row, n], n < NN <= 3
// (the last layer of a conditioner has 1 - 3 outputs: dW = dout^T h; the td <= 3 coordinate columns of layer 0: dz0^T y), with the
// column sums of both operands (whichever is the layer's A gives its bias gradient).  As a 32-row staged MFMA block such a product
// fills 4 of a block's 36 tile slots and still pays the block's load -> LDS -> two barriers: the workgroup's time was its 16
// blocks' latency.  Here: 16-byte loads of the wide rows straight into registers (W / 4 column groups x 256 / (W / 4) row lanes,
// four rows in flight per thread), the row lanes' sums joined through LDS in a fixed order.
// WIDE_OUT: S is written as out[off + w RB + col0 + n] (w = output channel), the bias from the wide sums; else as
// out[off + n RB + w] (n = output channel), the bias from the narrow sums.
template <bool WIDE_OUT>
__device__ __forceinline__ void dw_narrow(gcp wide, int W, gcp nar, int ldn, int NN, int r_lo, int r_hi, float* lds, gp out,
                                          int off, int boff, int RB, int col0) {
    const int cgs = W >> 2, RL = 256 / cgs;
    const int cg = threadIdx.x % cgs, rl = threadIdx.x / cgs;
    f4 acc0 = pf_splat(0.f), acc1 = pf_splat(0.f), acc2 = pf_splat(0.f), wsum = pf_splat(0.f);
    float ns0 = 0.f, ns1 = 0.f, ns2 = 0.f;
    const int j1 = NN > 1 ? 1 : 0, j2 = NN > 2 ? 2 : 0;
    for (int r = r_lo + rl; r < r_hi; r += 4 * RL) {
        f4 w[4];
        float n0[4], n1[4], n2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * RL;
            const bool ok = rr < r_hi;
            const size_t rc = ok ? rr : r_lo;
            const f4 wv = *(gc4p)(wide + rc * W + cg * 4);
            const float a = nar[rc * ldn], b = nar[rc * ldn + j1], c = nar[rc * ldn + j2];
            w[u] = ok ? wv : pf_splat(0.f);
            n0[u] = ok ? a : 0.f; n1[u] = (ok && NN > 1) ? b : 0.f; n2[u] = (ok && NN > 2) ? c : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc0 += w[u] * n0[u]; acc1 += w[u] * n1[u]; acc2 += w[u] * n2[u]; wsum += w[u];
            ns0 += n0[u]; ns1 += n1[u]; ns2 += n2[u];
        }
    }
    // [rl][w] -> (S[w][0], S[w][1], S[w][2], wsum[w]); behind it [rl] -> the narrow sums (column group 0 holds every row once)
    f4* l4 = reinterpret_cast<f4*>(lds);
#pragma unroll
    for (int c = 0; c < 4; ++c) l4[rl * W + cg * 4 + c] = f4{acc0[c], acc1[c], acc2[c], wsum[c]};
    f4* ln = l4 + RL * W;
    if (cg == 0) ln[rl] = f4{ns0, ns1, ns2, 0.f};
    __syncthreads();
    for (int wv = threadIdx.x; wv < W; wv += 256) {
        f4 t = l4[wv];
        for (int k = 1; k < RL; ++k) t += l4[k * W + wv];
        if (WIDE_OUT) {
            out[off + wv * RB + col0] = t[0];
            if (NN > 1) out[off + wv * RB + col0 + 1] = t[1];
            if (NN > 2) out[off + wv * RB + col0 + 2] = t[2];
            out[boff + wv] = t[3];
        } else {
            out[off + wv] = t[0];
            if (NN > 1) out[off + RB + wv] = t[1];
            if (NN > 2) out[off + 2 * RB + wv] = t[2];
        }
    }
    if (!WIDE_OUT && threadIdx.x == 0) {
        f4 t = ln[0];
        for (int k = 1; k < RL; ++k) t += ln[k];
        out[boff] = t[0];
        if (NN > 1) out[boff + 1] = t[1];
        if (NN > 2) out[boff + 2] = t[2];
    }
}

// MLP_DW_WAVES waves share one staged block (as csrc/train_fused.hip ec_dw_kernel: more waves per SIMD keep the matrix pipe fed
// while others sit in the load -> LDS -> barrier phase)
#ifndef PF_MLP_DW_NARROW
#define PF_MLP_DW_NARROW 1
#endif
#ifndef PF_DW_OCC
#define PF_DW_OCC 1
#endif
__global__ __launch_bounds__(64 * MLP_DW_WAVES, PF_DW_OCC) void mlp_dw_kernel(MlpBatch bb) {
    constexpr int NTH = 64 * MLP_DW_WAVES;
    extern __shared__ float lds[];
    const PfMlpTrain p = mlp_desc(bb);
    const MlpG G = mlp_glob(p);
    const int chunk = mlp_chunk(p);
    // PF_MLP_DW_DZSUM: layer 0 in two pieces - the conditioning columns from the replica-summed dz0 over rows / cdiv rows (the extra
    // blockIdx.y = nl), the td coordinate columns over all rows (blockIdx.y = 0, ONE column tile) - both into chunk blockIdx.x's
    // partial tile, disjoint column tiles, so the reduction does not know
    const bool split0 = (p.flags & PF_MLP_DW_DZSUM) != 0 && p.cdiv > 1;
    const bool cpart = split0 && (int)blockIdx.y == p.nl;
    if ((int)blockIdx.x * chunk >= p.rows || ((int)blockIdx.y >= p.nl && !cpart)) return;
    gp part = G.ws;
    const MlpShape sh = mlp_shape(p);
    const MlpDwLayout L = mlp_dw_layout(p, sh);
    const int l = cpart ? 0 : (int)blockIdx.y;
    const bool ypart = split0 && !cpart && l == 0;
#if PF_MLP_DW_NARROW
    {
        const int r_lo = blockIdx.x * chunk, r_hi = min(p.rows, r_lo + chunk);
        gp outp = part + (size_t)blockIdx.x * L.total;
        if (ypart) {                                   // dz0^T y: [64, td] behind the conditioning columns; dz0's column sums = the bias gradient
            dw_narrow<true>((gcp)G.dz[0], sh.wo[0], G.y, p.ldy, p.td, r_lo, r_hi, lds, outp, L.off[0], L.boff[0], L.wb16[0], p.cc);
            return;
        }
        const int wol = sel3(sh.wo, l), wil_ = sel3(sh.wi, l);
        if (!cpart && l == p.nl - 1 && l > 0 && wol <= 3 && (wil_ == 16 || wil_ == 32 || wil_ == 64 || wil_ == 128)) {
            dw_narrow<false>((gcp)sel2(G.h, l - 1), wil_, G.dout, wol, wol, r_lo, r_hi, lds, outp, sel3(L.off, l), sel3(L.boff, l),
                             sel3(L.wb16, l), 0);
            return;
        }
    }
#endif
    // the wave index through readfirstlane: everything derived from it (a wave's tile list, the branch on its tile count) is then
    // wave-uniform TO THE COMPILER - with `threadIdx.x >> 6` hipcc put every tile's read + MFMA under its own exec mask
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), row = lane & 15, q = lane >> 4;
    const int RA = sel3(sh.wo16, l), RB = sel3(L.wb16, l);
    // row stride = 16 (mod 32) floats: the two k rows a 32-lane group of a ds_read_b32 touches sit 16 banks apart (a 144-wide
    // block padded to 160 put them on the SAME banks: every B read of the widest layer was a 2-way conflict)
    // (and a COMPILE-TIME stride: with run-time strides every (k step, tile) read kept its own address register - 150 VGPRs of
    // addresses, 328 in all - and hipcc, scheduling for register pressure, issued read -> wait -> MFMA one at a time)
    constexpr int lda = MLP_LD, ldb = MLP_LD;
    float* As = lds;
    float* Bs = lds + MLP_EB * lda;
    gcp asrc = cpart ? (gcp)G.dc : (l == p.nl - 1 ? G.dout : (gcp)sel2(G.dz, l));
    gcp hsrc = l > 0 ? (gcp)sel2(G.h, l - 1) : (gcp)nullptr;
    const int wa = sel3(sh.wo, l), wil = sel3(sh.wi, l);
    const int offl = sel3(L.off, l), boffl = sel3(L.boff, l);
    const int ct0 = ypart ? p.cc / 16 : 0;                                   // first column tile / column tiles of this workgroup
    const int NT = ypart ? 1 : (cpart ? p.cc / 16 : RB / 16), NRT = RA / 16;
    // a wave's output tiles are CONSECUTIVE ids (row tile major): with 4 row tiles (64 output channels) and 4 waves they are one
    // row tile's columns, so the A fragment of a k step is read once for all of them (round 4 dealt the tiles round-robin: two
    // ds_read_b32 per MFMA, the LDS pipe next to saturated)
    int rts[MLP_SLOTS], cts[MLP_SLOTS];
    bool val[MLP_SLOTS];
    const int nper = (NRT * NT + MLP_DW_WAVES - 1) / MLP_DW_WAVES;
#pragma unroll
    for (int s = 0; s < MLP_SLOTS; ++s) {
        const int id = wave * nper + s;
        val[s] = s < nper && id < NRT * NT;
        rts[s] = val[s] ? id / NT : 0; cts[s] = ct0 + (val[s] ? id % NT : 0);
    }
    bool same = true;                                  // all of this wave's tiles in one row tile: one A read per k step
#pragma unroll
    for (int s = 1; s < MLP_SLOTS; ++s) same = same && (!val[s] || rts[s] == rts[0]);
    const int nper_w = min(nper, max(0, NRT * NT - wave * nper));     // tiles this wave really owns (the last wave may own fewer)
    f4 acc[MLP_SLOTS];
#pragma unroll
    for (int s = 0; s < MLP_SLOTS; ++s) acc[s] = pf_splat(0.f);
    const int dsh = 31 - __clz(p.cdiv);                 // cdiv is a power of two
    const int esh = cpart ? dsh : 0;                     // the conditioning piece walks rows / cdiv summed rows (chunk and rows are multiples of cdiv)
    const int r_lo = blockIdx.x * (chunk >> esh), r_hi = min(p.rows >> esh, r_lo + (chunk >> esh));
    // (true by the early return above; without it hipcc guards the block loop and the kernel comes out at 174 + 104 registers,
    // one wave per SIMD, instead of 123 + 72, two)
    __builtin_assume(r_lo < r_hi);
    // staging in float4 units: a block's A image is 32 rows x RA columns, its B image 32 rows x bw columns (layer 0: the cc
    // conditioning columns; the td <= 3 columns behind them come from y, one scalar per thread).  RA / 4 and bw / 4 divide the
    // 256 threads, so unit n of a thread is ITS unit 0 moved down by n * (256 / units-per-row) rows: one pointer, one LDS
    // address and uniform strides per operand (round 4 kept a pointer, a row, a column and an LDS address per unit - with the
    // fully unrolled k loop that was 250+ VGPRs, two waves per SIMD).  Every load is issued UNCONDITIONALLY through an address
    // that is valid even when the unit is not (its value is then replaced by zero): the per-unit `if (valid) load` became ~120
    // exec-masked regions with their own waits.
    constexpr int UN = (MLP_EB * 32 + NTH - 1) / NTH;                  // <= 4 units per thread and operand (128 columns)
    const bool vecA = (wa & 3) == 0;                   // the last layer of a conditioner has 1 - 3 output columns: scalar loads
    const int ra4 = RA / 4;
    const int bw = l > 0 ? wil : p.cc, bw4 = bw / 4;    // both are multiples of 16
    gcp bsrc = l > 0 ? hsrc : G.c;
    const int elA0 = threadIdx.x / ra4, cA = (threadIdx.x - elA0 * ra4) * 4, dEA = NTH / ra4;
    const int elB0 = threadIdx.x / bw4, cB = (threadIdx.x - elB0 * bw4) * 4, dEB = NTH / bw4;
    const bool colA = cA < wa;
    gcp pA = asrc + (size_t)(r_lo + min(elA0, MLP_EB - 1)) * wa + (colA ? cA : 0);
    const int bsh = (l > 0 || cpart) ? 0 : dsh;         // layer 0 reads the conditioning row of point (row >> dsh)
    int rowB = r_lo + elB0;
    const long long offA = (long long)dEA * wa;        // unit n + 1 from unit n
    const long long stepA = (long long)MLP_EB * wa;
    float* const ldsA = As + elA0 * lda + cA;
    float* const ldsB = Bs + elB0 * ldb + cB;
    // layer 0: the td <= 3 columns behind the conditioning part come from y (row stride ldy): one scalar per thread and block
    const bool ycol = l == 0 && !cpart && p.td > 0 && (int)threadIdx.x < MLP_EB * p.td;
    const int yel = ycol ? threadIdx.x / p.td : 0, yw = ycol ? threadIdx.x - yel * p.td : 0;
    gcp pY = ycol ? G.y + (size_t)(r_lo + yel) * p.ldy + yw : bsrc;
    const long long stepY = ycol ? (long long)MLP_EB * p.ldy : 0;
    // bias gradients = column sums of A: thread (column, row group) sums its rows of every block from LDS
    const int bcol = threadIdx.x & (RA - 1), bgrp = threadIdx.x / RA, brows = MLP_EB / (NTH / RA);
    float bsum = 0.f;
    // zero once what no block store writes: the padding columns read by the 16-wide column tiles
    for (int i = threadIdx.x; i < MLP_EB * (lda + ldb); i += NTH) As[i] = 0.f;
    f4 ra[UN], rbv[UN];
    float yv = 0.f;
    auto fetch = [&](int rb) {
#pragma unroll
        for (int n = 0; n < UN; ++n) {
            const bool ok = colA && elA0 + n * dEA < MLP_EB && rb + elA0 + n * dEA < r_hi;
            gcp ptr = ok ? pA + n * offA : asrc;
            f4 v;
            if (vecA) v = *(gc4p)ptr;
            else {
                v = pf_splat(0.f);
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const float x = ptr[w < wa ? w : 0];
                    v[w] = w < wa ? x : 0.f;
                }
            }
            ra[n] = ok ? v : pf_splat(0.f);
        }
        pA += stepA;
#pragma unroll
        for (int n = 0; n < UN; ++n) {
            const bool ok = !ypart && elB0 + n * dEB < MLP_EB && rb + elB0 + n * dEB < r_hi;
            const f4 v = *(gc4p)(ok ? bsrc + (size_t)((rowB + n * dEB) >> bsh) * bw + cB : bsrc);
            rbv[n] = ok ? v : pf_splat(0.f);
        }
        rowB += MLP_EB;
        {
            const float x = *pY;
            yv = (ycol && rb + yel < r_hi) ? x : 0.f;
            pY += stepY;
        }
    };
    fetch(r_lo);
    for (int rb = r_lo; rb < r_hi; rb += MLP_EB) {
        __syncthreads();
#if !(PF_DW_ABL & 4)
#pragma unroll
        for (int n = 0; n < UN; ++n)
            if (colA && elA0 + n * dEA < MLP_EB) *reinterpret_cast<f4*>(ldsA + n * dEA * lda) = ra[n];
#pragma unroll
        for (int n = 0; n < UN; ++n)
            if (elB0 + n * dEB < MLP_EB) *reinterpret_cast<f4*>(ldsB + n * dEB * ldb) = rbv[n];
        if (ycol) Bs[yel * ldb + p.cc + yw] = yv;
#endif
        __syncthreads();
#if !(PF_DW_ABL & 2)
        if (rb + MLP_EB < r_hi) fetch(rb + MLP_EB);
#endif
        for (int e = 0; e < brows; ++e) bsum += As[(bgrp * brows + e) * lda + bcol];
        // branch-free per tile count: every operand read of a k step is issued before its MFMAs (with a uniform `if (val[s])`
        // around each read + MFMA pair hipcc waited for every LDS read in front of its MFMA: ~200 cycles per MFMA)
        const float* ar0 = As + q * lda + row;
        const float* br0 = Bs + q * ldb + row;
#if !(PF_DW_ABL & 1)
        switch (same ? nper_w : -nper_w) {
#define PF_DWB(NP)                                                                                     \
            case NP: dw_block<NP, true>(ar0, br0, rts, cts, acc); break;                     \
            case -NP: dw_block<NP, false>(ar0, br0, rts, cts, acc); break;
            PF_DWB(1) PF_DWB(2) PF_DWB(3) PF_DWB(4) PF_DWB(5) PF_DWB(6) PF_DWB(7) PF_DWB(8) PF_DWB(9)
#undef PF_DWB
            default: break;
        }
#endif
    }
    // the row groups' bias sums through LDS, added in group order (the same for every chunk: deterministic)
    __syncthreads();
    As[bgrp * lda + bcol] = bsum;
    __syncthreads();
    bsum = 0.f;
    if ((int)threadIdx.x < RA)
        for (int gI = 0; gI < NTH / RA; ++gI) bsum += As[gI * lda + threadIdx.x];
    gp out = part + (size_t)blockIdx.x * L.total;
    if (threadIdx.x < RA && !cpart) out[boffl + threadIdx.x] = bsum;
#pragma unroll
    for (int s = 0; s < MLP_SLOTS; ++s)
        if (val[s])
#pragma unroll
            for (int r = 0; r < 4; ++r) out[offl + (rts[s] * 16 + 4 * q + r) * RB + cts[s] * 16 + row] = acc[s][r];
}

// partial sums -> dW[l] [wo, in_l] (column j < td of layer 0 sits behind the cc conditioning columns in the partials), db[l]
constexpr int MLP_RG = PF_MLP_RG;           // groups of 64 threads that share the chunk range of an output element
__global__ __launch_bounds__(64 * MLP_RG) void mlp_dw_reduce_kernel(MlpBatch bb) {
    const PfMlpTrain p = mlp_desc(bb);
    const MlpG G = mlp_glob(p);
    gcp part = G.ws;
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
    gp db = sel3(G.db, l);
    if (j == inl) { if (db) db[c] = (float)s; }
    else sel3(G.dW, l)[(size_t)c * inl + j] = (float)s;
}

static inline MlpBatch mlp_one(const PfMlpTrain& p) { MlpBatch b; b.p[0] = p; return b; }

template <typename KERNEL>
void allow_lds(KERNEL k, size_t bytes) {
    pf_allow_lds(reinterpret_cast<const void*>(k), bytes);
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
    if (p->nl == 2) { allow_lds(mlp_fwd_kernel<2>, lds); hipLaunchKernelGGL(mlp_fwd_kernel<2>, dim3(grid), dim3(256), lds, s, mlp_one(*p)); }
    else { allow_lds(mlp_fwd_kernel<3>, lds); hipLaunchKernelGGL(mlp_fwd_kernel<3>, dim3(grid), dim3(256), lds, s, mlp_one(*p)); }
    return pf_last_launch_status();
}

extern "C" int pf_mlp_train_bwd(const PfMlpTrain* p, void* stream) {
    int st = mlp_check(p);
    if (st) return st;
    if (p->flags) return PF_ERR_UNSUPPORTED;             // PF_MLP_DW_DZSUM reads dc: only where dc is not an output (pf_mlp_train_dw_batch)
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
        if (p->nl == 2) { allow_lds(mlp_bwd_kernel<2>, lds); hipLaunchKernelGGL(mlp_bwd_kernel<2>, dim3(grid), dim3(256), lds, s, mlp_one(*p)); }
        else { allow_lds(mlp_bwd_kernel<3>, lds); hipLaunchKernelGGL(mlp_bwd_kernel<3>, dim3(grid), dim3(256), lds, s, mlp_one(*p)); }
    }
    const int chunk = mlp_chunk(*p);
    const int nchunk = (p->rows + chunk - 1) / chunk;
    s = pf_dw_fork(s);                                   // weight gradients: on their own stream when one is set (pf_train_set_dw_stream)
    {
        int ramax = 0, rbmax = 0;
        for (int l = 0; l < p->nl; ++l) { ramax = ramax > sh.wo16[l] ? ramax : sh.wo16[l]; rbmax = rbmax > L.wb16[l] ? rbmax : L.wb16[l]; }
        const size_t lds = sizeof(float) * (size_t)MLP_EB * 2 * MLP_LD;
        allow_lds(mlp_dw_kernel, lds);
        hipLaunchKernelGGL(mlp_dw_kernel, dim3(nchunk, p->nl), dim3(64 * MLP_DW_WAVES), lds, s, mlp_one(*p));
    }
    int total = 0;
    for (int l = 0; l < p->nl; ++l) total += sh.wo[l] * (sh.in[l] + 1);
    hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3((total + 63) / 64), dim3(64 * MLP_RG), 0, s, mlp_one(*p));
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
    (void)dev_descs;                                      // (kept in the ABI: the descriptors now travel as the kernels' own argument)
    if (descs[0].nl == 2) { allow_lds(mlp_fwd_kernel<2>, lds); hipLaunchKernelGGL(mlp_fwd_kernel<2>, dim3(gmax, 1, n), dim3(256), lds, s, b); }
    else { allow_lds(mlp_fwd_kernel<3>, lds); hipLaunchKernelGGL(mlp_fwd_kernel<3>, dim3(gmax, 1, n), dim3(256), lds, s, b); }
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
        if (p->flags) return PF_ERR_UNSUPPORTED;
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
        const size_t w2 = sizeof(float) * (size_t)MLP_EB * 2 * MLP_LD;
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
    (void)dev_descs;                                      // (kept in the ABI: the descriptors now travel as the kernels' own argument)
    if (descs[0].nl == 2) { allow_lds(mlp_bwd_kernel<2>, lds_b); hipLaunchKernelGGL(mlp_bwd_kernel<2>, dim3(gmax, 1, n), dim3(256), lds_b, s, b); }
    else { allow_lds(mlp_bwd_kernel<3>, lds_b); hipLaunchKernelGGL(mlp_bwd_kernel<3>, dim3(gmax, 1, n), dim3(256), lds_b, s, b); }
    s = pf_dw_fork(s);                                   // (dev_descs must then be this call's own: the next upload is not ordered behind it)
    allow_lds(mlp_dw_kernel, lds_w);
    hipLaunchKernelGGL(mlp_dw_kernel, dim3(cmax, descs[0].nl, n), dim3(64 * MLP_DW_WAVES), lds_w, s, b);
    hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3((tmax + 63) / 64, 1, n), dim3(64 * MLP_RG), 0, s, b);
    return pf_last_launch_status();
}

// weight gradients only: dz / h / dout of every network are already in memory (csrc/train_flowchain.hip writes them from its
// own chain kernel); split-K launch + reduction over all n networks
extern "C" int pf_mlp_train_dw_batch(const PfMlpTrain* descs, int n, void* dev_descs, void* stream) {
    if (!descs || !dev_descs) return PF_ERR_NULL;
    if (n < 1 || n > MLP_BATCH_MAX) return PF_ERR_SHAPE;
    size_t lds_w = 0;
    int cmax = 1, tmax = 1, extra = 0;
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
        if (p->flags & ~PF_MLP_DW_DZSUM) return PF_ERR_UNSUPPORTED;
        if ((p->flags & PF_MLP_DW_DZSUM) && p->cdiv > 1) {               // dc = the replica-summed dz[0] (see the header)
            if (!p->dc) return PF_ERR_NULL;
            if (p->cc % 16 != 0 || mlp_chunk(*p) % p->cdiv != 0) return PF_ERR_SHAPE;
            extra = 1;
        }
        const MlpShape sh = mlp_shape(*p);
        const MlpDwLayout L = mlp_dw_layout(*p, sh);
        int ramax = 0, rbmax = 0, total = 0;
        for (int l = 0; l < p->nl; ++l) {
            ramax = ramax > sh.wo16[l] ? ramax : sh.wo16[l]; rbmax = rbmax > L.wb16[l] ? rbmax : L.wb16[l];
            total += sh.wo[l] * (sh.in[l] + 1);
        }
        const size_t w2 = sizeof(float) * (size_t)MLP_EB * 2 * MLP_LD;
        lds_w = w2 > lds_w ? w2 : lds_w;
        const int nchunk = (p->rows + mlp_chunk(*p) - 1) / mlp_chunk(*p);
        cmax = nchunk > cmax ? nchunk : cmax;
        tmax = total > tmax ? total : tmax;
        b.p[k] = *p;
    }
    hipStream_t s = pf_dw_fork((hipStream_t)stream);     // all of it is weight-gradient work
    (void)dev_descs;                                      // (kept in the ABI: the descriptors now travel as the kernels' own argument)
    allow_lds(mlp_dw_kernel, lds_w);
    hipLaunchKernelGGL(mlp_dw_kernel, dim3(cmax, descs[0].nl + extra, n), dim3(64 * MLP_DW_WAVES), lds_w, s, b);
    hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3((tmax + 63) / 64, 1, n), dim3(64 * MLP_RG), 0, s, b);
    return pf_last_launch_status();
}
