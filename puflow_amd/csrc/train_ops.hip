// Building blocks of the TRAINING step (train-mode forward with BatchNorm batch statistics, and the
// backward of every layer).  The reference trains with eager PyTorch ops on materialised
// [B,C,N,K] tensors (modules/discrete/interpflow.py:203-258, train_pu1k.py:53-74); here each of those
// ops is one hand-written kernel pair on channels-last [rows, C] fp32 tensors, wired into autograd
// by puflow_amd/train_ops.py.  Training runs on 256-point patches (32 x 256 points, 131 072 edges),
// so this first version is deliberately un-fused; the fused MFMA-chain kernels serve inference.
//
//   gemm            C = op(A) op(B) (+bias)    every Conv2d(1x1) / Linear forward, dX and dW (split-K slabs,
//                                              deterministic reduce)            f32 MFMA 16x16x4, 64x64 tiles
//   colstat / bn    BatchNorm2d(train) + LeakyReLU forward / backward (two-pass mean / variance)
//   act             LeakyReLU / ReLU forward / backward
//   edge_feature    [x_i, x_j, x_j - x_i] gather forward, scatter-add backward   (interpflow.py:223-232)
//   maxpool_k       max over the K neighbours + argmax, backward                  (interpflow.py:245)
//   scatter_rows    backward of a row gather (z[idx], interpflow.py:183)
//   softmax_wsum    softmax over K of R logit channels, weighted sum of neighbour latents, backward (interpflow.py:180-185)
//   group_sum       backward of repeat_interleave(cs, R) (interpflow.py:319)
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"
#include "pf_mfma.h"

namespace {

// ------------------------------------------------------------------------------------------ GEMM
struct GemmArgs {
    const float* A; long long sam, sak;      // A(m,k) = A[m*sam + k*sak]
    const float* B; long long sbk, sbn;      // B(k,n) = B[k*sbk + n*sbn]
    float* C; long long ldc;                 // C(m,n) = C[m*ldc + n]   (or slab z: C + z*M*ldc)
    const float* bias;                       // per column n, nullable (ignored when splitk > 1: added by the reduce)
    int M, N, K, kchunk;                     // kchunk = K range per blockIdx.z
    const float* add;                        // nullable: C(m,n) += add[m*ldc + n] (same layout as C; may BE C; ignored when splitk > 1:
};                                           //           added by the reduce) - a gradient that already holds another consumer's part

// Output tile BM x BN per 256-thread workgroup: WAVES_M x WAVES_N waves, each TM x TN MFMA tiles of 16x16; K-step 16.
// The launcher picks the shape from (M, N): layer GEMMs here are very skinny (N = 8..128 forward, M = 8..128 for dW),
// so a square tile would waste most of its MFMAs.
// VEC: both operands are read with float4 loads along their contiguous dimension (requires that dimension's
// extent and leading stride to be multiples of 4 and 16-byte aligned bases); otherwise scalar guarded loads.
template <int WAVES_M, int WAVES_N, int TM, int TN, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    constexpr int BM = WAVES_M * TM * 16, BN = WAVES_N * TN * 16;
    __shared__ float As[BM][20];          // [m][k], row stride 20 floats: 16-B aligned rows
    __shared__ float Bs[16][BN + 4];      // [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_lo = blockIdx.z * g.kchunk;
    const int k_hi = min(g.K, k_lo + g.kchunk);
    const bool a_kfast = g.sak == 1, b_nfast = g.sbn == 1;
    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = pf_splat(0.f);

    // VEC: the next K-step's operand tiles are fetched into registers while the current one is multiplied (the layer GEMMs
    // are 2..32 K-steps long: with load -> barrier -> multiply -> barrier per step every step paid a full memory latency)
    constexpr int NA = (BM * 4 + 255) / 256, NB = (BN * 4 + 255) / 256;
    f4 ra[NA], rb[NB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int v = tid + i * 256;
            f4 x = pf_splat(0.f);
            if (v < BM * 4) {
                if (a_kfast) {
                    const int r = v >> 2, k = (v & 3) * 4;
                    const int gm = m0 + r, gk = k0 + k;
                    if (gm < g.M && gk < k_hi) x = *reinterpret_cast<const f4*>(g.A + gm * g.sam + gk);
                } else {
                    const int k = v / (BM / 4), r = (v % (BM / 4)) * 4;
                    const int gm = m0 + r, gk = k0 + k;
                    if (gm < g.M && gk < k_hi) x = *reinterpret_cast<const f4*>(g.A + gk * g.sak + gm);
                }
            }
            ra[i] = x;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int v = tid + i * 256;
            f4 x = pf_splat(0.f);
            if (v < BN * 4) {
                if (b_nfast) {
                    const int k = v / (BN / 4), n = (v % (BN / 4)) * 4;
                    const int gn = n0 + n, gk = k0 + k;
                    if (gn < g.N && gk < k_hi) x = *reinterpret_cast<const f4*>(g.B + gk * g.sbk + gn);
                } else {
                    const int n = v >> 2, k = (v & 3) * 4;
                    const int gn = n0 + n, gk = k0 + k;
                    if (gn < g.N && gk < k_hi) x = *reinterpret_cast<const f4*>(g.B + gn * g.sbn + gk);
                }
            }
            rb[i] = x;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int v = tid + i * 256;
            if (v < BM * 4) {
                const f4 x = ra[i];
                if (a_kfast) {
                    const int r = v >> 2, k = (v & 3) * 4;
                    *reinterpret_cast<f4*>(&As[r][k]) = x;
                } else {
                    const int k = v / (BM / 4), r = (v % (BM / 4)) * 4;
                    As[r][k] = x.x; As[r + 1][k] = x.y; As[r + 2][k] = x.z; As[r + 3][k] = x.w;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int v = tid + i * 256;
            if (v < BN * 4) {
                const f4 x = rb[i];
                if (b_nfast) {
                    const int k = v / (BN / 4), n = (v % (BN / 4)) * 4;
                    *reinterpret_cast<f4*>(&Bs[k][n]) = x;
                } else {
                    const int n = v >> 2, k = (v & 3) * 4;
                    Bs[k][n] = x.x; Bs[k + 1][n] = x.y; Bs[k + 2][n] = x.z; Bs[k + 3][n] = x.w;
                }
            }
        }
    };
    if (VEC) fetch(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += 16) {
        if (VEC) {
            stash();
        } else {
            for (int v = tid; v < BM * 16; v += 256) {
                const int r = a_kfast ? (v >> 4) : (v % BM);
                const int k = a_kfast ? (v & 15) : (v / BM);
                const int gm = m0 + r, gk = k0 + k;
                As[r][k] = (gm < g.M && gk < k_hi) ? g.A[gm * g.sam + gk * g.sak] : 0.f;
            }
            for (int v = tid; v < BN * 16; v += 256) {
                const int n = b_nfast ? (v % BN) : (v >> 4);
                const int kb = b_nfast ? (v / BN) : (v & 15);
                const int gn = n0 + n, gkb = k0 + kb;
                Bs[kb][n] = (gn < g.N && gkb < k_hi) ? g.B[gkb * g.sbk + gn * g.sbn] : 0.f;
            }
        }
        __syncthreads();
        if (VEC && k0 + 16 < k_hi) fetch(k0 + 16);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(wm * TM + i) * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[kk * 4 + (lane >> 4)][(wn * TN + j) * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = pf_mfma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* C = g.C + (long long)blockIdx.z * g.M * g.ldc;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (wm * TM + i) * 16 + 4 * (lane >> 4) + r;
                if (m < g.M && n < g.N) C[m * g.ldc + n] = acc[i][j][r] + (g.bias ? g.bias[n] : 0.f) + (g.add ? g.add[m * g.ldc + n] : 0.f);
            }
        }
}

// ---- round 5: the same GEMM (same MFMA chain per output element, bit-identical results) with conflict-free LDS images ----
// gemm_kernel above stages through As[BM][20] / Bs[16][BN + 4]: an operand whose contiguous dimension is NOT k is transposed
// with scalar ds_write_b32 at a 20-float stride (8-way bank conflicts), the b32 fragment reads are 2-way (PMC: 54 - 86 % of
// the LDS cycles were conflict cycles), every 16-deep step pays two barriers and the epilogue stores 64-byte pieces.  Here:
//   * an operand that is contiguous along k goes to LDS in FRAGMENT order [16-k group][tile][lane = q 16 + row][j]
//     (element j of lane (row, q) is k = 4 j + q: what MFMA step j of the group reads), written by four ds_write_b32 whose
//     32-lane groups cover 32 banks, read back as ONE ds_read_b128 per tile and group;
//   * an operand that is contiguous along its row index (m / n) keeps its memory order [k][rows + 16]: ds_write_b128 rows,
//     ds_read_b32 fragments whose two k rows of a 32-lane group sit 16 banks apart;
//   * 32-deep steps, two LDS buffers, ONE barrier per step, the next step's operands in registers during the MFMAs;
//   * the output tile leaves through LDS as whole rows (float4 per lane).
// The products and their order are those of gemm_kernel (k ascending through v_mfma_f32_16x16x4_f32, the same split-K chunks).
template <int WAVES_M, int WAVES_N, int TM, int TN>
__global__ __launch_bounds__(256) void gemm2_kernel(GemmArgs g, int cvec) {
    constexpr int BM = WAVES_M * TM * 16, BN = WAVES_N * TN * 16, BK = 32;
    constexpr int LDA = BM % 32 == 0 ? BM + 16 : BM + 32, LDB = BN % 32 == 0 ? BN + 16 : BN + 32;   // row stride = 16 mod 32 floats
    constexpr int ASZ = BK * LDA, BSZ = BK * LDB, LDC = BN + 4;
    extern __shared__ __attribute__((aligned(16))) float g2lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ml = lane & 15, q = lane >> 4;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_lo = blockIdx.z * g.kchunk;
    const int k_hi = min(g.K, k_lo + g.kchunk);
    const bool a_kfast = g.sak == 1, b_kfast = g.sbk == 1 && g.sbn != 1;
    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = pf_splat(0.f);
    constexpr int NA = (BM * 8 + 255) / 256, NB = (BN * 8 + 255) / 256;
    f4 ra[NA], rb[NB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int v = tid + i * 256;
            f4 x = pf_splat(0.f);
            if (v < BM * 8) {
                if (a_kfast) {
                    const int gm = m0 + (v >> 3), gk = k0 + (v & 7) * 4;
                    if (gm < g.M && gk < k_hi) x = *reinterpret_cast<const f4*>(g.A + gm * g.sam + gk);
                } else {
                    const int gk = k0 + v / (BM / 4), gm = m0 + (v % (BM / 4)) * 4;
                    if (gm < g.M && gk < k_hi) x = *reinterpret_cast<const f4*>(g.A + gk * g.sak + gm);
                }
            }
            ra[i] = x;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int v = tid + i * 256;
            f4 x = pf_splat(0.f);
            if (v < BN * 8) {
                if (b_kfast) {
                    const int gn = n0 + (v >> 3), gk = k0 + (v & 7) * 4;
                    if (gn < g.N && gk < k_hi) x = *reinterpret_cast<const f4*>(g.B + gn * g.sbn + gk);
                } else {
                    const int gk = k0 + v / (BN / 4), gn = n0 + (v % (BN / 4)) * 4;
                    if (gn < g.N && gk < k_hi) x = *reinterpret_cast<const f4*>(g.B + gk * g.sbk + gn);
                }
            }
            rb[i] = x;
        }
    };
    auto stash = [&](float* As, float* Bs) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int v = tid + i * 256;
            if (v < BM * 8) {
                const f4 x = ra[i];
                if (a_kfast) {
                    const int r = v >> 3, c = v & 7;
                    float* p = As + (c >> 2) * (BM * 16) + ((r >> 4) * 64 + (r & 15)) * 4 + (c & 3);
                    p[0] = x.x; p[64] = x.y; p[128] = x.z; p[192] = x.w;
                } else {
                    *reinterpret_cast<f4*>(As + (v / (BM / 4)) * LDA + (v % (BM / 4)) * 4) = x;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int v = tid + i * 256;
            if (v < BN * 8) {
                const f4 x = rb[i];
                if (b_kfast) {
                    const int r = v >> 3, c = v & 7;
                    float* p = Bs + (c >> 2) * (BN * 16) + ((r >> 4) * 64 + (r & 15)) * 4 + (c & 3);
                    p[0] = x.x; p[64] = x.y; p[128] = x.z; p[192] = x.w;
                } else {
                    *reinterpret_cast<f4*>(Bs + (v / (BN / 4)) * LDB + (v % (BN / 4)) * 4) = x;
                }
            }
        }
    };
    fetch(k_lo);
    stash(g2lds, g2lds + ASZ);
    __syncthreads();
    int cur = 0;
    for (int k0 = k_lo; k0 < k_hi; k0 += BK) {
        const float* As = g2lds + cur * (ASZ + BSZ);
        const float* Bs = As + ASZ;
        const bool more = k0 + BK < k_hi;
        if (more) fetch(k0 + BK);
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            f4 a[TM], b[TN];
            if (a_kfast) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f4*>(As + gq * (BM * 16) + ((wm * TM + i) * 64 + lane) * 4);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) a[i][kk] = As[(gq * 16 + kk * 4 + q) * LDA + (wm * TM + i) * 16 + ml];
            }
            if (b_kfast) {
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f4*>(Bs + gq * (BN * 16) + ((wn * TN + j) * 64 + lane) * 4);
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) b[j][kk] = Bs[(gq * 16 + kk * 4 + q) * LDB + (wn * TN + j) * 16 + ml];
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = pf_mfma(a[i][kk], b[j][kk], acc[i][j]);
        }
        if (more) {
            float* An = g2lds + (cur ^ 1) * (ASZ + BSZ);
            stash(An, An + ASZ);
        }
        __syncthreads();
        cur ^= 1;
    }
    // ---- the tile through LDS: [BM][BN + 4] (a lane group's two 4-row blocks sit 16 banks apart), then whole rows out
    float* Cs = g2lds;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[((wm * TM + i) * 16 + 4 * q + r) * LDC + (wn * TN + j) * 16 + ml] = acc[i][j][r];
    __syncthreads();
    float* C = g.C + (long long)blockIdx.z * g.M * g.ldc;
    for (int v = tid; v < BM * (BN / 4); v += 256) {
        const int row = v / (BN / 4), c4 = (v % (BN / 4)) * 4;
        const int m = m0 + row, n = n0 + c4;
        if (m >= g.M || n >= g.N) continue;
        f4 x = *reinterpret_cast<const f4*>(Cs + row * LDC + c4);
        if (cvec && n + 3 < g.N) {
            if (g.bias) { const f4 bb = *reinterpret_cast<const f4*>(g.bias + n); x += bb; }
            if (g.add) { const f4 aa = *reinterpret_cast<const f4*>(g.add + m * g.ldc + n); x += aa; }
            *reinterpret_cast<f4*>(C + m * g.ldc + n) = x;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < g.N) C[m * g.ldc + n + e] = x[e] + (g.bias ? g.bias[n + e] : 0.f) + (g.add ? g.add[m * g.ldc + n + e] : 0.f);
        }
    }
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
void gemm2_launch(const GemmArgs& g, int split, hipStream_t s) {
    constexpr int BM = WAVES_M * TM * 16, BN = WAVES_N * TN * 16;
    constexpr int LDA = BM % 32 == 0 ? BM + 16 : BM + 32, LDB = BN % 32 == 0 ? BN + 16 : BN + 32;
    constexpr int loop_floats = 2 * 32 * (LDA + LDB), out_floats = BM * (BN + 4);
    constexpr size_t lds = sizeof(float) * (size_t)(loop_floats > out_floats ? loop_floats : out_floats);
    pf_allow_lds(reinterpret_cast<const void*>(gemm2_kernel<WAVES_M, WAVES_N, TM, TN>), lds);
    auto al16 = [](const void* p) { return (reinterpret_cast<unsigned long long>(p) & 15ull) == 0; };
    const long long ldc = g.ldc;
    const int cvec = (ldc % 4 == 0) && al16(g.C) && (!g.bias || al16(g.bias)) && (!g.add || al16(g.add)) && (((long long)g.M * ldc) % 4 == 0);
    const dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, split);
    hipLaunchKernelGGL((gemm2_kernel<WAVES_M, WAVES_N, TM, TN>), grid, dim3(256), lds, s, g, cvec);
}

// ---- split-precision GEMM: the same tiles on the fp16 / bf16 matrix pipe ------------------------------------------
// The f32 MFMA runs at 1/16 of the 16-bit rate; the big layer GEMMs above already sit near ITS roofline.  Here both
// operands are split while they are staged into LDS ([row][k] images, k contiguous: one 16-B read per MFMA operand):
//   NS = 2  x = hi + lo in fp16 (natural-scale low half, pf_mfma.h "f16n"): 3 MFMAs per 32-deep step - forward GEMMs
//           (activations and weights are O(1e-3 .. 1e2): inside the fp16 range)
//   NS = 3  x = hi + mid + lo in bf16: 6 MFMAs per step, fp32 exponent range - the GEMMs that take a gradient operand
//           (dX = dY W, dW = dY^T X: gradients reach 1e-8 and would underflow an fp16 split)
// Results are fp32-class (>= 22 significant bits per product, fp32 accumulation); 5.3x / 2.7x fewer MFMA cycles than f32.
template <int NS> struct SplitT;
template <> struct SplitT<2> { typedef _Float16 T; };
template <> struct SplitT<3> { typedef __bf16 T; };

template <int NS, int WAVES_M, int WAVES_N, int TM, int TN, bool VEC>
__global__ __launch_bounds__(256) void gemm_split_kernel(GemmArgs g) {
    typedef typename SplitT<NS>::T T;
    typedef T T8 __attribute__((ext_vector_type(8)));
    constexpr int BM = WAVES_M * TM * 16, BN = WAVES_N * TN * 16, BK = 32, LDK = BK + 8;   // row stride 80 B: 16-B aligned
    __shared__ T As[NS][BM][LDK];
    __shared__ T Bs[NS][BN][LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_lo = blockIdx.z * g.kchunk;
    const int k_hi = min(g.K, k_lo + g.kchunk);
    const bool a_kfast = g.sak == 1, b_kfast = g.sbk == 1;
    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = pf_splat(0.f);
    // one element -> its NS parts
    auto split_store = [&](T* p0, long long part_stride, float x) {
        if constexpr (NS == 2) {
            const _Float16 h = (_Float16)x;
            p0[0] = h;
            p0[part_stride] = (_Float16)(x - (float)h);
        } else {
            const __bf16 h = (__bf16)x;
            const float r1 = x - (float)h;
            const __bf16 m = (__bf16)r1;
            p0[0] = h;
            p0[part_stride] = m;
            p0[2 * part_stride] = (__bf16)(r1 - (float)m);
        }
    };
    constexpr long long PSA = (long long)BM * LDK, PSB = (long long)BN * LDK;

    for (int k0 = k_lo; k0 < k_hi; k0 += BK) {
        // ---- stage A [BM][BK] and B^T [BN][BK], converting on the way
        for (int v = tid; v < BM * (BK / 4); v += 256) {
            int r, k;
            f4 x = pf_splat(0.f);
            if (a_kfast) {                                      // 4 consecutive k of one row
                r = v / (BK / 4); k = (v % (BK / 4)) * 4;
                const int gm = m0 + r, gk = k0 + k;
                if (gm < g.M) {
                    if (VEC && gk + 3 < k_hi) x = *reinterpret_cast<const f4*>(g.A + gm * g.sam + gk);
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (gk + e < k_hi) x[e] = g.A[gm * g.sam + (gk + e) * g.sak];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) split_store(&As[0][r][k + e], PSA, x[e]);
            } else {                                            // 4 consecutive rows of one k
                k = v / (BM / 4); r = (v % (BM / 4)) * 4;
                const int gm = m0 + r, gk = k0 + k;
                if (gk < k_hi) {
                    if (VEC && gm + 3 < g.M) x = *reinterpret_cast<const f4*>(g.A + gk * g.sak + gm * g.sam);
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (gm + e < g.M) x[e] = g.A[(gm + e) * g.sam + gk * g.sak];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) split_store(&As[0][r + e][k], PSA, x[e]);
            }
        }
        for (int v = tid; v < BN * (BK / 4); v += 256) {
            int n, k;
            f4 x = pf_splat(0.f);
            if (b_kfast) {                                      // 4 consecutive k of one column n
                n = v / (BK / 4); k = (v % (BK / 4)) * 4;
                const int gn = n0 + n, gk = k0 + k;
                if (gn < g.N) {
                    if (VEC && gk + 3 < k_hi) x = *reinterpret_cast<const f4*>(g.B + gn * g.sbn + gk);
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (gk + e < k_hi) x[e] = g.B[(gk + e) * g.sbk + gn * g.sbn];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) split_store(&Bs[0][n][k + e], PSB, x[e]);
            } else {                                            // 4 consecutive n of one k
                k = v / (BN / 4); n = (v % (BN / 4)) * 4;
                const int gn = n0 + n, gk = k0 + k;
                if (gk < k_hi) {
                    if (VEC && gn + 3 < g.N) x = *reinterpret_cast<const f4*>(g.B + gk * g.sbk + gn * g.sbn);
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (gn + e < g.N) x[e] = g.B[gk * g.sbk + (gn + e) * g.sbn];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) split_store(&Bs[0][n + e][k], PSB, x[e]);
            }
        }
        __syncthreads();
        T8 a[NS][TM], b[NS][TN];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[s][i] = *reinterpret_cast<const T8*>(&As[s][(wm * TM + i) * 16 + (lane & 15)][8 * (lane >> 4)]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[s][j] = *reinterpret_cast<const T8*>(&Bs[s][(wn * TN + j) * 16 + (lane & 15)][8 * (lane >> 4)]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f4 x = acc[i][j];
                if constexpr (NS == 2) {
                    x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][i], b[1][j], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1][i], b[0][j], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][i], b[0][j], x, 0, 0, 0);
                } else {
                    x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[2][j], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2][i], b[0][j], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][i], b[1][j], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[1][j], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][i], b[0][j], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[0][j], x, 0, 0, 0);
                }
                acc[i][j] = x;
            }
        __syncthreads();
    }
    float* C = g.C + (long long)blockIdx.z * g.M * g.ldc;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (wm * TM + i) * 16 + 4 * (lane >> 4) + r;
                if (m < g.M && n < g.N) C[m * g.ldc + n] = acc[i][j][r] + (g.bias ? g.bias[n] : 0.f) + (g.add ? g.add[m * g.ldc + n] : 0.f);
            }
        }
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
void gemm_split_launch(int ns, const GemmArgs& g, int split, bool vec, hipStream_t s) {
    constexpr int BM = WAVES_M * TM * 16, BN = WAVES_N * TN * 16;
    const dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, split);
    if (ns == 2) {
        if (vec) hipLaunchKernelGGL((gemm_split_kernel<2, WAVES_M, WAVES_N, TM, TN, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_split_kernel<2, WAVES_M, WAVES_N, TM, TN, false>), grid, dim3(256), 0, s, g);
    } else {
        if (vec) hipLaunchKernelGGL((gemm_split_kernel<3, WAVES_M, WAVES_N, TM, TN, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_split_kernel<3, WAVES_M, WAVES_N, TM, TN, false>), grid, dim3(256), 0, s, g);
    }
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
void gemm_launch(const GemmArgs& g, int split, bool vec, hipStream_t s) {
    constexpr int BM = WAVES_M * TM * 16, BN = WAVES_N * TN * 16;
    const dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, split);
    if (vec) hipLaunchKernelGGL((gemm_kernel<WAVES_M, WAVES_N, TM, TN, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_kernel<WAVES_M, WAVES_N, TM, TN, false>), grid, dim3(256), 0, s, g);
}

#ifndef PF_GEMM_T9
#define PF_GEMM_T9 256
#endif
#ifndef PF_GEMM_SMALL_TILES
#define PF_GEMM_SMALL_TILES 1
#endif
// 64 x 64 tiles (shape 7), or - where those leave the chip with one 4-wave workgroup per CU or less (the [8192, 32..128] input
// gradients of the EdgeConv units: 128 / 256 tiles, every k-step's loads exposed) - 32 x 32 (9; 32 x 64 = 8 for N <= 32).  Same box,
// [8192, C] x K = 4 S: C = 32: 14.4 -> 10.6 us, 64: 24.9 -> 15.7, 128: 25.1 -> 21.9; training step 4.13 -> 4.08 ms
inline int gemm_small_tile(int M, int N) {
#if PF_GEMM_SMALL_TILES
    const long long t64 = (long long)((M + 63) / 64) * ((N + 63) / 64);
    if (t64 <= PF_GEMM_T9 && M >= 32 && N > 32) return 9;
    if (t64 <= 256 && M >= 32) return 8;
#endif
    return 7;
}
// tile shape for (M, N): 0 = 128x128, 1..3 = 256 x {16,32,64} (skinny N), 4..6 = {16,32,64} x 256 (skinny M)
inline int gemm_shape(int M, int N) {
    // a skinny output whose 256-row tiles would not even give every second CU a workgroup (the [8192, 16..64] input-gradient
    // GEMMs of the training step: 32 tiles, 65 us for 0.27 G MAC) takes 64 x 64 tiles instead: 4 x the workgroups, no split-K
    // reduction, the wasted tile columns cost nothing at this size
    if (N <= 64 && M > 64 && (M + 255) / 256 < 128 && (M + 63) / 64 >= 64) return gemm_small_tile(M, N);
    if (N <= 16) return 1;
    if (N <= 32) return 2;
    if (N <= 64 && M > 64) return 3;
    if (M <= 16) return 4;
    if (M <= 32) return 5;
    if (M <= 64) return 6;
    // 128 x 128 tiles leave most CUs idle on the point-level GEMMs of the training step ([8192, 128..512] outputs: 64..256
    // tiles, one 1-wave-per-SIMD workgroup per CU): 64 x 64 tiles there
    if ((long long)((M + 127) / 128) * ((N + 127) / 128) < 1024) return gemm_small_tile(M, N);
    return 0;
}
inline void gemm_tile_dims(int shape, int& bm, int& bn) {
    static const int d[10][2] = {{128, 128}, {256, 16}, {256, 32}, {256, 64}, {16, 256}, {32, 256}, {64, 256}, {64, 64}, {32, 64},
                                 {32, 32}};
    bm = d[shape][0]; bn = d[shape][1];
}

// C = sum over split-K slabs (+ bias): 64 consecutive outputs x 4 slab lanes per workgroup, fixed combine order
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float* __restrict__ slabs, float* C,
                                                         const float* __restrict__ bias, int M, int N, long long ldc,
                                                         int nslab, const float* add) {
    __shared__ float sh[4][64];
    const int l = threadIdx.x & 63, part = threadIdx.x >> 6;
    const long long t = (long long)blockIdx.x * 64 + l;
    const long long MN = (long long)M * N;
    float s = 0.f;
    if (t < MN)
        for (int z = part; z < nslab; z += 4) s += slabs[(long long)z * MN + t];
    sh[part][l] = s;
    __syncthreads();
    if (part == 0 && t < MN) {
        const int m = (int)(t / N), n = (int)(t % N);
        C[m * ldc + n] = ((sh[0][l] + sh[1][l]) + (sh[2][l] + sh[3][l])) + (bias ? bias[n] : 0.f) + (add ? add[m * ldc + n] : 0.f);
    }
}

// ----------------------------------------------------------------------------- column statistics
// partial[chunk][which][c] over rows [chunk*rows_per, ...): MODE 0: sum x          (1 value)
//                                                             MODE 1: sum (x - mean[c])^2     (1)
//                                                             MODE 2: sum dz, sum dz*xhat     (2)  dz = dy * lrelu'(pre)
template <int MODE>
__global__ __launch_bounds__(256) void colstat_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float slope, long long R, int C, int rows_per,
                                                     float* __restrict__ partial) {
    __shared__ float s0[4][64], s1[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const long long r_lo = (long long)blockIdx.y * rows_per, r_hi = min(R, r_lo + rows_per);
    float a0 = 0.f, a1 = 0.f;
    if (c < C) {
        const float mu = MODE >= 1 ? mean[c] : 0.f;
        const float is = MODE == 2 ? invstd[c] : 0.f, ga = MODE == 2 ? gamma[c] : 0.f, be = MODE == 2 ? beta[c] : 0.f;
        for (long long r = r_lo + rl; r < r_hi; r += 4) {
            const float v = x[r * C + c];
            if (MODE == 0) a0 += v;
            else if (MODE == 1) { const float d = v - mu; a0 += d * d; }
            else {
                const float xh = (v - mu) * is;
                const float dz = dy[r * C + c] * ((xh * ga + be) > 0.f ? 1.f : slope);
                a0 += dz; a1 += dz * xh;
            }
        }
    }
    s0[rl][threadIdx.x & 63] = a0; s1[rl][threadIdx.x & 63] = a1;
    __syncthreads();
    if (rl == 0 && c < C) {
        const int l = threadIdx.x;
        partial[((long long)blockIdx.y * 2 + 0) * C + c] = (s0[0][l] + s0[1][l]) + (s0[2][l] + s0[3][l]);
        if (MODE == 2) partial[((long long)blockIdx.y * 2 + 1) * C + c] = (s1[0][l] + s1[1][l]) + (s1[2][l] + s1[3][l]);
    }
}

// float4 variant for C % 4 == 0: a thread owns 4 consecutive channels, `cgl` (power of two <= 64) channel lanes cover
// one row segment and 256 / cgl row lanes walk the rows, so every wave-instruction reads whole contiguous rows.
template <int MODE>
__global__ __launch_bounds__(256) void colstat4_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float slope, long long R, int C, int rows_per, int cgl,
                                                      float* __restrict__ partial) {
    __shared__ f4 s0[256], s1[256];
    const int cl = threadIdx.x % cgl, rl = threadIdx.x / cgl, nrl = 256 / cgl;
    const int c = (blockIdx.x * cgl + cl) * 4;
    const long long r_lo = (long long)blockIdx.y * rows_per, r_hi = min(R, r_lo + rows_per);
    f4 a0 = pf_splat(0.f), a1 = pf_splat(0.f);
    if (c < C) {
        f4 mu = pf_splat(0.f), is = pf_splat(0.f), ga = pf_splat(0.f), be = pf_splat(0.f);
        if (MODE >= 1) mu = *reinterpret_cast<const f4*>(mean + c);
        if (MODE == 2) {
            is = *reinterpret_cast<const f4*>(invstd + c);
            ga = *reinterpret_cast<const f4*>(gamma + c);
            be = *reinterpret_cast<const f4*>(beta + c);
        }
        for (long long r = r_lo + rl; r < r_hi; r += nrl) {
            const f4 v = *reinterpret_cast<const f4*>(x + r * C + c);
            if (MODE == 0) a0 += v;
            else if (MODE == 1) { const f4 d = v - mu; a0 += d * d; }
            else {
                const f4 g = *reinterpret_cast<const f4*>(dy + r * C + c);
                const f4 xh = (v - mu) * is;
                const f4 pre = xh * ga + be;
                f4 dz;
#pragma unroll
                for (int i = 0; i < 4; ++i) dz[i] = g[i] * (pre[i] > 0.f ? 1.f : slope);
                a0 += dz; a1 += dz * xh;
            }
        }
    }
    s0[threadIdx.x] = a0; s1[threadIdx.x] = a1;
    __syncthreads();
    for (int st = nrl >> 1; st > 0; st >>= 1) {                    // fixed tree over the row lanes
        if (rl < st) { s0[threadIdx.x] += s0[threadIdx.x + st * cgl]; s1[threadIdx.x] += s1[threadIdx.x + st * cgl]; }
        __syncthreads();
    }
    if (rl == 0 && c < C) {
        *reinterpret_cast<f4*>(partial + ((long long)blockIdx.y * 2 + 0) * C + c) = s0[cl];
        if (MODE == 2) *reinterpret_cast<f4*>(partial + ((long long)blockIdx.y * 2 + 1) * C + c) = s1[cl];
    }
}

template <int MODE>
void colstat_launch(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma,
                    const float* beta, float slope, long long R, int C, int nchunk, int rows_per, float* partial, hipStream_t s) {
    if (C % 4 == 0) {
        int cgl = 1;
        while (cgl < C / 4 && cgl < 64) cgl <<= 1;
        hipLaunchKernelGGL(colstat4_kernel<MODE>, dim3((C / 4 + cgl - 1) / cgl, nchunk), dim3(256), 0, s, x, dy, mean, invstd, gamma,
                           beta, slope, R, C, rows_per, cgl, partial);
    } else {
        hipLaunchKernelGGL(colstat_kernel<MODE>, dim3((C + 63) / 64, nchunk), dim3(256), 0, s, x, dy, mean, invstd, gamma, beta,
                           slope, R, C, rows_per, partial);
    }
}

// out[which][c] = scale * sum_chunks partial   (fixed order: 4 interleaved chunk lanes, then a fixed tree)
__global__ __launch_bounds__(256) void colstat_final_kernel(const float* __restrict__ partial, int nchunk, int C, int nwhich,
                                                           float scale, float* __restrict__ out) {
    __shared__ float sh[4][64];
    const int l = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + l;
    for (int w = 0; w < nwhich; ++w) {
        float s = 0.f;
        if (c < C)
            for (int k = part; k < nchunk; k += 4) s += partial[((long long)k * 2 + w) * C + c];
        sh[part][l] = s;
        __syncthreads();
        if (part == 0 && c < C) out[w * C + c] = ((sh[0][l] + sh[1][l]) + (sh[2][l] + sh[3][l])) * scale;
        __syncthreads();
    }
}

// BatchNorm backward: the two column sums once, written unscaled to (dbeta, dgamma) and scaled by 1/R to sums[2][C] - one
// launch, no device-to-device copy nodes (a copy node that reads `sums` right before a kernel node that overwrites it is a
// write-after-read across a memcpy -> kernel edge of a captured graph; the per-op path is captured whenever it is taken)
__global__ __launch_bounds__(256) void colstat_final_bwd_kernel(const float* __restrict__ partial, int nchunk, int C, float inv_r,
                                                               float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                               float* __restrict__ sums) {
    __shared__ float sh[4][64];
    const int l = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + l;
    for (int w = 0; w < 2; ++w) {
        float s = 0.f;
        if (c < C)
            for (int k = part; k < nchunk; k += 4) s += partial[((long long)k * 2 + w) * C + c];
        sh[part][l] = s;
        __syncthreads();
        if (part == 0 && c < C) {
            const float t = (sh[0][l] + sh[1][l]) + (sh[2][l] + sh[3][l]);
            (w == 0 ? dbeta : dgamma)[c] = t * 1.0f;
            sums[w * C + c] = t * inv_r;
        }
        __syncthreads();
    }
}

// save[0..C) = mean (copied by the kernel: no copy node), save[C..2C) = 1/sqrt(var_b + eps); running statistics (nullable)
__global__ void bn_finish_kernel(const float* __restrict__ var_b, int C, float eps, float momentum, float unbias,
                                 const float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ run_mean,
                                 float* __restrict__ run_var, float* __restrict__ mean_out) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    invstd[c] = 1.0f / sqrtf(var_b[c] + eps);
    if (mean_out && mean_out != mean) mean_out[c] = mean[c];
    if (run_mean) {
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean[c];
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (var_b[c] * unbias);
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float slope, long long total, int C,
                                                      float* __restrict__ y) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float v = (x[i] - mean[c]) * invstd[c] * gamma[c] + beta[c];
        y[i] = v > 0.f ? v : v * slope;
    }
}

// dx = gamma*invstd*(dz - s1/R - xhat*s2/R)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ sums /*[2][C] already / R*/,
                                                          float slope, long long total, int C, float* __restrict__ dx) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float xh = (x[i] - mean[c]) * invstd[c];
        const float dz = dy[i] * ((xh * gamma[c] + beta[c]) > 0.f ? 1.f : slope);
        dx[i] = gamma[c] * invstd[c] * (dz - sums[c] - xh * sums[C + c]);
    }
}

// y = x > 0 ? x : slope*x  ;  dx = dy * (y > 0 ? 1 : slope)
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float slope, long long total,
                                                     float* __restrict__ y) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const float v = x[i];
        y[i] = v > 0.f ? v : v * slope;
    }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float slope,
                                                     long long total, float* __restrict__ dx) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
        dx[i] = dy[i] * (y[i] > 0.f ? 1.f : slope);
}

// ------------------------------------------------------------------------- edge features / pooling
// out[e][0:C] = x_i, [C:2C] = x_j, [2C:3C] = x_j - x_i ; e = t*K + k, j = (t / N)*N + idx[e]
__global__ __launch_bounds__(256) void edge_feature_fwd_kernel(const float* __restrict__ x, const int* __restrict__ idx,
                                                              int N, int K, int C, long long total /*E*C*/,
                                                              float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long e = i / C;
        const int c = (int)(i % C);
        const long long t = e / K;
        const long long j = (t / N) * N + idx[e];
        const float xi = x[t * C + c], xj = x[j * C + c];
        float* o = out + e * 3 * C;
        o[c] = xi; o[C + c] = xj; o[2 * C + c] = xj - xi;
    }
}
// dx[t] += g1 - g3 ; dx[j] += g2 + g3
__global__ __launch_bounds__(256) void edge_feature_bwd_kernel(const float* __restrict__ g, const int* __restrict__ idx,
                                                              int N, int K, int C, long long total,
                                                              float* __restrict__ dx) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long e = i / C;
        const int c = (int)(i % C);
        const long long t = e / K;
        const long long j = (t / N) * N + idx[e];
        const float* gg = g + e * 3 * C;
        atomicAdd(&dx[t * C + c], gg[c] - gg[2 * C + c]);
        atomicAdd(&dx[j * C + c], gg[C + c] + gg[2 * C + c]);
    }
}

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ y, int K, int C, long long total /*T*C*/,
                                                         float* __restrict__ out, int* __restrict__ arg) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long t = i / C;
        const int c = (int)(i % C);
        float best = y[(t * K) * C + c];
        int bi = 0;
        for (int k = 1; k < K; ++k) {
            const float v = y[(t * K + k) * C + c];
            if (v > best) { best = v; bi = k; }
        }
        out[i] = best; arg[i] = bi;
    }
}
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ arg, int K,
                                                         int C, long long total /*T*K*C*/, float* __restrict__ dx) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long e = i / C;
        const int c = (int)(i % C);
        const long long t = e / K;
        const int k = (int)(e % K);
        dx[i] = arg[t * C + c] == k ? dy[t * C + c] : 0.f;
    }
}

// out[row[e]] += g[e]   (rows given as batch-local idx: row = (t / N)*N + idx[e], t = e / K)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ g, const int* __restrict__ idx, int N,
                                                          int K, int C, long long total, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long e = i / C;
        const int c = (int)(i % C);
        const long long t = e / K;
        const long long j = (t / N) * N + idx[e];
        atomicAdd(&out[j * C + c], g[i]);
    }
}

// out[t][c] = sum_r g[t*R + r][c]
__global__ __launch_bounds__(256) void group_sum_kernel(const float* __restrict__ g, int R, int C, long long total /*T*C*/,
                                                       float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long t = i / C;
        const int c = (int)(i % C);
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += g[(t * R + r) * C + c];
        out[i] = s;
    }
}

// ---- softmax over K of R logit channels + weighted latent sum; one thread per point (K = 8, R <= 8)
constexpr int SW_K = 8, SW_RMAX = 8;
__global__ __launch_bounds__(256) void softmax_wsum_fwd_kernel(const float* __restrict__ w /*[T,K,ldw] first R used*/,
                                                              int ldw, const float* __restrict__ zj /*[T,K,3]*/, int R,
                                                              long long T, float* __restrict__ a /*[T,K,R]*/,
                                                              float* __restrict__ fz /*[T,3,R]*/) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    for (int r = 0; r < R; ++r) {
        float m = -__builtin_inff();
        for (int k = 0; k < SW_K; ++k) m = fmaxf(m, w[(t * SW_K + k) * ldw + r]);
        float e[SW_K], s = 0.f;
        for (int k = 0; k < SW_K; ++k) { e[k] = expf(w[(t * SW_K + k) * ldw + r] - m); s += e[k]; }
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
        for (int k = 0; k < SW_K; ++k) {
            const float ak = e[k] / s;
            a[(t * SW_K + k) * R + r] = ak;
            const float* z = zj + (t * SW_K + k) * 3;
            o0 += ak * z[0]; o1 += ak * z[1]; o2 += ak * z[2];
        }
        fz[(t * 3 + 0) * R + r] = o0; fz[(t * 3 + 1) * R + r] = o1; fz[(t * 3 + 2) * R + r] = o2;
    }
}
// dzj[t,k,c] = sum_r a dfz ; dw[t,k,r] = a (da - sum_k a da), da[t,k,r] = sum_c dfz[t,c,r] zj[t,k,c]; dw beyond R = 0
__global__ __launch_bounds__(256) void softmax_wsum_bwd_kernel(const float* __restrict__ a, const float* __restrict__ zj,
                                                              const float* __restrict__ dfz, int R, int ldw, long long T,
                                                              float* __restrict__ dw, float* __restrict__ dzj) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    float dz[SW_K][3];
    for (int k = 0; k < SW_K; ++k) dz[k][0] = dz[k][1] = dz[k][2] = 0.f;
    for (int r = 0; r < R; ++r) {
        const float g0 = dfz[(t * 3 + 0) * R + r], g1 = dfz[(t * 3 + 1) * R + r], g2 = dfz[(t * 3 + 2) * R + r];
        float da[SW_K], dot = 0.f;
        for (int k = 0; k < SW_K; ++k) {
            const float* z = zj + (t * SW_K + k) * 3;
            const float ak = a[(t * SW_K + k) * R + r];
            da[k] = g0 * z[0] + g1 * z[1] + g2 * z[2];
            dot += ak * da[k];
            dz[k][0] += ak * g0; dz[k][1] += ak * g1; dz[k][2] += ak * g2;
        }
        for (int k = 0; k < SW_K; ++k) dw[(t * SW_K + k) * ldw + r] = a[(t * SW_K + k) * R + r] * (da[k] - dot);
    }
    for (int k = 0; k < SW_K; ++k) {
        for (int r = R; r < ldw; ++r) dw[(t * SW_K + k) * ldw + r] = 0.f;
        for (int c = 0; c < 3; ++c) dzj[(t * SW_K + k) * 3 + c] = dz[k][c];
    }
}

inline unsigned grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

// C[M,N] = A(M,K) B(K,N) (+ bias[N]); generic element strides.  ws: split-K slabs (>= pf_gemm_ws_floats).
extern "C" long long pf_gemm_ws_floats(int M, int N, int K) {
    int bm, bn;
    gemm_tile_dims(gemm_shape(M, N), bm, bn);
    const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    // split-K when the output has too few tiles to fill 256 CUs (dW GEMMs: tiny M x N, K = rows up to 131072):
    // aim at ~1024 workgroups, K chunks of at least 128, slabs capped at 16 M floats
    int split = 1;
    if (tiles < 512 && K >= 1024) {
        long long sp = (1024 + tiles - 1) / tiles;
        if (sp > K / 128) sp = K / 128;
        const long long cap = (16ll << 20) / ((long long)M * N);
        if (sp > cap) sp = cap;
        if (sp > 256) sp = 256;
        split = sp < 1 ? 1 : (int)sp;
    }
    return split > 1 ? (long long)split * M * N : 0;
}

// arith: 0 = f32 MFMA (bit-exact fp32 fma chain; gemm2_kernel when both operands take float4 loads), 1 = the same on the round-1
// kernel (gemm_kernel: the A/B reference of tests/test_gpu_train_fused.py, bit-identical results), 2 = split-fp16 (forward GEMMs), 3 = split-bf16 (gradient operands)
extern "C" int pf_gemm_ex(int arith, const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn,
                          float* C, long long ldc, const float* bias, int M, int N, int K, float* ws, long long ws_floats,
                          void* stream) {
    return pf_gemm_addend(arith, A, sam, sak, B, sbk, sbn, C, ldc, bias, nullptr, M, N, K, ws, ws_floats, stream, nullptr);
}
// (internal, pf_api_internal.h) the same with an addend: C = A B + bias + addend, addend [M, ldc] laid out like C (it may be C)
int pf_gemm_addend(int arith, const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn,
                   float* C, long long ldc, const float* bias, const float* addend, int M, int N, int K, float* ws,
                   long long ws_floats, void* stream, int* slabs_left) {
    if (slabs_left) *slabs_left = 0;
    if (arith != 0 && arith != 1 && arith != 2 && arith != 3) return PF_ERR_UNSUPPORTED;
    if (!A || !B || !C) return PF_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0) return PF_ERR_SHAPE;
    const long long need = pf_gemm_ws_floats(M, N, K);
    int split = need ? (int)(need / ((long long)M * N)) : 1;
    if (need && (!ws || ws_floats < need)) return PF_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const bool use_ws = split > 1;
    GemmArgs g{A, sam, sak, B, sbk, sbn, use_ws ? ws : C, use_ws ? (long long)N : ldc, use_ws ? nullptr : bias, M, N, K, 0,
               use_ws ? nullptr : addend};
    g.kchunk = ((K + split - 1) / split + 31) / 32 * 32;
    split = (K + g.kchunk - 1) / g.kchunk;
    // float4 path: the contiguous dimension of each operand must be 4-aligned in extent, stride and base
    auto al16 = [](const void* p) { return (reinterpret_cast<unsigned long long>(p) & 15ull) == 0; };
    const bool va = (sak == 1) ? (K % 4 == 0 && sam % 4 == 0 && g.kchunk % 4 == 0) : (sam == 1 && M % 4 == 0 && sak % 4 == 0);
    const bool vb = (sbn == 1) ? (N % 4 == 0 && sbk % 4 == 0) : (sbk == 1 && K % 4 == 0 && sbn % 4 == 0);
    const bool vec = va && vb && al16(A) && al16(B);
    if (arith >= 2) {
        // float4 staging of the split kernel: 4 consecutive elements along each operand's contiguous dimension
        const bool va2 = (sak == 1) ? (sam % 4 == 0) : (sam == 1 && sak % 4 == 0);
        const bool vb2 = (sbk == 1) ? (sbn % 4 == 0) : (sbn == 1 && sbk % 4 == 0);
        const bool vec2 = va2 && vb2 && al16(A) && al16(B) && (sak == 1 || sam == 1) && (sbk == 1 || sbn == 1);
        switch (gemm_shape(M, N)) {
            case 0: gemm_split_launch<2, 2, 4, 4>(arith, g, split, vec2, s); break;
            case 1: gemm_split_launch<4, 1, 4, 1>(arith, g, split, vec2, s); break;
            case 2: gemm_split_launch<4, 1, 4, 2>(arith, g, split, vec2, s); break;
            case 3: gemm_split_launch<4, 1, 4, 4>(arith, g, split, vec2, s); break;
            case 4: gemm_split_launch<1, 4, 1, 4>(arith, g, split, vec2, s); break;
            case 5: gemm_split_launch<1, 4, 2, 4>(arith, g, split, vec2, s); break;
            case 7: case 8: case 9: gemm_split_launch<2, 2, 2, 2>(arith, g, split, vec2, s); break;
            default: gemm_split_launch<1, 4, 4, 4>(arith, g, split, vec2, s); break;
        }
    } else if (arith == 0 && vec) {
        switch (gemm_shape(M, N)) {
            case 0: gemm2_launch<2, 2, 4, 4>(g, split, s); break;
            case 1: gemm2_launch<4, 1, 4, 1>(g, split, s); break;
            case 2: gemm2_launch<4, 1, 4, 2>(g, split, s); break;
            case 3: gemm2_launch<4, 1, 4, 4>(g, split, s); break;
            case 4: gemm2_launch<1, 4, 1, 4>(g, split, s); break;
            case 5: gemm2_launch<1, 4, 2, 4>(g, split, s); break;
            case 7: gemm2_launch<2, 2, 2, 2>(g, split, s); break;
            case 8: gemm2_launch<2, 2, 1, 2>(g, split, s); break;
            case 9: gemm2_launch<2, 2, 1, 1>(g, split, s); break;
            default: gemm2_launch<1, 4, 4, 4>(g, split, s); break;
        }
    } else
    switch (gemm_shape(M, N)) {
        case 0: gemm_launch<2, 2, 4, 4>(g, split, vec, s); break;
        case 1: gemm_launch<4, 1, 4, 1>(g, split, vec, s); break;
        case 2: gemm_launch<4, 1, 4, 2>(g, split, vec, s); break;
        case 3: gemm_launch<4, 1, 4, 4>(g, split, vec, s); break;
        case 4: gemm_launch<1, 4, 1, 4>(g, split, vec, s); break;
        case 5: gemm_launch<1, 4, 2, 4>(g, split, vec, s); break;
        case 7: case 8: case 9: gemm_launch<2, 2, 2, 2>(g, split, vec, s); break;
        default: gemm_launch<1, 4, 4, 4>(g, split, vec, s); break;
    }
    if (use_ws && slabs_left && !bias && !addend) *slabs_left = split;       // the caller sums the slabs itself
    else if (use_ws)
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)(((long long)M * N + 63) / 64)), dim3(256), 0, s, ws, C, bias, M,
                           N, ldc, split, addend);
    return pf_last_launch_status();
}

// C [M, ldc] = sum of nslab split-K slabs [nslab][M * N] (fixed combine order): the reduction step of pf_gemm, for callers that
// produce their own slabs (csrc/train_fused.hip)
extern "C" int pf_gemm_reduce(const float* slabs, float* C, int M, int N, long long ldc, int nslab, void* stream) {
    if (!slabs || !C) return PF_ERR_NULL;
    if (M <= 0 || N <= 0 || nslab <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)(((long long)M * N + 63) / 64)), dim3(256), 0, (hipStream_t)stream, slabs, C,
                       (const float*)nullptr, M, N, ldc, nslab, (const float*)nullptr);
    return pf_last_launch_status();
}

extern "C" int pf_gemm(const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn, float* C,
                       long long ldc, const float* bias, int M, int N, int K, float* ws, long long ws_floats, void* stream) {
    return pf_gemm_ex(0, A, sam, sak, B, sbk, sbn, C, ldc, bias, M, N, K, ws, ws_floats, stream);
}

// BatchNorm(train) + LeakyReLU forward on x [R,C].  save [2][C] = mean, invstd (out); running stats updated in place
// (nullable); ws >= 2*nchunk*C + 2*C floats with nchunk = pf_bn_chunks(R).
extern "C" int pf_bn_chunks(long long R) { long long n = (R + 1023) / 1024; return (int)(n > 1024 ? 1024 : (n < 1 ? 1 : n)); }

extern "C" int pf_bn_lrelu_fwd(const float* x, long long R, int C, const float* gamma, const float* beta, float slope,
                               float eps, float momentum, float* run_mean, float* run_var, float* y, float* save, float* ws,
                               void* stream) {
    if (!x || !gamma || !beta || !y || !save || !ws) return PF_ERR_NULL;
    if (R <= 1 || C <= 0) return PF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = pf_bn_chunks(R);
    const int rows_per = (int)((R + nchunk - 1) / nchunk);
    float* partial = ws;
    float* var_b = ws + (long long)2 * nchunk * C;
    dim3 gc((C + 63) / 64);
    colstat_launch<0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, R, C, nchunk, rows_per, partial, s);
    hipLaunchKernelGGL(colstat_final_kernel, gc, dim3(256), 0, s, partial, nchunk, C, 1, 1.0f / (float)R, save);
    colstat_launch<1>(x, nullptr, save, nullptr, nullptr, nullptr, 0.f, R, C, nchunk, rows_per, partial, s);
    hipLaunchKernelGGL(colstat_final_kernel, gc, dim3(256), 0, s, partial, nchunk, C, 1, 1.0f / (float)R, var_b);
    hipLaunchKernelGGL(bn_finish_kernel, gc, dim3(64), 0, s, var_b, C, eps, momentum, (float)R / (float)(R - 1), save, save + C,
                       run_mean, run_var, (float*)nullptr);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(R * C)), dim3(256), 0, s, x, save, save + C, gamma, beta, slope, R * C, C, y);
    return pf_last_launch_status();
}

// backward: dx [R,C], dgamma [C], dbeta [C]
extern "C" int pf_bn_lrelu_bwd(const float* x, const float* dy, long long R, int C, const float* gamma, const float* beta,
                               float slope, const float* save, float* dx, float* dgamma, float* dbeta, float* ws,
                               void* stream) {
    if (!x || !dy || !gamma || !beta || !save || !dx || !dgamma || !dbeta || !ws) return PF_ERR_NULL;
    if (R <= 1 || C <= 0) return PF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = pf_bn_chunks(R);
    const int rows_per = (int)((R + nchunk - 1) / nchunk);
    float* partial = ws;
    float* sums = ws + (long long)2 * nchunk * C;          // [2][C]: sum dz, sum dz*xhat
    dim3 gc((C + 63) / 64);
    colstat_launch<2>(x, dy, save, save + C, gamma, beta, slope, R, C, nchunk, rows_per, partial, s);
    hipLaunchKernelGGL(colstat_final_bwd_kernel, gc, dim3(256), 0, s, partial, nchunk, C, 1.0f / (float)R, dbeta, dgamma, sums);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(R * C)), dim3(256), 0, s, x, dy, save, save + C, gamma, beta, sums, slope,
                       R * C, C, dx);
    return pf_last_launch_status();
}

// ---- the same BatchNorm in stages, so that the per-column sums can be all-reduced between them (SyncBN across ranks:
// puflow_amd/train_ops.py SyncBnLreluFn).  Statistics are UNSCALED sums here; the host divides by the global row count.
// out[c] = sum_r x[r,c]  (mean_in == NULL)  or  sum_r (x[r,c] - mean_in[c])^2.   ws >= 2*nchunk*C floats.
extern "C" int pf_bn_colstat(const float* x, long long R, int C, const float* mean_in, float* out, float* ws, void* stream) {
    if (!x || !out || !ws) return PF_ERR_NULL;
    if (R <= 0 || C <= 0) return PF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = pf_bn_chunks(R);
    const int rows_per = (int)((R + nchunk - 1) / nchunk);
    if (mean_in) colstat_launch<1>(x, nullptr, mean_in, nullptr, nullptr, nullptr, 0.f, R, C, nchunk, rows_per, ws, s);
    else colstat_launch<0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, R, C, nchunk, rows_per, ws, s);
    hipLaunchKernelGGL(colstat_final_kernel, dim3((C + 63) / 64), dim3(256), 0, s, ws, nchunk, C, 1, 1.0f, out);
    return pf_last_launch_status();
}

// y = lrelu(gamma (x - mean) / sqrt(var_b + eps) + beta) with GIVEN (global) mean / biased variance; save = [mean | invstd];
// running statistics updated with momentum and var_b * unbias (nullable).
extern "C" int pf_bn_apply_stats(const float* x, long long R, int C, const float* mean, const float* var_b, float unbias,
                                 const float* gamma, const float* beta, float slope, float eps, float momentum,
                                 float* run_mean, float* run_var, float* y, float* save, void* stream) {
    if (!x || !mean || !var_b || !gamma || !beta || !y || !save) return PF_ERR_NULL;
    if (R <= 0 || C <= 0) return PF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 63) / 64), dim3(64), 0, s, var_b, C, eps, momentum, unbias, mean, save + C,
                       run_mean, run_var, save);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(R * C)), dim3(256), 0, s, x, save, save + C, gamma, beta, slope, R * C, C, y);
    return pf_last_launch_status();
}

// backward, stage 1: sums[2][C] = (sum dz, sum dz * xhat) over THIS rank's rows (= dbeta, dgamma of the local loss)
extern "C" int pf_bn_bwd_sums(const float* x, const float* dy, long long R, int C, const float* gamma, const float* beta,
                              float slope, const float* save, float* sums, float* ws, void* stream) {
    if (!x || !dy || !gamma || !beta || !save || !sums || !ws) return PF_ERR_NULL;
    if (R <= 0 || C <= 0) return PF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = pf_bn_chunks(R);
    const int rows_per = (int)((R + nchunk - 1) / nchunk);
    colstat_launch<2>(x, dy, save, save + C, gamma, beta, slope, R, C, nchunk, rows_per, ws, s);
    hipLaunchKernelGGL(colstat_final_kernel, dim3((C + 63) / 64), dim3(256), 0, s, ws, nchunk, C, 2, 1.0f, sums);
    return pf_last_launch_status();
}

// backward, stage 2: dx from the GLOBAL means[2][C] = all-reduced sums / global row count
extern "C" int pf_bn_bwd_apply(const float* x, const float* dy, long long R, int C, const float* gamma, const float* beta,
                               float slope, const float* save, const float* means, float* dx, void* stream) {
    if (!x || !dy || !gamma || !beta || !save || !means || !dx) return PF_ERR_NULL;
    if (R <= 0 || C <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(R * C)), dim3(256), 0, (hipStream_t)stream, x, dy, save, save + C, gamma,
                       beta, means, slope, R * C, C, dx);
    return pf_last_launch_status();
}

// column sums of g [R,C] -> out [C] (bias gradients).  ws >= 2*nchunk*C floats.
extern "C" int pf_colsum(const float* g, long long R, int C, float* out, float* ws, void* stream) {
    if (!g || !out || !ws) return PF_ERR_NULL;
    if (R <= 0 || C <= 0) return PF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = pf_bn_chunks(R);
    const int rows_per = (int)((R + nchunk - 1) / nchunk);
    colstat_launch<0>(g, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, R, C, nchunk, rows_per, ws, s);
    hipLaunchKernelGGL(colstat_final_kernel, dim3((C + 63) / 64), dim3(256), 0, s, ws, nchunk, C, 1, 1.0f, out);
    return pf_last_launch_status();
}

extern "C" int pf_act_fwd(const float* x, float slope, long long total, float* y, void* stream) {
    if (!x || !y) return PF_ERR_NULL;
    if (total <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, slope, total, y);
    return pf_last_launch_status();
}
extern "C" int pf_act_bwd(const float* y, const float* dy, float slope, long long total, float* dx, void* stream) {
    if (!y || !dy || !dx) return PF_ERR_NULL;
    if (total <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, y, dy, slope, total, dx);
    return pf_last_launch_status();
}

extern "C" int pf_edge_feature_fwd(const float* x, const int* idx, int B, int N, int K, int C, float* out, void* stream) {
    if (!x || !idx || !out) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || K <= 0 || C <= 0) return PF_ERR_SHAPE;
    const long long total = (long long)B * N * K * C;
    hipLaunchKernelGGL(edge_feature_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, idx, N, K, C, total, out);
    return pf_last_launch_status();
}
// dx [B*N, C] must be zero-filled by the caller (accumulates)
extern "C" int pf_edge_feature_bwd(const float* g, const int* idx, int B, int N, int K, int C, float* dx, void* stream) {
    if (!g || !idx || !dx) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || K <= 0 || C <= 0) return PF_ERR_SHAPE;
    const long long total = (long long)B * N * K * C;
    hipLaunchKernelGGL(edge_feature_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, g, idx, N, K, C, total, dx);
    return pf_last_launch_status();
}

extern "C" int pf_maxpool_k_fwd(const float* y, long long T, int K, int C, float* out, int* arg, void* stream) {
    if (!y || !out || !arg) return PF_ERR_NULL;
    if (T <= 0 || K <= 0 || C <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(T * C)), dim3(256), 0, (hipStream_t)stream, y, K, C, T * C, out, arg);
    return pf_last_launch_status();
}
extern "C" int pf_maxpool_k_bwd(const float* dy, const int* arg, long long T, int K, int C, float* dx, void* stream) {
    if (!dy || !arg || !dx) return PF_ERR_NULL;
    if (T <= 0 || K <= 0 || C <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(T * K * C)), dim3(256), 0, (hipStream_t)stream, dy, arg, K, C, T * K * C, dx);
    return pf_last_launch_status();
}

// out [B*N, C] += g [B*N*K, C] scattered by idx (out zero-filled by the caller)
extern "C" int pf_scatter_rows(const float* g, const int* idx, int B, int N, int K, int C, float* out, void* stream) {
    if (!g || !idx || !out) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || K <= 0 || C <= 0) return PF_ERR_SHAPE;
    const long long total = (long long)B * N * K * C;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, g, idx, N, K, C, total, out);
    return pf_last_launch_status();
}

// the same sum as a gather over the transposed lists (pf_knn_csr of idx, lists sorted): one order, run after run - the
// deterministic form (PF_TRAIN_DETERMINISTIC) of the un-fused latent gather's backward
__global__ __launch_bounds__(256) void scatter_rows_det_kernel(const float* __restrict__ g, const int* __restrict__ off,
                                                              const int* __restrict__ edge, int C, long long total,
                                                              float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long long j = t / C;
    const int c = (int)(t - j * C);
    float s = 0.f;
    for (int q = off[j]; q < off[j + 1]; ++q) s += g[(long long)edge[q] * C + c];
    out[t] = s;
}
extern "C" int pf_scatter_rows_det(const float* g, const int* csr_off, const int* csr_edge, long long T, int C, float* out, void* stream) {
    if (!g || !csr_off || !csr_edge || !out) return PF_ERR_NULL;
    if (T <= 0 || C <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(scatter_rows_det_kernel, dim3((unsigned)((T * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, csr_off, csr_edge, C, T * C, out);
    return pf_last_launch_status();
}

extern "C" int pf_group_sum(const float* g, long long T, int R, int C, float* out, void* stream) {
    if (!g || !out) return PF_ERR_NULL;
    if (T <= 0 || R <= 0 || C <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(group_sum_kernel, dim3(grid_for(T * C)), dim3(256), 0, (hipStream_t)stream, g, R, C, T * C, out);
    return pf_last_launch_status();
}

extern "C" int pf_softmax_wsum_fwd(const float* w, int ldw, const float* zj, int K, int R, long long T, float* a, float* fz,
                                   void* stream) {
    if (!w || !zj || !a || !fz) return PF_ERR_NULL;
    if (K != SW_K || R <= 0 || R > SW_RMAX || R > ldw || T <= 0) return PF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(softmax_wsum_fwd_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ldw, zj, R,
                       T, a, fz);
    return pf_last_launch_status();
}
extern "C" int pf_softmax_wsum_bwd(const float* a, const float* zj, const float* dfz, int K, int R, int ldw, long long T,
                                   float* dw, float* dzj, void* stream) {
    if (!a || !zj || !dfz || !dw || !dzj) return PF_ERR_NULL;
    if (K != SW_K || R <= 0 || R > SW_RMAX || R > ldw || T <= 0) return PF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(softmax_wsum_bwd_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, zj, dfz, R,
                       ldw, T, dw, dzj);
    return pf_last_launch_status();
}

// ============================================================================================================
// Flow-block elementwise algebra on [R,3] rows (training path): ActNorm scale/shift, the additive coupling +
// channel reverse + conditional affine injector, and the Gaussian log-likelihood, each with its backward.
// Reference: modules/flows/normalize.py:30-43, coupling.py:55-58,82-85,114-118,132-151, permutate.py:75-80,
// modules/utils/probs.py:73-75.  One thread per row; parameter gradients are reduced with pf_colsum by the caller.
// ============================================================================================================
namespace {

// y = x*exp(logs)+bias (inv=0)   or   y = (x-bias)*exp(-logs) (inv=1)
__global__ __launch_bounds__(256) void actnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ logs,
                                                         const float* __restrict__ bias, int inv, long long R,
                                                         float* __restrict__ y) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = x[r * 3 + c];
        y[r * 3 + c] = inv ? (v - bias[c]) * expf(-logs[c]) : v * expf(logs[c]) + bias[c];
    }
}
// dx, and per-row contributions glogs / gbias [R,3] (column-summed by the caller)
__global__ __launch_bounds__(256) void actnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ logs, const float* __restrict__ bias,
                                                         int inv, long long R, float* __restrict__ dx,
                                                         float* __restrict__ glogs, float* __restrict__ gbias) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g = dy[r * 3 + c], v = x[r * 3 + c];
        if (inv) {
            const float e = expf(-logs[c]);
            dx[r * 3 + c] = g * e;
            gbias[r * 3 + c] = -g * e;
            glogs[r * 3 + c] = -g * (v - bias[c]) * e;
        } else {
            const float e = expf(logs[c]);
            dx[r * 3 + c] = g * e;
            gbias[r * 3 + c] = g;
            glogs[r * 3 + c] = g * v * e;
        }
    }
}

// forward direction: h2 = y[td:] - o ; v = reverse(cat[h1,h2]) ; out = (v - t) * exp(-s)
__global__ __launch_bounds__(256) void couple_inject_fwd_kernel(const float* __restrict__ y, const float* __restrict__ o,
                                                               const float* __restrict__ s, const float* __restrict__ t,
                                                               int td, long long R, float* __restrict__ out) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float h[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) h[c] = y[r * 3 + c] - (c >= td ? o[r * (3 - td) + (c - td)] : 0.f);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[r * 3 + c] = (h[2 - c] - t[r * 3 + c]) * expf(-s[r * 3 + c]);
}
__global__ __launch_bounds__(256) void couple_inject_bwd_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                                               const float* __restrict__ s, int td, long long R,
                                                               float* __restrict__ dy, float* __restrict__ dobuf,
                                                               float* __restrict__ ds, float* __restrict__ dt) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float dv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g = dout[r * 3 + c];
        dv[c] = g * expf(-s[r * 3 + c]);
        ds[r * 3 + c] = -g * out[r * 3 + c];
        dt[r * 3 + c] = -dv[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g = dv[2 - c];                       // un-reverse
        dy[r * 3 + c] = g;
        if (c >= td) dobuf[r * (3 - td) + (c - td)] = -g;
    }
}

// inverse direction, first half: v = reverse(u * exp(s) + t)
__global__ __launch_bounds__(256) void inject_inv_fwd_kernel(const float* __restrict__ u, const float* __restrict__ s,
                                                            const float* __restrict__ t, long long R, float* __restrict__ v) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) v[r * 3 + (2 - c)] = fmaf(u[r * 3 + c], expf(s[r * 3 + c]), t[r * 3 + c]);
}
__global__ __launch_bounds__(256) void inject_inv_bwd_kernel(const float* __restrict__ u, const float* __restrict__ s,
                                                            const float* __restrict__ dv, long long R, float* __restrict__ du,
                                                            float* __restrict__ ds, float* __restrict__ dt) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g = dv[r * 3 + (2 - c)], e = expf(s[r * 3 + c]);
        du[r * 3 + c] = g * e;
        ds[r * 3 + c] = g * u[r * 3 + c] * e;
        dt[r * 3 + c] = g;
    }
}
// inverse direction, second half: out = cat[v[:td], v[td:] + o]   (backward: dv = dout, do = dout[td:])
__global__ __launch_bounds__(256) void couple_add_kernel(const float* __restrict__ v, const float* __restrict__ o, int td,
                                                        long long R, float* __restrict__ out) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[r * 3 + c] = v[r * 3 + c] + (c >= td ? o[r * (3 - td) + (c - td)] : 0.f);
}
__global__ __launch_bounds__(256) void slice_tail_kernel(const float* __restrict__ g, int td, long long R, float* __restrict__ o) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    for (int c = td; c < 3; ++c) o[r * (3 - td) + (c - td)] = g[r * 3 + c];
}

// per-batch sums over M consecutive values: out[b] = sum_m f(x[b*M + m]);  mode 0: x, mode 1: -0.5*(x^2 + log 2 pi)
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ x, long long M, int mode, float* __restrict__ out) {
    __shared__ float sh[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    for (long long m = tid; m < M; m += 256) {
        const float v = x[(long long)b * M + m];
        acc += mode == 0 ? v : -0.5f * (v * v + 1.8378770664093453f);
    }
    sh[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    if (tid == 0) out[b] = sh[0];
}
// dx[b*M + m] = g[b] * (mode 0: 1, mode 1: -x)
__global__ __launch_bounds__(256) void batch_sum_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, long long M,
                                                           int mode, long long total, float* __restrict__ dx) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const float gg = g[i / M];
        dx[i] = mode == 0 ? gg : -gg * x[i];
    }
}

inline unsigned rows_grid(long long R) { return (unsigned)((R + 255) / 256); }

}  // namespace

extern "C" int pf_actnorm_fwd(const float* x, const float* logs, const float* bias, int inv, long long R, float* y, void* stream) {
    if (!x || !logs || !bias || !y) return PF_ERR_NULL;
    if (R <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(actnorm_fwd_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, x, logs, bias, inv, R, y);
    return pf_last_launch_status();
}
extern "C" int pf_actnorm_bwd(const float* x, const float* dy, const float* logs, const float* bias, int inv, long long R,
                              float* dx, float* glogs_rows, float* gbias_rows, void* stream) {
    if (!x || !dy || !logs || !bias || !dx || !glogs_rows || !gbias_rows) return PF_ERR_NULL;
    if (R <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(actnorm_bwd_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, x, dy, logs, bias, inv, R, dx,
                       glogs_rows, gbias_rows);
    return pf_last_launch_status();
}
extern "C" int pf_couple_inject_fwd(const float* y, const float* o, const float* s, const float* t, int td, long long R,
                                    float* out, void* stream) {
    if (!y || !o || !s || !t || !out) return PF_ERR_NULL;
    if (R <= 0 || td < 1 || td > 2) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(couple_inject_fwd_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, y, o, s, t, td, R, out);
    return pf_last_launch_status();
}
extern "C" int pf_couple_inject_bwd(const float* out, const float* dout, const float* s, int td, long long R, float* dy,
                                    float* do_, float* ds, float* dt, void* stream) {
    if (!out || !dout || !s || !dy || !do_ || !ds || !dt) return PF_ERR_NULL;
    if (R <= 0 || td < 1 || td > 2) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(couple_inject_bwd_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, out, dout, s, td, R, dy,
                       do_, ds, dt);
    return pf_last_launch_status();
}
extern "C" int pf_inject_inv_fwd(const float* u, const float* s, const float* t, long long R, float* v, void* stream) {
    if (!u || !s || !t || !v) return PF_ERR_NULL;
    if (R <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(inject_inv_fwd_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, u, s, t, R, v);
    return pf_last_launch_status();
}
extern "C" int pf_inject_inv_bwd(const float* u, const float* s, const float* dv, long long R, float* du, float* ds, float* dt,
                                 void* stream) {
    if (!u || !s || !dv || !du || !ds || !dt) return PF_ERR_NULL;
    if (R <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(inject_inv_bwd_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, u, s, dv, R, du, ds, dt);
    return pf_last_launch_status();
}
extern "C" int pf_couple_add(const float* v, const float* o, int td, long long R, float* out, void* stream) {
    if (!v || !o || !out) return PF_ERR_NULL;
    if (R <= 0 || td < 1 || td > 2) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(couple_add_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, v, o, td, R, out);
    return pf_last_launch_status();
}
extern "C" int pf_slice_tail(const float* g, int td, long long R, float* o, void* stream) {
    if (!g || !o) return PF_ERR_NULL;
    if (R <= 0 || td < 1 || td > 2) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(slice_tail_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, g, td, R, o);
    return pf_last_launch_status();
}
extern "C" int pf_batch_sum_fwd(const float* x, int B, long long M, int mode, float* out, void* stream) {
    if (!x || !out) return PF_ERR_NULL;
    if (B <= 0 || M <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(batch_sum_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, M, mode, out);
    return pf_last_launch_status();
}
extern "C" int pf_batch_sum_bwd(const float* x, const float* g, int B, long long M, int mode, float* dx, void* stream) {
    if (!x || !g || !dx) return PF_ERR_NULL;
    if (B <= 0 || M <= 0) return PF_ERR_SHAPE;
    const long long total = (long long)B * M;
    hipLaunchKernelGGL(batch_sum_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, g, M, mode, total, dx);
    return pf_last_launch_status();
}

// DistanceEncoder.distance_vec (interpflow.py:100-115): out [B*N*K, 10] = [x_i, x_j, x_i - x_j, |x_i - x_j|] (inputs only: no backward)
namespace {
__global__ __launch_bounds__(256) void dist_feature_kernel(const float* __restrict__ xyz, const int* __restrict__ idx, int N, int K,
                                                          long long E, float* __restrict__ out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const long long t = e / K;
    const long long j = (t / N) * N + idx[e];
    float* o = out + e * 10;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = xyz[t * 3 + c], b = xyz[j * 3 + c];
        o[c] = a; o[3 + c] = b; v[c] = __fsub_rn(a, b); o[6 + c] = v[c];
    }
    o[9] = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(v[0], v[0]), __fmul_rn(v[1], v[1])), __fmul_rn(v[2], v[2])));
}
}  // namespace

extern "C" int pf_dist_feature(const float* xyz, const int* idx, int B, int N, int K, float* out, void* stream) {
    if (!xyz || !idx || !out) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || K <= 0) return PF_ERR_SHAPE;
    const long long E = (long long)B * N * K;
    hipLaunchKernelGGL(dist_feature_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xyz, idx, N, K, E, out);
    return pf_last_launch_status();
}
