// Fused EdgeConv dense block: gather -> 4x(1x1 conv + BN + LeakyReLU, dense/growing) -> 1x1 conv
// -> max over the K=16 neighbours.  Replaces FeatureExtractUnit.forward
// (modules/discrete/interpflow.py:190-248) in eval mode; BN folded on the host, edge feature
// [x_i, x_j, x_j - x_i] folded into per-point vectors P[i] / Q[j] (puflow_amd/packing.py).
//
// Mapping: one MFMA column tile = the 16 neighbours of ONE point; output channels on MFMA rows.
// Growth features never leave registers (pf_mfma.h layout); the only per-edge memory traffic is
// the gather of Q[j] (S floats) as accumulator initialisers.  A wave processes P points at a time
// (P independent MFMA chains sharing each weight fragment).
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"
#include "pf_mfma.h"

namespace {

struct EcArgs {
    const float* pq;     // [T, 2S]   P | Q   (PQ variant)            or nullptr
    const float* xyz;    // [T, 3]                                  (C3 variant)
    const float* tab;    // [S, 8]  PA(3) QB(3) pb 0                (C3 variant)
    const int* idx;      // [T, 16] neighbour index inside the batch item
    const f4* wg;        // fragment-packed G1..G_{NCONV-1}, Gout
    float* out;          // [T, ODIM]
    int T;               // B*N points
    int N;               // points per batch item
    int chunk;           // ceil(ntiles / 8) for the XCD-aware tile order
    int ntiles;          // workgroup tiles
    // fused next-unit P|Q GEMM (PQF kernels, small batches): the arguments of pointwise.hip pq_gemm_kernel
    const u4* pqf_w;     // f16n fragment image [ROWS/16][CP][hi / lo][64 lanes]
    const float* pqf_bias;
    const float* pqf_scales;
    float* pqf_out;      // [T, ROWS]
};

// ---- the next unit's P|Q vectors from inside the producing EdgeConv kernel (small batches) --------------------------------
// At 4 x 2048 points pf_pq_gemm is a 12 us launch for 2 us of work (weight prologue, staging, drain); here the 16 points a
// workgroup has just finished become one MFMA column tile: every wave leaves its points' pooled features in LDS as ready-made
// B operands (fp32 -> hi / natural lo, the split of pf_pairn), one barrier, then every wave multiplies the tile by ITS rows of
// the [ROWS x ODIM] matrix (fragments straight from L2: 16 KiB per wave and tile - fine for a few tiles per workgroup, too
// much L2 traffic next to the Q gathers at 32 x 2048, where the separate HBM-bound kernel stays).  Up to PQF_NT tiles are
// staged before the GEMM runs (one weight fetch and its L2 latency per flush, not per tile).  Products, order and rescale
// are those of pq_gemm_kernel: the table is bit-identical.  hb: [PQF_NT tiles][CP][hi / lo][64 lanes] x 16 B.
template <int ODIM>
__device__ __forceinline__ void pqf_stage(u4* hb, int ch, int pt, float v) {
    _Float16* h = reinterpret_cast<_Float16*>(hb);
    const int cp = ch >> 5, j = ((ch >> 4) & 1) * 4 + (ch & 3), slot = 16 * ((ch & 15) >> 2) + pt;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    h[((cp * 2 + 0) * 64 + slot) * 8 + j] = hi;
    h[((cp * 2 + 1) * 64 + slot) * 8 + j] = lo;
}

constexpr int PQF_NT = 4;      // workgroup tiles staged before the GEMM runs on them (its weights come from L2 once per flush)

template <int ODIM, int ROWS, int NW>
__device__ __forceinline__ void pqf_gemm(const EcArgs& a, const u4* hb, int nst, const int (&tpt)[PQF_NT], int wave_u, int lane) {
    constexpr int CP = ODIM / 32, RB = ROWS / 16 / NW;
    static_assert(ODIM % 32 == 0 && ROWS % (16 * NW) == 0, "rows split evenly over the waves");
    // weight fragments of OBW of this wave's RB 16-row blocks are fetched together: all of them in the 8-wave kernels (256
    // VGPRs per wave: one L2 latency per flush), one block at a time in the 16-wave kernel (128 VGPRs, none to spare)
    constexpr int OBW = NW <= 8 ? RB : 1;
    const int col = lane & 15, q = lane >> 4;
    const PfW2BufD<CP> ws(a.pqf_w, lane);
    const float inv = a.pqf_scales[6];
#pragma unroll
    for (int ob0 = 0; ob0 < RB; ob0 += OBW) {
        h8 wh[OBW][CP], wl[OBW][CP];
#pragma unroll
        for (int o = 0; o < OBW; ++o)
#pragma unroll
            for (int cp = 0; cp < CP; ++cp) {
                wh[o][cp] = ws.load((wave_u * RB + ob0 + o) * CP + cp, 0);
                wl[o][cp] = ws.load((wave_u * RB + ob0 + o) * CP + cp, 1);
            }
#pragma unroll
        for (int t = 0; t < PQF_NT; ++t) {
            if (t >= nst) break;                              // uniform
            const u4* hbt = hb + t * (CP * 2 * 64);
            f4 acc[OBW];
#pragma unroll
            for (int o = 0; o < OBW; ++o) acc[o] = pf_splat(0.f);
#pragma unroll
            for (int cp = 0; cp < CP; ++cp) {
                const h8 fh = __builtin_bit_cast(h8, hbt[(cp * 2 + 0) * 64 + lane]), fl = __builtin_bit_cast(h8, hbt[(cp * 2 + 1) * 64 + lane]);
#pragma unroll
                for (int o = 0; o < OBW; ++o) {
                    f4 x = acc[o];
                    if constexpr (PF_MMN_TERMS == 3) {
                        x = pf_mfma_f16(wh[o][cp], fl, x);
                        x = pf_mfma_f16(wl[o][cp], fh, x);
                    }
                    x = pf_mfma_f16(wh[o][cp], fh, x);
                    acc[o] = x;
                }
            }
            const int pt = tpt[t] + col;
            if (pt < a.T) {
#pragma unroll
                for (int o = 0; o < OBW; ++o) {
                    const f4 bias = *reinterpret_cast<const f4*>(a.pqf_bias + (wave_u * RB + ob0 + o) * 16 + 4 * q);     // L1-resident
                    f4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[o][r], inv, bias[r]);
                    *reinterpret_cast<f4*>(a.pqf_out + (size_t)pt * ROWS + (wave_u * RB + ob0 + o) * 16 + 4 * q) = v;
                }
            }
        }
    }
}

// staged-tile bookkeeping of a PQF kernel: slot of the current tile, first point of every staged tile (uniform values)
struct PqfState {
    int nst = 0;
    int tpt[PQF_NT] = {0, 0, 0, 0};
    __device__ __forceinline__ void push(int pt0) {
#pragma unroll
        for (int k = 0; k < PQF_NT; ++k) tpt[k] = k == nst ? pt0 : tpt[k];
        ++nst;
    }
};

template <int GB, int NCONV, int ODIM, bool C3, int P, int NW>
__global__ __launch_bounds__(NW * 64) void edgeconv_kernel(EcArgs a) {
    constexpr int G = GB * 16;
    constexpr int S = G * NCONV + ODIM;
    constexpr int NF = GB * NCONV;          // growth feature blocks
    constexpr int OBO = ODIM / 16;
    constexpr int OCH = 2;                  // conv_out blocks per accumulator chunk
    static_assert(OBO <= 16 && OBO % OCH == 0, "pooled store uses one lane column per output block");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;

    // all growth weights of the unit stay resident in LDS for the life of the (persistent) workgroup
    constexpr int NWF = GB * GB * (NCONV * (NCONV - 1) / 2) + OBO * NF;     // fragments (1 KiB each)
    __shared__ f4 wlds[NWF * 64];
    pf_stage_lds(wlds, a.wg, NWF * 64);
    __syncthreads();
    const PfWLds ws{wlds, lane};

    for (int v = blockIdx.x; v < 8 * a.chunk; v += gridDim.x) {
        const int tile = pf_xcd_tile(v, a.chunk);
        if (tile >= a.ntiles) continue;
        const int pt0 = (tile * NW + wave) * P;

        int gi[P], gj[P];
        float xi[P][3], xj[P][3];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            int g = pt0 + p;
            g = g < a.T ? g : a.T - 1;
            gi[p] = g;
            const int b = g / a.N;
            gj[p] = b * a.N + a.idx[(size_t)g * 16 + col];
            if constexpr (C3) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    xi[p][c] = a.xyz[(size_t)gi[p] * 3 + c];
                    xj[p][c] = a.xyz[(size_t)gj[p] * 3 + c];
                }
            }
        }

        // accumulator initialiser for rows [off, off+4) of the stacked S rows: P[i] + Q[j] (+bias)
        auto init = [&](int p, int off) -> f4 {
            if constexpr (C3) {
                f4 r;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f4* t = reinterpret_cast<const f4*>(a.tab + (size_t)(off + k) * 8);
                    const f4 t0 = t[0], t1 = t[1];
                    float s = t1.z;                                   // bias
                    s = fmaf(t0.x, xi[p][0], s); s = fmaf(t0.y, xi[p][1], s); s = fmaf(t0.z, xi[p][2], s);
                    s = fmaf(t0.w, xj[p][0], s); s = fmaf(t1.x, xj[p][1], s); s = fmaf(t1.y, xj[p][2], s);
                    r[k] = s;
                }
                return r;
            } else {
                const f4 pv = *reinterpret_cast<const f4*>(a.pq + (size_t)gi[p] * (2 * S) + off);
                const f4 qv = *reinterpret_cast<const f4*>(a.pq + (size_t)gj[p] * (2 * S) + S + off);
                return pv + qv;
            }
        };

        // Accumulator initialisers are gathered ONE STAGE AHEAD (stage = growth layer or conv_out chunk):
        // the loads for stage s+1 are issued before stage s's MFMA chain, so their L2 latency hides
        // under ~100+ MFMAs instead of stalling every layer.  Weights come from LDS (lgkmcnt), the
        // gathers from global memory (vmcnt): the two wait counters do not serialise each other.
        constexpr int NI = GB > OCH ? GB : OCH;
        f4 ini[2][P][NI];
        auto load_init = [&](auto nbc, int row0, f4 (&dst)[P][NI]) {
            constexpr int NB = decltype(nbc)::value;
#pragma unroll
            for (int ob = 0; ob < NB; ++ob)
#pragma unroll
                for (int p = 0; p < P; ++p) dst[p][ob] = init(p, row0 + ob * 16 + 4 * q);
        };
        using IGB = std::integral_constant<int, GB>;
        using IOC = std::integral_constant<int, OCH>;
        load_init(IGB{}, 0, ini[0]);
        if constexpr (NCONV > 1) load_init(IGB{}, G, ini[1]); else load_init(IOC{}, G * NCONV, ini[1]);

        f4 feat[P][NF];
        // layer 0: edge part only
#pragma unroll
        for (int ob = 0; ob < GB; ++ob)
#pragma unroll
            for (int p = 0; p < P; ++p) feat[p][ob] = pf_lrelu(ini[0][p][ob], 0.05f);

        // growth layers t = 1..NCONV-1: inputs = feature blocks [0, GB*t)
        pf_static_for<1, NCONV>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t + 1 < NCONV) load_init(IGB{}, G * (t + 1), ini[(t + 1) & 1]);
            else load_init(IOC{}, G * NCONV, ini[(t + 1) & 1]);
            f4 acc[P][GB];
#pragma unroll
            for (int ob = 0; ob < GB; ++ob)
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p][ob] = ini[t & 1][p][ob];
            pf_mm<GB, GB * t, GB * t>(ws, GB * GB * (t * (t - 1) / 2), feat, 0, acc, 0);
#pragma unroll
            for (int ob = 0; ob < GB; ++ob)
#pragma unroll
                for (int p = 0; p < P; ++p) feat[p][GB * t + ob] = pf_lrelu(acc[p][ob], 0.05f);
        });

        // conv_out in chunks of OCH blocks, max over the 16 neighbour columns, lane `col == ob` keeps block ob
        constexpr int FO = GB * GB * (NCONV * (NCONV - 1) / 2);
        f4 sel[P];
#pragma unroll
        for (int p = 0; p < P; ++p) sel[p] = pf_splat(0.f);
        pf_static_for<0, OBO / OCH>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int ob0 = c * OCH;
            constexpr int st = NCONV + c;
            if constexpr (c + 1 < OBO / OCH) load_init(IOC{}, G * NCONV + (ob0 + OCH) * 16, ini[(st + 1) & 1]);
            f4 acc[P][OCH];
#pragma unroll
            for (int o = 0; o < OCH; ++o)
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p][o] = ini[st & 1][p][o];
            pf_mm<OCH, NF, NF>(ws, FO + ob0 * NF, feat, 0, acc, 0);
#pragma unroll
            for (int o = 0; o < OCH; ++o)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    f4 m;
#pragma unroll
                    for (int r = 0; r < 4; ++r) m[r] = pf_rowmax16(acc[p][o][r]);
                    if (col == ob0 + o) sel[p] = m;
                }
        });
#pragma unroll
        for (int p = 0; p < P; ++p)
            if (col < OBO && pt0 + p < a.T)
                *reinterpret_cast<f4*>(a.out + (size_t)gi[p] * ODIM + col * 16 + 4 * q) = sel[p];
    }
}

// timing-only weight source of the -DPF_TUNING_VARIANTS ablation builds (edgeconv4_kernel DBG & 4)
struct EcWConst {                                             // DBG & 4 (timing-only builds): weights without memory traffic
    static constexpr int DEPTH = 2;
    int lane;
    __device__ __forceinline__ h8 load(int frag, int split) const {
        const _Float16 x = (_Float16)(float)((lane + frag + split) & 7);
        return (h8){x, x, x, x, x, x, x, x};
    }
};

// ---- "f16n": split-fp16 with a NATURAL-scale low half, one accumulator, conv_out with the operands swapped ----------
// Arithmetic: x = hi + lo, hi = rne_f16(x), lo = rne_f16(x - hi) (no 2^11 scale: v_fma_mixlo/mixhi_f16 writes it straight into
// the packed operand; gfx950's fp16 MFMA honours subnormal operands, so a small lo keeps an ABSOLUTE precision of 2^-25 in
// the units the operand is stored in).  All three product terms (hi.hi, hi.lo, lo.hi) go into ONE fp32 accumulator: no second
// accumulator, no fold.  To keep the subnormal floor far below fp32 rounding the host stores feature block u as 4^u x_u and the
// weights of layer t on block u as 4^(t-u) W (packing.ec4_scales: exact powers of two, folded into the matrices and into the
// rows of the P|Q table), so the accumulators of layer t hold 4^t v_t and LeakyReLU needs no rescale; conv_out holds 256 y.
// conv_out runs with the operands swapped: an activation fragment is bit-for-bit also the A operand with the EDGES on the
// MFMA rows (tools/probes/split_probe.hip), so D[edge][channel] puts a lane's 4 registers on 4 edges of one channel:
// max over K = 16 is 2 in-lane max + a reduce-scatter over the 4 lane rows (v_permlane32_swap / v_permlane16_swap):
// 28 VALU per point instead of 160.  Its accumulators are initialised with Q[j] alone (dword gathers, 64 B per 16 lanes);
// P[i] is added once per channel after the max.  VALU per point: ~180 (round 1 split-fp16 kernel: ~450).
constexpr float EC4_OUT_INV = 1.f / 256.f;       // conv_out accumulators hold 4^4 y (packing.ec4_scales)

template <int P, int NW, int DBG = 0, bool PQF = false>      // DBG bit mask (-DPF_TUNING_VARIANTS timing builds only; wrong results): 1 no gathers, 2 no MFMAs, 4 no LDS weight reads;  PQF: + the next unit's P|Q vectors (pqf_gemm)
__global__ __launch_bounds__(NW * 64) void edgeconv4_kernel(EcArgs a) {
    static_assert(!PQF || NW * P == 16, "the fused P|Q GEMM takes a workgroup tile of 16 points as one MFMA column tile");
    constexpr int NCONV = 4, G = 32, S = 256, OBO = 8, OCH = 2, ODIM = 128;
    constexpr int NWF = 2 * (NCONV * (NCONV - 1) / 2) + OBO * NCONV;      // 44 (ob, pair) fragments, 2 KiB each
    constexpr int ROWB = 2 * S * 4;                                       // bytes per point of the P|Q table
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    __shared__ u4 wlds[NWF * 2 * 64];
    // P[i] (S floats, the same for all 16 edges of a point) is read ONCE per point as one 1-KiB wave load and staged in a
    // per-wave LDS slot; the per-layer pieces come back as broadcast ds_reads.  Loading them per layer straight into the
    // (edge, 4-channel) lane layout cost as many texture-addresser cycles as the Q gathers themselves (16 lanes fetching
    // the same 16 bytes still take a full quad-lane slot each): 392 -> 272 TA cycles per point.
    __shared__ f4 plds[NW * P][64];
    __shared__ u4 hb[PQF ? PQF_NT * (ODIM / 32) * 2 * 64 : 1];            // PQF: the staged tiles' pooled features as B operands
    PqfState pqf;
    pf_stage_lds(wlds, reinterpret_cast<const u4*>(a.wg), NWF * 2 * 64);
    __syncthreads();
    const PfW2Lds ws_lds{wlds, lane};
    const auto ws = [&] { if constexpr ((DBG & 4) != 0) return EcWConst{lane}; else return ws_lds; }();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.pq), 0, 0x7fffffff, 0x00020000);
    const int blkq = ((q & 1) << 1) | (q >> 1);                           // block a lane row ends up with (reduce-scatter)

    // the tile walk skips the holes of the XCD-aware order; the NEXT tile's indices and P rows are fetched while the
    // current tile computes (the idx -> address -> gather chain is otherwise exposed once per point)
    const int vend = 8 * a.chunk;
    auto next_valid = [&](int v) { while (v < vend && pf_xcd_tile(v, a.chunk) >= a.ntiles) v += gridDim.x; return v; };
    struct Pre { int g[P]; int jc[P]; int4 j4[P]; f4 prow[P]; };
    auto fetch = [&](int v, Pre& n) {
        const int pt0 = (pf_xcd_tile(v < vend ? v : blockIdx.x, a.chunk) * NW + wave) * P;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            int g = pt0 + p;
            g = g < a.T ? g : a.T - 1;
            g = __builtin_amdgcn_readfirstlane(g);
            n.g[p] = g;
            n.jc[p] = a.idx[(size_t)g * 16 + col];
            n.j4[p] = *reinterpret_cast<const int4*>(a.idx + (size_t)g * 16 + 4 * q);
            n.prow[p] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, g * ROWB, 0));
        }
    };
    int v = next_valid(blockIdx.x);
    Pre cur;
    if (v < vend) fetch(v, cur);
    while (v < vend) {
        const int pt0 = (pf_xcd_tile(v, a.chunk) * NW + wave) * P;
        const int vn = next_valid(v + gridDim.x);
        int vQ[P];                     // byte offset of Q[j_col] + this lane's 4 channels (growth layers, edges on columns)
        int vO[P][4];                  // byte offset of Q_out[j_(4q+r)] + channel `col` (conv_out, edges on rows)
        int gout[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int g = cur.g[p];
            const int bN = (g / a.N) * a.N;
            gout[p] = g;
            plds[wave * P + p][lane] = cur.prow[p];
            vQ[p] = (bN + cur.jc[p]) * ROWB + (S + 4 * q) * 4;
            vO[p][0] = (bN + cur.j4[p].x) * ROWB + (S + G * NCONV + col) * 4;
            vO[p][1] = (bN + cur.j4[p].y) * ROWB + (S + G * NCONV + col) * 4;
            vO[p][2] = (bN + cur.j4[p].z) * ROWB + (S + G * NCONV + col) * 4;
            vO[p][3] = (bN + cur.j4[p].w) * ROWB + (S + G * NCONV + col) * 4;
        }
        fetch(vn, cur);                // next tile (a harmless re-read of the first tile when there is none)
        // growth layer t: P_t[i] + Q_t[j] for this lane's 4 channels of both 16-channel blocks
        auto load_g = [&](int t, f4 (&dst)[P][2]) {
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int off = (G * t + 16 * ob) * 4;
                    if constexpr (DBG & 1) { dst[p][ob] = pf_splat((float)(off + vQ[p]) * 1e-9f); continue; }
                    dst[p][ob] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vQ[p], off, 0));
                }
        };
        auto p_g = [&](int p, int t, int ob) -> f4 { return plds[wave * P + p][(G * t + 16 * ob) / 4 + q]; };
        // conv_out chunk: Q_out[j_(4q+r)][16 ob + col], r = 0..3 -> the four accumulator registers
        auto load_o = [&](int ob0, f4 (&dst)[P][2]) {
#pragma unroll
            for (int o = 0; o < OCH; ++o)
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (DBG & 1) { dst[p][o][r] = (float)(vO[p][r] + ob0 + o) * 1e-9f; continue; }
                        dst[p][o][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vO[p][r], (ob0 + o) * 64, 0));
                    }
        };
        f4 ini[2][P][2];
        load_g(0, ini[0]);
        load_g(1, ini[1]);
        PfPairN feat[P][NCONV];
#pragma unroll
        for (int p = 0; p < P; ++p)
            feat[p][0] = pf_pairn(pf_lrelu(ini[0][p][0] + p_g(p, 0, 0), 0.05f), pf_lrelu(ini[0][p][1] + p_g(p, 0, 1), 0.05f));
        pf_static_for<1, NCONV>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t + 1 < NCONV) load_g(t + 1, ini[(t + 1) & 1]); else load_o(0, ini[(t + 1) & 1]);
            f4 acc[P][2];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p][ob] = ini[t & 1][p][ob] + p_g(p, t, ob);
            if constexpr (!(DBG & 2)) pf_mmn<false, 2, t, t>(ws, 2 * (t * (t - 1) / 2), feat, acc);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if constexpr (DBG & 8) { feat[p][t] = feat[p][0]; feat[p][t].h[0] += (_Float16)acc[p][0].x; continue; }
                feat[p][t] = pf_pairn(pf_lrelu(acc[p][0], 0.05f), pf_lrelu(acc[p][1], 0.05f));
            }
        });
        constexpr int FO = 2 * (NCONV * (NCONV - 1) / 2);
        float m[P][OBO];
        pf_static_for<0, OBO / OCH>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int ob0 = c * OCH;
            constexpr int st = NCONV + c;
            if constexpr (c + 1 < OBO / OCH) load_o(ob0 + OCH, ini[(st + 1) & 1]);
            f4 acc[P][OCH];
#pragma unroll
            for (int o = 0; o < OCH; ++o)
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p][o] = ini[st & 1][p][o];
            if constexpr (!(DBG & 2)) pf_mmn<true, OCH, NCONV, NCONV>(ws, FO + ob0 * NCONV, feat, acc);
#pragma unroll
            for (int o = 0; o < OCH; ++o)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    m[p][ob0 + o] = fmaxf(fmaxf(acc[p][o].x, acc[p][o].y), fmaxf(acc[p][o].z, acc[p][o].w));
        });
#pragma unroll
        for (int p = 0; p < P; ++p) {
            // lane row q now holds block blkq (o0) / 4 + blkq (o1), channel `col` of it
            const float o0 = pf_rsmax16(pf_rsmax32(m[p][0], m[p][1]), pf_rsmax32(m[p][2], m[p][3]));
            const float o1 = pf_rsmax16(pf_rsmax32(m[p][4], m[p][5]), pf_rsmax32(m[p][6], m[p][7]));
            const int ch = 16 * blkq + col;
            const float* pr = reinterpret_cast<const float*>(plds[wave * P + p]);
            const float p0 = pr[G * NCONV + ch], p1 = pr[G * NCONV + 64 + ch];
            const float h0 = fmaf(o0, EC4_OUT_INV, p0), h1 = fmaf(o1, EC4_OUT_INV, p1);
            if (pt0 + p < a.T) {
                float* o = a.out + (size_t)gout[p] * ODIM + ch;
                o[0] = h0;
                o[64] = h1;
            }
            if constexpr (PQF) {
                u4* hbt = hb + pqf.nst * ((ODIM / 32) * 2 * 64);
                pqf_stage<ODIM>(hbt, ch, wave * P + p, h0);
                pqf_stage<ODIM>(hbt, ch + 64, wave * P + p, h1);
            }
        }
        if constexpr (PQF) {
            pqf.push(pf_xcd_tile(v, a.chunk) * NW * P);
            if (pqf.nst == PQF_NT || vn >= vend) {             // uniform: flush the staged tiles (two barriers per flush)
                __syncthreads();
                pqf_gemm<ODIM, 2 * S, NW>(a, hb, pqf.nst, pqf.tpt, __builtin_amdgcn_readfirstlane(wave), lane);
                __syncthreads();
                pqf.nst = 0;
            }
        }
        v = vn;
    }
}

template <int P, int NW, int DBG = 0, bool PQF = false>
int launch4(const EcArgs& a0, hipStream_t s) {
    EcArgs a = a0;
    a.ntiles = (a.T + NW * P - 1) / (NW * P);
    a.chunk = (a.ntiles + 7) / 8;
    int grid = 8 * a.chunk;
    if (grid > 256) grid = 256;                           // 88 KiB of LDS: one persistent workgroup per CU
    hipLaunchKernelGGL((edgeconv4_kernel<P, NW, DBG, PQF>), dim3(grid), dim3(NW * 64), 0, s, a);
    return pf_last_launch_status();
}

// ---- f16n variant of the narrow units 0 / 1 (same arithmetic and scale plan as edgeconv4_kernel) -------------------
// Growth layers are 16 channels wide (unit 0: 8, zero-padded): two layers share one 32-channel MFMA step.  conv_out
// runs with the operands swapped (edges on the MFMA rows) and a reduce-scatter max-pool; feature block t is stored as
// 4^t x_t, the accumulators of layer t hold 4^t v_t, conv_out's 256 y (packing.ec1n_scales).
// C3 (unit 0): every pre-activation is one MFMA step against the folded edge table (raw inputs e = (x_i, x_j, 1) in 8 of
// the 32 k-slots); the same e registers are the B operand of the growth rows and the A operand of the conv_out rows.
// PQ (unit 1): Q[j] gathers as accumulator initialisers, P[i] staged per wave in LDS (one load per point).
// Fragments in LDS: G1 | G2 | G3 (2 pairs) | Gout (OBO x 2 pairs) | [C3: edge table, S/16 x 1 pair].
template <int ODIM, bool C3, int P, int NW, bool PQF = false>      // PQF: + the next unit's P|Q vectors (pqf_gemm; ROWS = 256 after unit 0, 512 after unit 1)
__global__ __launch_bounds__(NW * 64) void edgeconv1n_kernel(EcArgs a) {
    static_assert(!PQF || NW * P == 16, "the fused P|Q GEMM takes a workgroup tile of 16 points as one MFMA column tile");
    constexpr int NCONV = 4, S = 16 * NCONV + ODIM, SB = S / 16, OBO = ODIM / 16;
    constexpr int PQ_ROWS = ODIM == 32 ? 256 : 512;
    constexpr int FO = 4, FT = FO + OBO * 2, NWF = FT + (C3 ? SB : 0);
    constexpr int ROWB = 2 * S * 4;                                       // bytes per point of the P|Q table (unit 1)
    static_assert(OBO == 2 || OBO == 4, "reduce-scatter below is written for 2 or 4 output blocks");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    __shared__ u4 wlds[NWF * 128];
    __shared__ float plds[C3 ? 1 : NW * P][C3 ? 1 : S];
    __shared__ u4 hb[PQF ? PQF_NT * (ODIM / 32) * 2 * 64 : 1];
    PqfState pqf;
    pf_stage_lds(wlds, reinterpret_cast<const u4*>(a.wg), NWF * 128);
    __syncthreads();
    const PfW2Lds ws{wlds, lane};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(C3 ? a.xyz : a.pq), 0, 0x7fffffff, 0x00020000);

    for (int v = blockIdx.x; v < 8 * a.chunk; v += gridDim.x) {
        const int tile = pf_xcd_tile(v, a.chunk);
        if (tile >= a.ntiles) continue;
        const int pt0 = (tile * NW + wave) * P;
        int gi[P], vQ[P], vO[P][4];
        PfPairN e[P][1];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            int g = pt0 + p;
            g = g < a.T ? g : a.T - 1;
            g = __builtin_amdgcn_readfirstlane(g);
            gi[p] = g;
            const int bN = (g / a.N) * a.N;
            const int jc = bN + a.idx[(size_t)g * 16 + col];
            if constexpr (C3) {
                float xi[3], xj[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    xi[c] = a.xyz[(size_t)g * 3 + c];
                    xj[c] = a.xyz[(size_t)jc * 3 + c];
                }
                const f4 z4 = pf_splat(0.f);
                const f4 e0 = {xi[0], xi[1], xi[2], xj[0]}, e1 = {xj[1], xj[2], 0.f, 1.f};
                e[p][0] = pf_pairn(q == 0 ? e0 : z4, q == 0 ? e1 : z4);
            } else {
                const int4 j4 = *reinterpret_cast<const int4*>(a.idx + (size_t)g * 16 + 4 * q);
                vQ[p] = jc * ROWB + (S + 4 * q) * 4;
                vO[p][0] = (bN + j4.x) * ROWB + (S + 16 * NCONV + col) * 4;
                vO[p][1] = (bN + j4.y) * ROWB + (S + 16 * NCONV + col) * 4;
                vO[p][2] = (bN + j4.z) * ROWB + (S + 16 * NCONV + col) * 4;
                vO[p][3] = (bN + j4.w) * ROWB + (S + 16 * NCONV + col) * 4;
                // P[i]: S floats, one coalesced load per point (lanes beyond S / 2 idle), staged for broadcast reads
                if (lane * 2 < S) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const f2 pv = *reinterpret_cast<const f2*>(a.pq + (size_t)g * (2 * S) + lane * 2);
                    *reinterpret_cast<f2*>(&plds[wave * P + p][lane * 2]) = pv;
                }
            }
        }
        // growth pre-activation block t (this lane's 4 channels x its edge column), conv_out block (4 edges x its channel)
        f4 pre[P][NCONV];
        f4 outi[P][OBO];
        if constexpr (C3) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
#pragma unroll
                for (int b = 0; b < NCONV; ++b) pre[p][b] = pf_splat(0.f);
#pragma unroll
                for (int b = 0; b < OBO; ++b) outi[p][b] = pf_splat(0.f);
            }
            pf_mmn<false, NCONV, 1, 1>(ws, FT, e, pre);
            pf_mmn<true, OBO, 1, 1>(ws, FT + NCONV, e, outi);
        } else {
#pragma unroll
            for (int b = 0; b < NCONV; ++b)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    pre[p][b] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vQ[p], b * 64, 0));
#pragma unroll
            for (int b = 0; b < OBO; ++b)
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        outi[p][b][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vO[p][r], b * 64, 0));
#pragma unroll
            for (int b = 0; b < NCONV; ++b)
#pragma unroll
                for (int p = 0; p < P; ++p) pre[p][b] += *reinterpret_cast<const f4*>(&plds[wave * P + p][16 * b + 4 * q]);
        }

        PfPairN fp[P][2];
        f4 last[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            last[p] = pf_lrelu(pre[p][0], 0.05f);
            fp[p][0] = pf_pairn(last[p], pf_splat(0.f));
        }
        pf_static_for<1, NCONV>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int CPT = (t + 1) / 2, F0 = (t / 2) * ((t + 1) / 2);
            f4 acc[P][1];
#pragma unroll
            for (int p = 0; p < P; ++p) acc[p][0] = pre[p][t];
            pf_mmn<false, 1, CPT, CPT>(ws, F0, fp, acc);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const f4 f = pf_lrelu(acc[p][0], 0.05f);
                if constexpr (t % 2 == 1) fp[p][t / 2] = pf_pairn(last[p], f);
                else fp[p][t / 2] = pf_pairn(f, pf_splat(0.f));
                last[p] = f;
            }
        });
        pf_mmn<true, OBO, 2, 2>(ws, FO, fp, outi);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float m[OBO];
#pragma unroll
            for (int b = 0; b < OBO; ++b) m[b] = fmaxf(fmaxf(outi[p][b].x, outi[p][b].y), fmaxf(outi[p][b].z, outi[p][b].w));
            float o;
            int ch;
            if constexpr (OBO == 4) {           // lane row q ends up with block {0, 2, 1, 3}[q]
                o = pf_rsmax16(pf_rsmax32(m[0], m[1]), pf_rsmax32(m[2], m[3]));
                ch = 16 * (((q & 1) << 1) | (q >> 1)) + col;
            } else {                            // rows 0,1 hold block 0, rows 2,3 block 1 (both copies complete)
                const float n = pf_rsmax32(m[0], m[1]);
                o = pf_rsmax16(n, n);
                ch = 16 * (q >> 1) + col;
            }
            float pv = 0.f;
            if constexpr (!C3) pv = plds[wave * P + p][16 * NCONV + ch];
            const float hv = fmaf(o, EC4_OUT_INV, pv);
            if (pt0 + p < a.T && (OBO == 4 || (q & 1) == 0))
                a.out[(size_t)gi[p] * ODIM + ch] = hv;
            if constexpr (PQF) {
                if (OBO == 4 || (q & 1) == 0) pqf_stage<ODIM>(hb + pqf.nst * ((ODIM / 32) * 2 * 64), ch, wave * P + p, hv);
            }
        }
        if constexpr (PQF) {
            pqf.push(tile * NW * P);
            if (pqf.nst == PQF_NT) {                 // uniform: flush (the tail is flushed after the loop)
                __syncthreads();
                pqf_gemm<ODIM, PQ_ROWS, NW>(a, hb, pqf.nst, pqf.tpt, __builtin_amdgcn_readfirstlane(wave), lane);
                __syncthreads();
                pqf.nst = 0;
            }
        }
    }
    if constexpr (PQF) {
        if (pqf.nst > 0) {
            __syncthreads();
            pqf_gemm<ODIM, PQ_ROWS, NW>(a, hb, pqf.nst, pqf.tpt, __builtin_amdgcn_readfirstlane(wave), lane);
        }
    }
}

template <int ODIM, bool C3, int P, int NW, bool PQF = false>
int launch1n(const EcArgs& a0, hipStream_t s) {
    EcArgs a = a0;
    a.ntiles = (a.T + NW * P - 1) / (NW * P);
    a.chunk = (a.ntiles + 7) / 8;
    int grid = 8 * a.chunk;
    const int cap = 256 * (32 / NW);                      // persistent: resident workgroups only (LDS < 40 KiB each)
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL((edgeconv1n_kernel<ODIM, C3, P, NW, PQF>), dim3(grid), dim3(NW * 64), 0, s, a);
    return pf_last_launch_status();
}

// Launch shapes (points per wave P, waves per workgroup NW).  The default build carries ONE shape per kernel - the shipped
// one (tools/tune_edgeconv.py / tune_ec4.py sweeps on MI355X); -DPF_TUNING_VARIANTS adds the alternatives and the timing-only
// ablation instantiations (DBG != 0: wrong results) for the tuning tools.
template <int ODIM, bool C3>
int launch1n_v(const EcArgs& a, hipStream_t s, int variant) {
    switch (variant) {
        case 1: return launch1n<ODIM, C3, 2, 8>(a, s);                  // shipped
#ifdef PF_TUNING_VARIANTS
        case 0: return launch1n<ODIM, C3, 1, 8>(a, s);
        case 2: return launch1n<ODIM, C3, 1, 16>(a, s);
        case 3: return launch1n<ODIM, C3, 2, 4>(a, s);
#endif
        default: return PF_ERR_UNSUPPORTED;
    }
}

template <int GB, int NCONV, int ODIM, bool C3, int P, int NW>
int launch_v(const EcArgs& a0, hipStream_t s) {
    EcArgs a = a0;
    a.ntiles = (a.T + NW * P - 1) / (NW * P);
    a.chunk = (a.ntiles + 7) / 8;
    constexpr int NF = GB * NCONV;
    constexpr int lds = (GB * GB * (NCONV * (NCONV - 1) / 2) + (ODIM / 16) * NF) * 1024;
    int per_cu = (160 * 1024) / lds;                       // workgroups that fit one CU's LDS
    const int by_waves = 32 / NW;                          // 32 waves per CU
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu < 1) per_cu = 1;
    int grid = 8 * a.chunk;
    const int cap = 256 * per_cu;                          // persistent: resident workgroups only
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL((edgeconv_kernel<GB, NCONV, ODIM, C3, P, NW>), dim3(grid), dim3(NW * 64), 0, s, a);
    return pf_last_launch_status();
}

// the exact-fp32 reference kernel: shipped shape of unit class GB/ODIM = (1, 32): (2, 8); (1, 64): (1, 8); (2, 128): (1, 16)
template <int GB, int NCONV, int ODIM, bool C3>
int launch(const EcArgs& a, hipStream_t s, int variant) {
    constexpr int BEST = ODIM == 32 ? 0 : (ODIM == 64 ? 2 : 3);
    if (variant == BEST) {
        if constexpr (BEST == 0) return launch_v<GB, NCONV, ODIM, C3, 2, 8>(a, s);
        else if constexpr (BEST == 2) return launch_v<GB, NCONV, ODIM, C3, 1, 8>(a, s);
        else return launch_v<GB, NCONV, ODIM, C3, 1, 16>(a, s);
    }
#ifdef PF_TUNING_VARIANTS
    switch (variant) {
        case 0: return launch_v<GB, NCONV, ODIM, C3, 2, 8>(a, s);
        case 1: return launch_v<GB, NCONV, ODIM, C3, 2, 4>(a, s);
        case 2: return launch_v<GB, NCONV, ODIM, C3, 1, 8>(a, s);
        case 3: return launch_v<GB, NCONV, ODIM, C3, 1, 16>(a, s);
        default: break;
    }
#endif
    return PF_ERR_UNSUPPORTED;
}

}  // namespace

// cfg 0 / 1 / 2: the exact-fp32 kernel (v_mfma_f32_16x16x4_f32; the in-library A/B reference) on unit 0 (C=3, g=8 padded to 16,
//                odim 32; needs `tab`), unit 1 (g=16, odim 64), units 2..5 (g=32, odim 128); UNSCALED P|Q table
// cfg 7 / 8 / 9: the product arithmetic (split-fp16 with a natural-scale low half): units 2..5 (packing: ec4_w), unit 0
//                (ec1n_w[0], edge table inside wfrag), unit 1 (ec1n_w[1]); SCALED P|Q table (packing.ec4_scales)
// (cfg 3..6 were the round-1 split-bf16 / scaled split-fp16 generations: removed, see the git history)
// variant: launch shape; pf_edgeconv() passes the shipped one.
extern "C" int pf_edgeconv_tuned(int cfg, int variant, const float* pq_or_xyz, const float* tab, const int* idx,
                                 const float* wfrag, float* out, int B, int N, void* stream) {
    if (!pq_or_xyz || !idx || !wfrag || !out) return PF_ERR_NULL;
    if (B <= 0 || N < 16 || (long long)B * N > (1ll << 30)) return PF_ERR_SHAPE;
    EcArgs a{};
    a.idx = idx; a.wg = reinterpret_cast<const f4*>(wfrag); a.out = out; a.T = B * N; a.N = N;
    hipStream_t s = (hipStream_t)stream;
    switch (cfg) {
        case 0:
            if (!tab) return PF_ERR_NULL;
            a.xyz = pq_or_xyz; a.tab = tab;
            return launch<1, 4, 32, true>(a, s, variant);
        case 1: a.pq = pq_or_xyz; return launch<1, 4, 64, false>(a, s, variant);
        case 2: a.pq = pq_or_xyz; return launch<2, 4, 128, false>(a, s, variant);
        case 7:
            {
                // the kernel addresses the P|Q table with 32-bit byte offsets: whole batch items per launch, < 2 GiB of table
                const long long maxT = 0x7fffffffll / 2048;
                if (N > maxT) return PF_ERR_UNSUPPORTED;
                const int Bc = (int)(maxT / N);
                for (int b0 = 0; b0 < B; b0 += Bc) {
                    const int nb = B - b0 < Bc ? B - b0 : Bc;
                    EcArgs c = a;
                    c.pq = pq_or_xyz + (size_t)b0 * N * 512; c.idx = idx + (size_t)b0 * N * 16; c.out = out + (size_t)b0 * N * 128;
                    c.T = nb * N;
                    int rc;
                    switch (variant) {
                        case 0: rc = launch4<1, 16>(c, s); break;          // shipped
#ifdef PF_TUNING_VARIANTS
                        case 1: rc = launch4<2, 8>(c, s); break;
                        case 2: rc = launch4<1, 8>(c, s); break;
                        case 3: rc = launch4<2, 4>(c, s); break;
                        case 8: rc = launch4<1, 16, 1>(c, s); break;      // no gathers
                        case 9: rc = launch4<1, 16, 2>(c, s); break;      // no MFMAs
                        case 10: rc = launch4<1, 16, 4>(c, s); break;     // no LDS weight reads
                        case 11: rc = launch4<1, 16, 5>(c, s); break;     // no gathers, no LDS weight reads
                        case 12: rc = launch4<1, 16, 3>(c, s); break;     // no gathers, no MFMAs
                        case 13: rc = launch4<1, 16, 6>(c, s); break;     // no MFMAs, no LDS reads: gathers + VALU only
                        case 14: rc = launch4<1, 16, 13>(c, s); break;    // no gathers, no LDS reads, no growth epilogues (lrelu + split)
#endif
                        default: return PF_ERR_UNSUPPORTED;
                    }
                    if (rc != PF_OK) return rc;
                }
                return PF_OK;
            }
        case 8:
            a.xyz = pq_or_xyz;
            if ((long long)B * N * 12 > 0x7fffffffll) return PF_ERR_SHAPE;
            return launch1n_v<32, true>(a, s, variant);
        case 9:
            a.pq = pq_or_xyz;
            if ((long long)B * N * 1024 > 0x7fffffffll) return PF_ERR_SHAPE;      // 32-bit buffer offsets
            return launch1n_v<64, false>(a, s, variant);
        default: return PF_ERR_UNSUPPORTED;
    }
}

extern "C" int pf_edgeconv(int cfg, const float* pq_or_xyz, const float* tab, const int* idx, const float* wfrag,
                           float* out, int B, int N, void* stream) {
    // shipped launch shapes: f32 reference (2,8) / (1,8) / (1,16); f16n units 0 / 1 -> (P=2, NW=8); units 2..5 -> (1, 16): one
    // point per wave, 16 waves share the 88 KiB of LDS-resident weights
    static const int best[10] = {0, 2, 3, -1, -1, -1, -1, 0, 1, 1};
    if (cfg < 0 || cfg > 9 || best[cfg] < 0) return PF_ERR_UNSUPPORTED;
    return pf_edgeconv_tuned(cfg, best[cfg], pq_or_xyz, tab, idx, wfrag, out, B, N, stream);
}

// EdgeConv unit `unit` (0..4) in the product arithmetic AND the next unit's P|Q vectors: out [B*N, odim] as pf_edgeconv (cfg 8 /
// 9 / 7), pq_next [B*N, rows] as pf_pq_gemm(unit, out, ...) - bit-identical to that pair of calls.  Small batches run both in
// ONE launch (the fused epilogue above: a 16-point workgroup tile is one MFMA column tile of the GEMM); larger ones as the two
// kernels (the P|Q GEMM is HBM-write-bound there and its weights would compete with the Q gathers for L2 bandwidth).
// w: blob base, off[13]: POST_SLOTS of unit `unit` (packing.py).
extern "C" int pf_pq_gemm(int unit, const float* h, const float* w, const long long* off, float* pq_next, int T, void* stream);

extern "C" int pf_edgeconv_pq(int unit, const float* pq_or_xyz, const int* idx, const float* wfrag, float* out, const float* w,
                              const long long* off, float* pq_next, int B, int N, int fuse, void* stream) {
    if (!pq_or_xyz || !idx || !wfrag || !out || !w || !off || !pq_next) return PF_ERR_NULL;
    if (unit < 0 || unit > 4) return PF_ERR_UNSUPPORTED;
    if (B <= 0 || N < 16 || (long long)B * N > (1ll << 30)) return PF_ERR_SHAPE;
    const long long T = (long long)B * N;
    // fuse: 0 = never, 1 = always (when the shape allows), -1 = by size: up to 16 384 points (8 tiles per workgroup)
    const bool fits32 = unit == 0 ? T * 12 <= 0x7fffffffll : (unit == 1 ? T * 1024 <= 0x7fffffffll : T * 2048 <= 0x7fffffffll);
    const bool fused = fits32 && (fuse == 1 || (fuse < 0 && T <= 16384));
    const int cfg = unit == 0 ? 8 : (unit == 1 ? 9 : 7);
    if (!fused) {
        const int rc = pf_edgeconv(cfg, pq_or_xyz, nullptr, idx, wfrag, out, B, N, stream);
        return rc != PF_OK ? rc : pf_pq_gemm(unit, out, w, off, pq_next, (int)T, stream);
    }
    EcArgs a{};
    a.idx = idx; a.wg = reinterpret_cast<const f4*>(wfrag); a.out = out; a.T = (int)T; a.N = N;
    a.pqf_w = reinterpret_cast<const u4*>(w + off[10]); a.pqf_bias = w + off[11]; a.pqf_scales = w + off[12]; a.pqf_out = pq_next;
    hipStream_t s = (hipStream_t)stream;
    if (unit == 0) { a.xyz = pq_or_xyz; return launch1n<32, true, 2, 8, true>(a, s); }
    a.pq = pq_or_xyz;
    if (unit == 1) return launch1n<64, false, 2, 8, true>(a, s);
    return launch4<1, 16, 0, true>(a, s);
}
