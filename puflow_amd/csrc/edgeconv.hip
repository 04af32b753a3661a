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

        f4 feat[P][NF];
        // layer 0: edge part only
#pragma unroll
        for (int ob = 0; ob < GB; ++ob)
#pragma unroll
            for (int p = 0; p < P; ++p) feat[p][ob] = pf_lrelu(init(p, ob * 16 + 4 * q), 0.05f);

        // growth layers t = 1..NCONV-1: inputs = feature blocks [0, GB*t)
        pf_static_for<1, NCONV>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            f4 acc[P][GB];
#pragma unroll
            for (int ob = 0; ob < GB; ++ob)
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p][ob] = init(p, G * t + ob * 16 + 4 * q);
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
            constexpr int ob0 = decltype(cc)::value * OCH;
            f4 acc[P][OCH];
#pragma unroll
            for (int o = 0; o < OCH; ++o)
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p][o] = init(p, G * NCONV + (ob0 + o) * 16 + 4 * q);
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

template <int GB, int NCONV, int ODIM, bool C3>
int launch(const EcArgs& a0, hipStream_t s) {
    constexpr int P = 2, NW = 4;
    EcArgs a = a0;
    a.ntiles = (a.T + NW * P - 1) / (NW * P);
    a.chunk = (a.ntiles + 7) / 8;
    int grid = 8 * a.chunk;
    const int cap = 256 * 8;                       // persistent cap: 8 workgroups of 256 threads per CU
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL((edgeconv_kernel<GB, NCONV, ODIM, C3, P, NW>), dim3(grid), dim3(NW * 64), 0, s, a);
    return pf_last_launch_status();
}

}  // namespace

// cfg: 0 = unit 0 (C=3, g=8 padded to 16, 4 convs, odim 32; C3 table variant)
//      1 = unit 1 (g=16, odim 64)    2 = units 2..5 (g=32, odim 128)
extern "C" int pf_edgeconv(int cfg, const float* pq_or_xyz, const float* tab, const int* idx, const float* wfrag,
                           float* out, int B, int N, void* stream) {
    if (!pq_or_xyz || !idx || !wfrag || !out) return PF_ERR_NULL;
    if (B <= 0 || N < 16 || (long long)B * N > (1ll << 30)) return PF_ERR_SHAPE;
    EcArgs a{};
    a.idx = idx; a.wg = reinterpret_cast<const f4*>(wfrag); a.out = out; a.T = B * N; a.N = N;
    hipStream_t s = (hipStream_t)stream;
    switch (cfg) {
        case 0:
            if (!tab) return PF_ERR_NULL;
            a.xyz = pq_or_xyz; a.tab = tab;
            return launch<1, 4, 32, true>(a, s);
        case 1: a.pq = pq_or_xyz; return launch<1, 4, 64, false>(a, s);
        case 2: a.pq = pq_or_xyz; return launch<2, 4, 128, false>(a, s);
        default: return PF_ERR_UNSUPPORTED;
    }
}
