// Issue-model probe for one gfx950 SIMD, in SHADER CYCLES (s_memtime), not wall time / nominal clock:
//   * cycles per MFMA (16x16x32 f16 and 32x32x16 f16) with n independent v_fma_f32 placed in every MFMA gap of the SAME wave
//   * the same work with the VALU stream on a DIFFERENT wave of the same SIMD
// Build: hipcc --offload-arch=gfx950 -O3 issue_model.hip -o issue_model ; run on the GPU box (tools/README.md).
// The loop bodies are inline asm so that hipcc cannot re-order or fuse them.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define VFMA(r) "v_fma_f32 %" #r ", %" #r ", %[k1], %[k2]\n\t"

template <int SHAPE, int NV, int NACC = 4>   // SHAPE 16 or 32; NV VALU per MFMA gap (0..8); NACC accumulators in rotation
__device__ __forceinline__ void body(h8 a, h8 b, f4 (&c)[4], f16v (&d)[2], float (&v)[8], float k1, float k2, int iters) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if constexpr (SHAPE == 16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[m % NACC]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d[m & 1]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(k1), "v"(k2));
        }
    }
}

// role: 0 = every wave runs MFMA+NV interleaved; 1 = even-numbered waves-on-SIMD run MFMA only, the others VALU only (4*NV per iter)
template <int SHAPE, int NV, int ROLE, int NACC = 4>
__global__ __launch_bounds__(1024) void k(int iters, float* out, long long* cyc, long long* rt) {
    const int wave = threadIdx.x >> 6;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)((threadIdx.x % 37) * 0.01f); b[i] = (_Float16)((i + threadIdx.x % 5) * 0.02f); }
    f4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f16v d[2];
    for (int i = 0; i < 16; ++i) { d[0][i] = 0; d[1][i] = 0; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    const float k1 = 1.0001f, k2 = 0.5f;
    __syncthreads();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (ROLE == 0) body<SHAPE, NV, NACC>(a, b, c, d, v, k1, k2, iters);
    else {
        // waves are dealt to SIMDs cyclically; waves w and w+4 share a SIMD in an 8-wave workgroup
        if (wave < 4) body<SHAPE, 0>(a, b, c, d, v, k1, k2, iters);
        else {
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int j = 0; j < 4 * NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(k1), "v"(k2));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += c[i].x;
    for (int i = 0; i < 8; ++i) s += v[i];
    s += d[0][0] + d[1][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0; rt[blockIdx.x * (blockDim.x >> 6) + wave] = r1 - r0; }
}

template <int SHAPE, int NV, int ROLE, int NACC = 4>
void run(const char* what, int waves, float* d, long long* dc) {
    const int iters = 4000, grid = 256;
    long long* dr = dc + 256 * 16;
    k<SHAPE, NV, ROLE, NACC><<<grid, waves * 64>>>(100, d, dc, dr);
    k<SHAPE, NV, ROLE, NACC><<<grid, waves * 64>>>(iters, d, dc, dr);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(grid * waves), hr(grid * waves);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hr.data(), dr, hr.size() * 8, hipMemcpyDeviceToHost);
    // per block: the LAST wave of each role to finish (older waves win the arbitration and finish early); median over blocks.
    // ROLE 1: MFMA waves are wave < 4 of each block, VALU waves the rest
    std::vector<long long> A, B;
    std::vector<double> clk;
    for (int bidx = 0; bidx < grid; ++bidx) {
        long long ma = 0, mb = 0;
        for (int w = 0; w < waves; ++w) {
            long long& m = (ROLE == 1 && w >= 4) ? mb : ma;
            m = std::max(m, h[bidx * waves + w]);
            clk.push_back(100.0 * h[bidx * waves + w] / (double)hr[bidx * waves + w]);
        }
        A.push_back(ma);
        if (mb) B.push_back(mb);
    }
    std::sort(A.begin(), A.end());
    std::sort(B.begin(), B.end());
    std::sort(clk.begin(), clk.end());
    const int per_simd = ROLE == 1 ? 1 : waves / 4;
    const double perA = (double)A[A.size() / 2] / (iters * 4.0) / per_simd;
    printf("%-44s shape %2d  valu/gap %d  waves/WG %2d : %6.2f cycles per MFMA on the SIMD  [%4.0f MHz]", what, SHAPE, NV, waves, perA, clk[clk.size() / 2]);
    if (!B.empty()) printf("   (VALU-only waves: %.2f cycles per %d VALU)", (double)B[B.size() / 2] / (iters * 4.0), NV);
    printf("\n");
}

int main() {
    float* d; long long* dc;
    (void)hipMalloc(&d, 256 * 1024 * 4);
    (void)hipMalloc(&dc, 2 * 256 * 16 * 8);
#define SWEEP(S)                                                                     \
    run<S, 0, 0>("same wave, 1 wave/SIMD", 4, d, dc);                                \
    run<S, 1, 0>("same wave, 1 wave/SIMD", 4, d, dc);                                \
    run<S, 2, 0>("same wave, 1 wave/SIMD", 4, d, dc);                                \
    run<S, 3, 0>("same wave, 1 wave/SIMD", 4, d, dc);                                \
    run<S, 4, 0>("same wave, 1 wave/SIMD", 4, d, dc);                                \
    run<S, 6, 0>("same wave, 1 wave/SIMD", 4, d, dc);                                \
    run<S, 8, 0>("same wave, 1 wave/SIMD", 4, d, dc);                                \
    run<S, 0, 0>("same wave, 2 waves/SIMD", 8, d, dc);    \
    run<S, 2, 0>("same wave, 2 waves/SIMD", 8, d, dc);    \
    run<S, 4, 0>("same wave, 2 waves/SIMD", 8, d, dc);    \
    run<S, 0, 0>("same wave, 4 waves/SIMD", 16, d, dc);   \
    run<S, 2, 0>("same wave, 4 waves/SIMD", 16, d, dc);   \
    run<S, 4, 0>("same wave, 4 waves/SIMD", 16, d, dc);   \
    run<S, 2, 1>("MFMA wave | VALU wave on one SIMD", 8, d, dc);   \
    run<S, 4, 1>("MFMA wave | VALU wave on one SIMD", 8, d, dc);   \
    run<S, 6, 1>("MFMA wave | VALU wave on one SIMD", 8, d, dc);
    SWEEP(16)
    SWEEP(32)
    // dependent accumulator chains of the 16x16x32 shape (every MFMA takes the previous one's D as C)
    run<16, 0, 0, 1>("ONE accumulator chain, 1 wave/SIMD", 4, d, dc);
    run<16, 0, 0, 2>("TWO accumulators alternating, 1 wave/SIMD", 4, d, dc);
    run<16, 2, 0, 1>("ONE accumulator chain, 1 wave/SIMD", 4, d, dc);
    run<16, 0, 0, 1>("ONE accumulator chain, 2 waves/SIMD", 8, d, dc);
    run<16, 0, 0, 1>("ONE accumulator chain, 4 waves/SIMD", 16, d, dc);
    run<16, 2, 0, 1>("ONE accumulator chain, 4 waves/SIMD", 16, d, dc);
    return 0;
}
