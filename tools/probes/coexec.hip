// Micro-benchmark (hipcc --offload-arch=gfx950 -O3 coexec.hip -o coexec): do VALU and MFMA instructions of one SIMD overlap?
// Result on MI355X (2 waves per SIMD, cycles per iteration at a nominal 2.4 GHz): 4 MFMA 16x16x32 f16 alone 104, 12 v_fma alone 70,
// both on DIFFERENT waves 151, interleaved in the SAME waves 274 for twice the work - nearly additive: a SIMD issues either
// a vector or a matrix instruction, the matrix pipe does not run in the shadow of another wave's VALU work.  See DESIGN.md 4.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
// mode bit0: waves 0..3 run MFMA loop; bit1: waves 4..7 run VALU loop; bit2: every wave alternates MFMA and VALU (same wave)
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
    const int wave = threadIdx.x >> 6;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
    f4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0+4, v5=v0+5, v6=v0+6, v7=v0+7;
    if (mode & 4) {
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            v3 = fmaf(v3, 1.0001f, 0.5f); v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
            v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f); v0 = fmaf(v0, 1.0001f, 0.5f);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
            v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
        }
    } else if (wave < 4) {
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
            }
    } else {
        if (mode & 2)
            for (int it = 0; it < iters; ++it) {
                v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
                v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
            }
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0.x + c1.y + c2.z + c3.w + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[] = {"", "mfma only (waves 0-3: 4 MFMA/iter)", "valu only (waves 4-7: 12 VALU/iter)", "both, different waves", "same wave: 4 MFMA + 12 VALU interleaved"};
    int modes[] = {1, 2, 3, 4};
    for (int m : modes) {
        k<<<256, 512>>>(m, 100, d);
        (void)hipEventRecord(e0);
        k<<<256, 512>>>(m, iters, d);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d %-42s %.3f ms  -> %.1f cycles/iter @2.4GHz\n", m, names[m], ms, ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
