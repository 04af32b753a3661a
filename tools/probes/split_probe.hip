// Functional probe for the instruction idioms of the split-fp16 EdgeConv kernel (csrc/edgeconv.hip, edgeconv4_kernel):
//   1. hi = v_cvt_pk_f16_f32 (RNE), lo = v_fma_mixlo/mixhi_f16(hi, -1.0, x): lo = fp16(x - hi) incl. SUBNORMAL results
//   2. v_permlane32_swap / v_permlane16_swap lane semantics (inline asm form used by the kernel)
//   3. an activation fragment is both the B operand (edges on columns) and the A operand (edges on rows) of
//      v_mfma_f32_16x16x32_f16: D_swapped = D^T
// Build: hipcc --offload-arch=gfx950 -O3 split_probe.hip -o split_probe ; prints PASS/FAIL per item.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_split(const float* in, unsigned* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = in[2 * i], b = in[2 * i + 1];
    unsigned hb = __builtin_bit_cast(unsigned, (h2){(_Float16)a, (_Float16)b});
    unsigned lb;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hb), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lb) : "v"(hb), "v"(b));
    out[2 * i] = hb;
    out[2 * i + 1] = lb;
}

__global__ void k_swap(unsigned* out) {
    const unsigned l = threadIdx.x;
    unsigned a = 0x100 + l, b = 0x200 + l, c = 0x300 + l, d = 0x400 + l;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));
    out[l] = a; out[64 + l] = b; out[128 + l] = c; out[192 + l] = d;
}

// act [16 edges][32 k], w [16 ch][32 k] (fp16-exact small integers): D1[ch][edge] = mfma(A = w, B = act), D2[edge][ch] = mfma(A = act, B = w)
__global__ void k_mfma(const float* act, const float* w, float* d1, float* d2) {
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    h8 fa, fw;
    for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)act[r * 32 + 8 * q + j]; fw[j] = (_Float16)w[r * 32 + 8 * q + j]; }
    f4 z = {0, 0, 0, 0};
    const f4 a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw, fa, z, 0, 0, 0);
    const f4 a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fw, z, 0, 0, 0);
    for (int k = 0; k < 4; ++k) { d1[(4 * q + k) * 16 + r] = a1[k]; d2[(4 * q + k) * 16 + r] = a2[k]; }
}

static float h2f(unsigned short h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; }

int main() {
    int fails = 0;
    {   // 1. split
        std::vector<float> v;
        for (int e = -20; e <= 12; ++e)
            for (int m = 0; m < 64; ++m) { float x = ldexpf(1.f + m * 0.0153931f + 1e-4f * e, e); v.push_back(x); v.push_back(-x * 0.731f); }
        v.push_back(0.f); v.push_back(-0.f); v.push_back(6.1e-5f); v.push_back(3.0e-8f);
        const int n = (int)v.size();
        float* di; unsigned* dout;
        (void)hipMalloc(&di, n * 4); (void)hipMalloc(&dout, n * 4);
        (void)hipMemcpy(di, v.data(), n * 4, hipMemcpyHostToDevice);
        k_split<<<(n / 2 + 63) / 64, 64>>>(di, dout, n);
        std::vector<unsigned> o(n);
        (void)hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
        int bad = 0, subn = 0; double worst = 0;
        for (int i = 0; i + 1 < n; i += 2)
            for (int s = 0; s < 2; ++s) {
                const float x = v[i + s];
                const unsigned short hb = (o[i] >> (16 * s)) & 0xffff, lb = (o[i + 1] >> (16 * s)) & 0xffff;
                const _Float16 eh = (_Float16)x;
                const _Float16 el = (_Float16)(x - (float)eh);
                unsigned short ehb, elb; memcpy(&ehb, &eh, 2); memcpy(&elb, &el, 2);
                if (ehb != hb || elb != lb) { if (bad < 5) printf("  split mismatch x=%g hi %04x/%04x lo %04x/%04x\n", x, hb, ehb, lb, elb); ++bad; }
                if ((elb & 0x7c00) == 0 && (elb & 0x3ff)) ++subn;
                const double rec = (double)h2f(hb) + (double)h2f(lb);
                if (x != 0) worst = fmax(worst, fabs(rec - x));
            }
        printf("1. split hi/lo (RNE, fma_mix, %d subnormal lo values kept): %s   max |hi+lo-x| = %.3g\n", subn, bad ? "FAIL" : "PASS", worst);
        fails += bad != 0;
    }
    {   // 2. swaps
        unsigned* d; (void)hipMalloc(&d, 256 * 4);
        k_swap<<<1, 64>>>(d);
        unsigned o[256]; (void)hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost);
        int bad = 0;
        for (unsigned l = 0; l < 64; ++l) {
            // permlane32_swap: vdst[32..63] <-> vsrc[0..31]
            const unsigned ea = l < 32 ? 0x100 + l : 0x200 + (l - 32), eb = l < 32 ? 0x100 + (l + 32) : 0x200 + l;
            // permlane16_swap: vdst odd rows <-> vsrc even rows
            const unsigned row = l >> 4;
            const unsigned ec = (row & 1) ? 0x400 + (l - 16) : 0x300 + l, ed = (row & 1) ? 0x400 + l : 0x300 + (l + 16);
            if (o[l] != ea || o[64 + l] != eb || o[128 + l] != ec || o[192 + l] != ed) {
                if (bad < 8) printf("  lane %u: a %x/%x b %x/%x c %x/%x d %x/%x\n", l, o[l], ea, o[64 + l], eb, o[128 + l], ec, o[192 + l], ed);
                ++bad;
            }
        }
        printf("2. permlane32_swap / permlane16_swap lane maps: %s\n", bad ? "FAIL" : "PASS");
        fails += bad != 0;
    }
    {   // 3. transposed MFMA
        float act[512], w[512], e1[256], o1[256], o2[256];
        for (int i = 0; i < 512; ++i) { act[i] = (float)((i * 7 + 3) % 11 - 5); w[i] = (float)((i * 5 + 1) % 13 - 6); }
        for (int c = 0; c < 16; ++c)
            for (int e = 0; e < 16; ++e) { float s = 0; for (int k = 0; k < 32; ++k) s += w[c * 32 + k] * act[e * 32 + k]; e1[c * 16 + e] = s; }
        float *da, *dw, *d1, *d2;
        (void)hipMalloc(&da, 2048); (void)hipMalloc(&dw, 2048); (void)hipMalloc(&d1, 1024); (void)hipMalloc(&d2, 1024);
        (void)hipMemcpy(da, act, 2048, hipMemcpyHostToDevice); (void)hipMemcpy(dw, w, 2048, hipMemcpyHostToDevice);
        k_mfma<<<1, 64>>>(da, dw, d1, d2);
        (void)hipMemcpy(o1, d1, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(o2, d2, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int c = 0; c < 16; ++c)
            for (int e = 0; e < 16; ++e) { if (o1[c * 16 + e] != e1[c * 16 + e]) ++bad; if (o2[e * 16 + c] != e1[c * 16 + e]) ++bad; }
        printf("3. activation fragment as B (D[ch][edge]) and as A (D[edge][ch]): %s\n", bad ? "FAIL" : "PASS");
        fails += bad != 0;
    }
    printf(fails ? "PROBE FAILED\n" : "PROBE OK\n");
    return fails;
}
