// Probe (hipcc --offload-arch=gfx950 -O2): does v_mfma_f32_16x16x32_f16 honour fp16 subnormal operands?  MI355X: yes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float av, float bv, float* out) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c.x; out[1] = (float)a[0]; }
}
int main() {
    float* d; hipMalloc(&d, 8);
    float tests[][2] = {{ldexpf(1, -20), 1.f}, {ldexpf(1, -24), 1.f}, {1.f, ldexpf(1, -20)}, {ldexpf(1,-14), 1.f}, {ldexpf(1,-15), ldexpf(1,-15)}};
    for (auto& t : tests) {
        k<<<1, 64>>>(t[0], t[1], d);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a=%g b=%g  mfma=%g expect=%g  cvt(a)=%g\n", t[0], t[1], h[0], 32.0 * t[0] * t[1], h[1]);
    }
    return 0;
}
