// Probe: how much of the MFMA rate survives when every MFMA triple takes its A operands from LDS (the weight stream of
// edgeconv4_kernel: two ds_read_b128 per three v_mfma_f32_16x16x32_f16), 16 waves per CU, one workgroup per CU.
//   MPF = MFMAs per weight fragment pair (3 = one point per wave, 6 = two points share a fragment, 12 = four)
// Reports shader cycles per MFMA on a SIMD (last wave of a block to finish) and the clock held.
// Build: hipcc --offload-arch=gfx950 -O3 lds_mfma.hip -o lds_mfma
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int NFRAG = 44;

template <int MPF, int DEPTH, bool USE_LDS>
__global__ __launch_bounds__(1024) void k(int iters, float* out, long long* cyc, long long* rt) {
    __shared__ u4 w[NFRAG * 2 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < NFRAG * 2 * 64; i += blockDim.x) w[i] = (u4){0x3c003c00u + i, 0x38003800u, 0x34003400u, 0x30003000u};
    __syncthreads();
    h8 b[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) b[j][i] = (_Float16)(0.01f * ((threadIdx.x + i + j) % 13));
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        h8 wb[DEPTH][2];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                wb[i][s] = USE_LDS ? __builtin_bit_cast(h8, w[(i * 2 + s) * 64 + lane]) : b[(i + s) & 3];
#pragma unroll
        for (int i = 0; i < NFRAG; ++i) {
            const h8 wh = wb[i % DEPTH][0], wl = wb[i % DEPTH][1];
            if (i + DEPTH < NFRAG) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    wb[i % DEPTH][s] = USE_LDS ? __builtin_bit_cast(h8, w[((i + DEPTH) * 2 + s) * 64 + lane]) : b[(i + s + 1) & 3];
            }
#pragma unroll
            for (int m = 0; m < MPF / 3; ++m) {
                f4 x = acc[(i + m) & 3];
                x = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, b[m & 3], x, 0, 0, 0);
                x = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, b[(m + 1) & 3], x, 0, 0, 0);
                x = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, b[(m + 2) & 3], x, 0, 0, 0);
                acc[(i + m) & 3] = x;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0].x + acc[1].y + acc[2].z + acc[3].w;
    if (lane == 0) { cyc[blockIdx.x * 16 + wave] = t1 - t0; rt[blockIdx.x * 16 + wave] = r1 - r0; }
}

template <int MPF, int DEPTH, bool USE_LDS>
void run(const char* what, int waves, float* d, long long* dc) {
    const int iters = 200 * 3 / MPF, grid = 256;
    long long* dr = dc + 256 * 16;
    k<MPF, DEPTH, USE_LDS><<<grid, waves * 64>>>(4, d, dc, dr);
    k<MPF, DEPTH, USE_LDS><<<grid, waves * 64>>>(iters, d, dc, dr);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(grid * 16), hr(grid * 16);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hr.data(), dr, hr.size() * 8, hipMemcpyDeviceToHost);
    std::vector<long long> A;
    std::vector<double> clk;
    for (int bidx = 0; bidx < grid; ++bidx) {
        long long ma = 0;
        for (int wv = 0; wv < waves; ++wv) { ma = std::max(ma, h[bidx * 16 + wv]); clk.push_back(100.0 * h[bidx * 16 + wv] / (double)hr[bidx * 16 + wv]); }
        A.push_back(ma);
    }
    std::sort(A.begin(), A.end());
    std::sort(clk.begin(), clk.end());
    const double n_mfma_simd = (double)iters * NFRAG * MPF * (waves / 4);
    printf("%-40s MFMA/frag %2d depth %d waves/CU %2d : %6.2f cycles per MFMA on the SIMD  [%4.0f MHz]\n", what, MPF, DEPTH, waves,
           A[A.size() / 2] / n_mfma_simd, clk[clk.size() / 2]);
}

int main() {
    float* d; long long* dc;
    (void)hipMalloc(&d, 256 * 1024 * 4);
    (void)hipMalloc(&dc, 2 * 256 * 16 * 8);
    run<3, 2, false>("operands in registers", 16, d, dc);
    run<3, 2, true>("weights from LDS", 16, d, dc);
    run<3, 4, true>("weights from LDS", 16, d, dc);
    run<6, 2, true>("weights from LDS", 16, d, dc);
    run<12, 2, true>("weights from LDS", 16, d, dc);
    run<3, 2, true>("weights from LDS", 8, d, dc);
    run<6, 2, true>("weights from LDS", 8, d, dc);
    run<3, 2, true>("weights from LDS", 4, d, dc);
    run<6, 2, true>("weights from LDS", 4, d, dc);
    return 0;
}
