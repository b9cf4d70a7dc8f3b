// Probe: what limits a [ds_read fragments -> s_waitcnt -> 4x v_mfma_f32_32x32x2_f32] loop on MI355X?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_loop_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// Variants (template V): 0 = MFMA only; 1 = 4 ds_read_b32 + wait + 4 MFMA (the conv kernels' k-pair);
// 2 = same, reads of the NEXT k-pair issued before the MFMAs (register double buffering);
// 3 = like 1 but TWO k-pairs per wait (8 reads, 8 MFMAs);  4 = like 2 with two k-pairs per stage.
// Occupancy is set by padding the static LDS (1, 2 or 4 workgroups of 4 waves per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V, int LDS_KB>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    __shared__ float lds[LDS_KB * 256];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5, wave = tid >> 6;
    for (int i = tid; i < LDS_KB * 256; i += 256) lds[i] = (float)(i & 15) * 0.001f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* ap = lds + ((wave >> 1) * 64 + l31) * 33 + h;
    const float* bp = lds + 4224 + ((wave & 1) * 64 + l31) * 33 + h;
    float a0 = 1.f, a1 = 2.f, b0 = 3.f, b1 = 4.f;
    if (V == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kp = 0; kp < 16; ++kp) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    } else if (V == 1 || V == 3) {
        constexpr int G = V == 1 ? 1 : 2;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kp = 0; kp < 16; kp += G) {
                float a[G][2], b[G][2];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    a[g][0] = ap[(kp + g) * 2];
                    a[g][1] = ap[(kp + g) * 2 + 32 * 33];
                    b[g][0] = bp[(kp + g) * 2];
                    b[g][1] = bp[(kp + g) * 2 + 32 * 33];
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][0], b[g][0], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][0], b[g][1], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][1], b[g][0], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][1], b[g][1], acc[1][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        constexpr int G = V == 2 ? 1 : 2;
        float a[2][G][2], b[2][G][2];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            a[0][g][0] = ap[g * 2]; a[0][g][1] = ap[g * 2 + 32 * 33];
            b[0][g][0] = bp[g * 2]; b[0][g][1] = bp[g * 2 + 32 * 33];
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kp = 0; kp < 16; kp += G) {
                const int cur = (kp / G) & 1, nxt = cur ^ 1;
                const int kn = (kp + G) & 15;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    a[nxt][g][0] = ap[(kn + g) * 2];
                    a[nxt][g][1] = ap[(kn + g) * 2 + 32 * 33];
                    b[nxt][g][0] = bp[(kn + g) * 2];
                    b[nxt][g][1] = bp[(kn + g) * 2 + 32 * 33];
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][g][0], b[cur][g][0], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][g][0], b[cur][g][1], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][g][1], b[cur][g][0], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][g][1], b[cur][g][1], acc[1][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4 * G, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * G, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V, int LDS_KB>
static void run(const char* name, float* d_out, int blocks_per_cu) {
    const int iters = 400, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<V, LDS_KB>), dim3(grid), dim3(256), 0, 0, d_out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<V, LDS_KB>), dim3(grid), dim3(256), 0, 0, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 /*waves*/ * iters * 64 /*mfma*/ * 4096.0;
    printf("%-44s WG/CU=%d  %8.1f us  %7.1f TFLOP/s  (%.1f%% of 157.3)\n", name, blocks_per_cu, ms * 1e3,
           flops / (ms * 1e-3) / 1e12, 100.0 * flops / (ms * 1e-3) / 157.3e12);
}

int main() {
    float* d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    // LDS_KB: 36 -> 4 WG/CU fit (144 KB), 72 -> 2 WG/CU, 144 -> 1 WG/CU
    run<0, 36>("V0 mfma only", d_out, 4);
    run<0, 72>("V0 mfma only", d_out, 2);
    run<0, 144>("V0 mfma only", d_out, 1);
    run<1, 36>("V1 4 reads + wait + 4 mfma", d_out, 4);
    run<1, 72>("V1 4 reads + wait + 4 mfma", d_out, 2);
    run<1, 144>("V1 4 reads + wait + 4 mfma", d_out, 1);
    run<2, 36>("V2 V1 + register double buffer", d_out, 4);
    run<2, 72>("V2 V1 + register double buffer", d_out, 2);
    run<2, 144>("V2 V1 + register double buffer", d_out, 1);
    run<3, 36>("V3 8 reads + wait + 8 mfma", d_out, 4);
    run<3, 144>("V3 8 reads + wait + 8 mfma", d_out, 1);
    run<4, 36>("V4 V3 + register double buffer", d_out, 4);
    run<4, 72>("V4 V3 + register double buffer", d_out, 2);
    run<4, 144>("V4 V3 + register double buffer", d_out, 1);
    hipFree(d_out);
    return 0;
}
