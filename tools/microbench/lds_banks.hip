// LDS bank-conflict probe for gfx950: cycles per ds_read_b128 / ds_write_b128 as a function of the row stride of
// the MFMA-operand access pattern (lane -> row = lane % 16, 16-byte piece = lane / 16).  One wave per workgroup.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_banks.hip -o lds_banks && ./lds_banks
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(int stride, int piece_stride, int mode, long long *out, float *sink) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x;
    for (int i = lane * 16; i < 65536; i += 64 * 16) *reinterpret_cast<f32x4 *>(lds + i) = (f32x4){1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    int off;
    if (mode == 0 || mode == 1) off = (lane % 16) * stride + (lane / 16) * piece_stride;      // MFMA operand pattern
    else off = (lane / 4) * stride + (lane % 4) * piece_stride;                                   // staging pattern: 4 pieces of a row per 4 lanes
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    for (int it = 0; it < 1024; ++it) {
        if (mode == 0 || mode == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += *reinterpret_cast<volatile f32x4 *>(lds + ((off + u * 4096) & 65535 & ~15));
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) *reinterpret_cast<volatile f32x4 *>(lds + ((off + u * 4096) & 65535 & ~15)) = acc;
        }
    }
    const long long t1 = clock64();
    if (lane == 0) out[0] = t1 - t0;
    sink[lane] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main() {
    long long *out; float *sink;
    hipMalloc(&out, 8); hipMalloc(&sink, 256);
    const char *names[4] = {"read  operand pattern (row = lane%16)", "write operand pattern", "read  staging pattern (row = lane/4)", "write staging pattern"};
    for (int mode = 0; mode < 4; ++mode) {
        printf("%s, 16-byte pieces adjacent:\n", names[mode]);
        for (int stride : {64, 80, 96, 112, 128, 144, 160, 176, 192, 208, 272, 288}) {
            probe<<<1, 64>>>(stride, 16, mode, out, sink);
            long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
            printf("  stride %3d B: %.1f cycles per instruction\n", stride, (double)h / (1024 * 8));
        }
    }
    return 0;
}
