// Micro-benchmark: what fraction of the fp32 MFMA peak survives each ingredient of the conv main loop?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_ceiling mfma_ceiling.hip && ./mfma_ceiling
// Variants (all: 256-thread workgroups, 4 independent 32x32 accumulators per wave, random operands):
//   0 bare MFMA stream, operands in registers
//   1 + A fragments re-read from LDS (2 ds_read_b128 per 16 MFMAs, double-buffered)
//   2 + B fragments streamed from an L2-resident global buffer (2 global_load_dwordx4 per 16 MFMAs, 1 stage ahead)
//   3 = 2 + second accumulator set folded every 576 MFMAs
// each at 1 and 2 workgroups per CU (1 / 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ __launch_bounds__(256, 2) void k(const v4f* __restrict__ w, float* __restrict__ out, int iters, int wstride) {
    extern __shared__ v4f lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2048; i += 256) lds[i] = v4f{(float)(i % 7) * 0.01f, 0.02f, -0.01f, 0.005f * (lane & 3)};
    __syncthreads();
    f32x16 acc[4], tot[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; tot[i][r] = 0.f; }
    v4f a0 = lds[lane], a1 = lds[64 + lane], b[4][2];
    const v4f* wp = w + (size_t)(blockIdx.x % 64) * wstride + (wave >> 1) * 512 + lane;
    for (int g = 0; g < 4; ++g) { b[g][0] = wp[g * 64]; b[g][1] = wp[256 + g * 64]; }
    v4f n0 = a0, n1 = a1;
    for (int it = 0; it < iters; ++it) {
        const v4f* wn = wp + (size_t)((it + 1) & 63) * 1024;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (VAR >= 1) { n0 = lds[((it * 4 + g) * 64 + lane) & 2047]; n1 = lds[((it * 4 + g) * 64 + 512 + lane) & 2047]; }
            __builtin_amdgcn_sched_barrier(0);
            const float x0[4] = {a0.x, a0.y, a0.z, a0.w}, x1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float y0[4] = {b[g][0].x, b[g][0].y, b[g][0].z, b[g][0].w}, y1[4] = {b[g][1].x, b[g][1].y, b[g][1].z, b[g][1].w};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[m], y0[m], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[m], y1[m], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[m], y0[m], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[m], y1[m], acc[3], 0, 0, 0);
            }
            if (VAR >= 2) { b[g][0] = wn[g * 64]; b[g][1] = wn[256 + g * 64]; }
            __builtin_amdgcn_sched_barrier(0);
            if (VAR >= 1) { a0 = n0; a1 = n1; }
        }
        if (VAR >= 3 && (it % 9) == 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { tot[i] += acc[i]; for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r] + tot[i][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int VAR>
double run(const v4f* w, float* out, int blocks, int iters, size_t lds_bytes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(256), lds_bytes, 0, w, out, iters, 64 * 1024);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(256), lds_bytes, 0, w, out, iters, 64 * 1024);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 3.0 * blocks * 4.0 * iters * 64.0 * 4096.0;
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t nw = (size_t)64 * 64 * 1024;   // 64 "layers" x 64 stages x 16 KB = 64 MB
    v4f* w; float* out;
    hipMalloc((void**)&w, nw * sizeof(v4f)); hipMalloc((void**)&out, 4096 * 256 * sizeof(float));
    std::vector<float> h(nw * 4);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 4000;
    for (int occ = 1; occ <= 2; ++occ) {
        const int blocks = 256 * occ * 4;   // 4 rounds
        const size_t lds = occ == 1 ? 81920 : 57344;   // forces 1 or 2 workgroups per CU
        printf("workgroups/CU %d: bare %.1f  +ldsA %.1f  +globalB %.1f  +fold %.1f TFLOP/s\n", occ,
               run<0>(w, out, blocks, iters, lds), run<1>(w, out, blocks, iters, lds), run<2>(w, out, blocks, iters, lds),
               run<3>(w, out, blocks, iters, lds));
    }
    return 0;
}
