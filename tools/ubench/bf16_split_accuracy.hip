// Micro-benchmark (numerics): can fp32 GEMM-shaped work run on the bf16 matrix pipe without losing fp32 accuracy?
//   hipcc --offload-arch=gfx950 -O3 -o bf16_split_accuracy bf16_split_accuracy.hip && ./bf16_split_accuracy
// Every fp32 operand is split exactly into three bf16 terms (x = hi + mid + lo, round-to-nearest at each step); a
// product a*b is then the sum of up to nine bf16*bf16 products, each exact in fp32, accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16. Compared against an fp64 host result, next to the fp32 MFMA (32x32x2) chain the
// convolution kernel uses today:
//   f32      fp32 MFMA, one accumulation chain            f32x2lvl  fp32 MFMA, folded every 288 (one 32-channel chunk)
//   x3       hi*hi + hi*mid + mid*hi                       x6        + mid*mid + hi*lo + lo*hi
//   x8       + mid*lo + lo*mid                              x9        + lo*lo        (x6/x9 also folded every 288)
//   h3       fp16 instead of bf16: x = hi + mid with 11-bit terms (|x - hi - mid| <= 2^-24 |x| while mid stays a normal
//            fp16, i.e. |x| >= 0.25 -- smaller values keep an ABSOLUTE error <= 2^-25; B is pre-scaled by 2^12 and the
//            result unscaled, A is not), three products hi*hi + hi*mid + mid*hi on v_mfma_f32_32x32x16_f16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// fp16 two-term split, three products; B scaled by 2^12 (exact), result unscaled at the end
__global__ void gemm_h3(const float* A, const float* B, float* C, int K, int fold) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc, tot;
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; tot[i] = 0.f; }
    for (int k = 0; k < K; k += 16) {
        f16x8 ah, am, bh, bm;
        for (int j = 0; j < 8; ++j) {
            const float x = A[r * K + k + 8 * h + j];
            ah[j] = (_Float16)x; am[j] = (_Float16)(x - (float)ah[j]);
            const float y = B[(k + 8 * h + j) * 32 + r] * 4096.0f;
            bh[j] = (_Float16)y; bm[j] = (_Float16)(y - (float)bh[j]);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        if (fold && ((k + 16) % fold) == 0) { tot += acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f; }
    }
    tot += acc;
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = tot[i] * (1.0f / 4096.0f);
}

// A: [32][K] row-major, B: [K][32] row-major, C: [32][32]
template <int MODE>
__global__ void gemm(const float* A, const float* B, float* C, int K, int fold) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc, tot;
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; tot[i] = 0.f; }
    if (MODE == 0) {
        for (int k = 0; k < K; k += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], acc, 0, 0, 0);
            if (fold && ((k + 2) % fold) == 0) { tot += acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f; }
        }
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
            for (int j = 0; j < 8; ++j) {
                __bf16 x, y, z;
                split3(A[r * K + k + 8 * h + j], x, y, z); ah[j] = x; am[j] = y; al[j] = z;
                split3(B[(k + 8 * h + j) * 32 + r], x, y, z); bh[j] = x; bm[j] = y; bl[j] = z;
            }
            // smallest terms first
            if (MODE >= 9) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, acc, 0, 0, 0);
            if (MODE >= 8) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bl, acc, 0, 0, 0);
                             acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bm, acc, 0, 0, 0); }
            if (MODE >= 6) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                             acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                             acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            if (fold && ((k + 16) % fold) == 0) { tot += acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f; }
        }
    }
    tot += acc;
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = tot[i];
}

int main() {
    const int Ks[3] = {1152, 2304, 4608};
    std::mt19937_64 rng(1234);
    std::normal_distribution<double> nd(0.0, 1.0);
    for (int K : Ks) {
        std::vector<float> A(32 * K), B(K * 32);
        for (auto& v : A) { const double x = nd(rng); v = (float)(x / (1.0 + std::exp(-x))); }     // SiLU(N(0,1))
        for (auto& v : B) v = (float)(nd(rng) / std::sqrt((double)K));
        std::vector<double> ref(1024, 0.0);
        std::vector<float> cpu32(1024, 0.f);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double s = 0.0; float f = 0.f;
                for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * (double)B[k * 32 + j]; f = std::fma(A[i * K + k], B[k * 32 + j], f); }
                ref[i * 32 + j] = s; cpu32[i * 32 + j] = f;
            }
        double rms = 0; for (double v : ref) rms += v * v; rms = std::sqrt(rms / 1024);
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> C(1024);
        auto report = [&](const char* name, const float* c) {
            double e2 = 0, emax = 0;
            for (int i = 0; i < 1024; ++i) { const double e = (double)c[i] - ref[i]; e2 += e * e; emax = std::fmax(emax, std::fabs(e)); }
            printf("K=%5d %-10s rms err / rms(C) = %.3e   max err / rms(C) = %.3e\n", K, name, std::sqrt(e2 / 1024) / rms, emax / rms);
        };
        report("cpu fma32", cpu32.data());
#define RUN(MODE, FOLD, NAME) hipLaunchKernelGGL(gemm<MODE>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, FOLD); \
        hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); report(NAME, C.data());
        RUN(0, 0, "f32")
        RUN(0, 288, "f32x2lvl")
        RUN(3, 0, "x3")
        RUN(6, 0, "x6")
        RUN(6, 288, "x6x2lvl")
        RUN(8, 0, "x8")
        RUN(9, 0, "x9")
        RUN(9, 288, "x9x2lvl")
        RUN(6, 96, "x6 fold96")
#define RUNH(FOLD, NAME) hipLaunchKernelGGL(gemm_h3, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, FOLD); \
        hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); report(NAME, C.data());
        RUNH(0, "h3")
        RUNH(288, "h3x2lvl")
        RUNH(96, "h3 fold96")
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
