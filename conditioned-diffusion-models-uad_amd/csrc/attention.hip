// Mid-block self-attention core for gfx950: softmax(q k^T / sqrt(d)) v, d = 64, exact fp32 on
// v_mfma_f32_32x32x2_f32, flash-style (the N x N score matrix never leaves registers).
//
// Replaces QKVAttention.forward (src/models/modules/OpenAI_Unet.py:457-476): q,k,v = chunk(qkv, 3);
// heads are contiguous groups of 64 channels; w = softmax_fp32((q s)^T (k s)), s = 64^-1/4; a = w v^T.
// The two s factors are applied as one exact 2^-3 scale of q.
//
// Work split: workgroup = (sample, head, 128 queries); wave = 32 queries; key tiles of 64 through LDS.
//   S^T[key][query] = K . Q^T   : keys land in the 16 accumulator registers, the query on the lane, so the
//                                 softmax row reduction is 32 in-lane values + one exchange with lane^32.
//   O^T[c][query]  += V^T . P^T : the S^T accumulator IS the B operand (same lane = same query, register r of
//                                 lane-half h = key (r&3) + 8 (r>>2) + 4 h), no LDS round trip; the matching A
//                                 operand V[key(r,h)][c] is a conflict-free ds_read_b32 across 32 channels.
#include "kernels.h"

namespace cddpm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                           int N, int C) {
    __shared__ float4 ldsK[64 * 16];   // [key][slot ^ (key & 15)]
    __shared__ float ldsV[64 * 64];    // [key][channel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int heads = C >> 6;
    const int nqb = (N + 127) >> 7;
    int bid = blockIdx.x;
    const int qb = bid % nqb;
    bid /= nqb;
    const int hd = bid % heads;
    const int b = bid / heads;
    const int C3 = 3 * C;
    const float* base = qkv + (size_t)b * N * C3;

    // Q fragment of this lane's query: channels 8g + 4 lh + {0..3}, pre-scaled by 1/8
    const int query = qb * 128 + wave * 32 + li;
    const int qrow = min(query, N - 1);
    float4 qreg[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float4 v = *reinterpret_cast<const float4*>(base + (size_t)qrow * C3 + hd * 64 + 8 * g + 4 * lh);
        v.x *= 0.125f; v.y *= 0.125f; v.z *= 0.125f; v.w *= 0.125f;
        qreg[g] = v;
    }

    f32x16 O[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[ct][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = 0; k0 < N; k0 += 64) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int key = e >> 4, slot = e & 15;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (k0 + key < N) {
                const float* rowp = base + (size_t)(k0 + key) * C3 + hd * 64 + 4 * slot;
                kv = *reinterpret_cast<const float4*>(rowp + C);
                vv = *reinterpret_cast<const float4*>(rowp + 2 * C);
            }
            ldsK[key * 16 + (slot ^ (key & 15))] = kv;
            *reinterpret_cast<float4*>(&ldsV[key * 64 + 4 * slot]) = vv;
        }
        __syncthreads();

        // ---- S^T = K . Q^T for 2 sub-tiles of 32 keys
        f32x16 S[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) S[kt][r] = 0.f;
            const int row = 32 * kt + li;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 kf = ldsK[row * 16 + ((2 * g + lh) ^ (row & 15))];
                S[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qreg[g].x, S[kt], 0, 0, 0);
                S[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qreg[g].y, S[kt], 0, 0, 0);
                S[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qreg[g].z, S[kt], 0, 0, 0);
                S[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qreg[g].w, S[kt], 0, 0, 0);
            }
        }

        // ---- online softmax over keys (registers + lane^32)
        float tmax = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (key >= N) S[kt][r] = -INFINITY;
                tmax = fmaxf(tmax, S[kt][r]);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __expf(S[kt][r] - m_new);
                S[kt][r] = p;
                psum += p;
            }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[ct][r] *= alpha;

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float v0 = ldsV[key * 64 + li];
                const float v1 = ldsV[key * 64 + 32 + li];
                O[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, S[kt][r], O[0], 0, 0, 0);
                O[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, S[kt][r], O[1], 0, 0, 0);
            }
    }

    if (query < N) {
        const float inv = 1.0f / l_run;
        float* orow = out + ((size_t)b * N + query) * C + hd * 64;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                float4 v;
                v.x = O[ct][4 * rq + 0] * inv;
                v.y = O[ct][4 * rq + 1] * inv;
                v.z = O[ct][4 * rq + 2] * inv;
                v.w = O[ct][4 * rq + 3] * inv;
                *reinterpret_cast<float4*>(orow + 32 * ct + 8 * rq + 4 * lh) = v;
            }
    }
}

void launch_attention(const float* qkv, float* out, int B, int N, int C, hipStream_t stream) {
    const int heads = C / 64;
    const int nqb = (N + 127) / 128;
    hipLaunchKernelGGL(attention_kernel, dim3((unsigned)(B * heads * nqb)), dim3(256), 0, stream, qkv, out, N, C);
}

}  // namespace cddpm
