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

// ------------------------------------------------------------------------------------------------------------------
// Backward of the attention core, flash-style like the forward: the N x N matrices P and dS never leave registers.
// With S = (q / 8) . k, P = softmax_rows(S), a = P v and an upstream gradient dA:
//   D_i = sum_c dA_ic a_ic,   dP_ij = dA_i . v_j,   dS_ij = P_ij (dP_ij - D_i),
//   dq_i = (1/8) sum_j dS_ij k_j,   dk_j = sum_i dS_ij (q_i / 8),   dv_j = sum_i P_ij dA_i.
// Two kernels, both in the forward's layout trick (the accumulator of the first product IS the B operand of the second):
//   attention_bwd_q_kernel  : lane = query (as the forward). Sweep 1 over the key tiles is the forward itself (m_i, l_i, a_i -> D_i);
//                             sweep 2 forms S^T and dP^T = V . dA^T by MFMA, dS^T in registers, dQ^T += K^T . dS^T. Also writes
//                             (m_i + log l_i, D_i) per query for the second kernel.
//   attention_bwd_kv_kernel : lane = key. Per query tile: S = Q . K^T and dP = dA . V^T by MFMA (queries in the accumulator registers),
//                             P and dS in registers, dV^T += dA^T . P, dK^T += (Q/8)^T . dS.
// Exact fp32 products (v_mfma_f32_32x32x2_f32) as the forward. 7 N x N x 64 products + the forward's 2 instead of the 5 GEMM launches
// + 2 softmax passes over materialised [B heads][N][N] matrices of the first version (3.4 ms of a 16 x 128 x 128 training step).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attention_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ da,
                                                                 float* __restrict__ dqkv, float* __restrict__ stats /*[B][heads][N][2]*/,
                                                                 int N, int C) {
    __shared__ float4 ldsK[64 * 16];   // K tile, [key][slot ^ (key & 15)]
    __shared__ float4 ldsV4[64 * 16];  // V tile, same layout (A operand of dP^T = V . dA^T)
    __shared__ float ldsP[64 * 64];    // plain [key][channel]: V in sweep 1 (O^T += V^T P^T), K in sweep 2 (dQ^T += K^T dS^T)
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
    const int query = qb * 128 + wave * 32 + li;
    const int qrow = min(query, N - 1);
    float4 qreg[8], dareg[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float4 v = *reinterpret_cast<const float4*>(base + (size_t)qrow * C3 + hd * 64 + 8 * g + 4 * lh);
        v.x *= 0.125f; v.y *= 0.125f; v.z *= 0.125f; v.w *= 0.125f;
        qreg[g] = v;
        dareg[g] = *reinterpret_cast<const float4*>(da + ((size_t)b * N + qrow) * C + hd * 64 + 8 * g + 4 * lh);
    }
    float m_run = -INFINITY, l_run = 0.f, Dq = 0.f;
    // ---------------- sweep 1: the forward (running max / sum, O^T) -> D
    {
        f32x16 O[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[ct][r] = 0.f;
        for (int k0 = 0; k0 < N; k0 += 64) {
            __syncthreads();
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
                *reinterpret_cast<float4*>(&ldsP[key * 64 + 4 * slot]) = vv;
            }
            __syncthreads();
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
                    const float pv = __expf(S[kt][r] - m_new);
                    S[kt][r] = pv;
                    psum += pv;
                }
            psum += __shfl_xor(psum, 32, 64);
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[ct][r] *= alpha;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float v0 = ldsP[key * 64 + li];
                    const float v1 = ldsP[key * 64 + 32 + li];
                    O[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, S[kt][r], O[0], 0, 0, 0);
                    O[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, S[kt][r], O[1], 0, 0, 0);
                }
        }
        // D = sum_c dA_c a_c, a = O / l. O^T[c][query]: register 4 rq + x of tile ct = channel 32 ct + 8 rq + 4 lh + x, i.e. dareg[4 ct + rq]
        const float inv = 1.0f / l_run;
        float dsum = 0.f;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const float4 d4 = dareg[4 * ct + rq];
                dsum += d4.x * O[ct][4 * rq + 0] + d4.y * O[ct][4 * rq + 1] + d4.z * O[ct][4 * rq + 2] + d4.w * O[ct][4 * rq + 3];
            }
        dsum += __shfl_xor(dsum, 32, 64);
        Dq = dsum * inv;
    }
    const float lse = m_run + __logf(l_run);
    if (query < N && lh == 0) {
        float* st = stats + (((size_t)b * heads + hd) * N + query) * 2;
        st[0] = lse; st[1] = Dq;
    }
    // ---------------- sweep 2: dQ^T += K^T . dS^T
    f32x16 dQ[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) dQ[ct][r] = 0.f;
    for (int k0 = 0; k0 < N; k0 += 64) {
        __syncthreads();
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
            ldsV4[key * 16 + (slot ^ (key & 15))] = vv;
            *reinterpret_cast<float4*>(&ldsP[key * 64 + 4 * slot]) = kv;
        }
        __syncthreads();
        f32x16 S[2], dP[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { S[kt][r] = 0.f; dP[kt][r] = 0.f; }
            const int row = 32 * kt + li;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 kf = ldsK[row * 16 + ((2 * g + lh) ^ (row & 15))];
                const float4 vf = ldsV4[row * 16 + ((2 * g + lh) ^ (row & 15))];
                S[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qreg[g].x, S[kt], 0, 0, 0);
                S[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qreg[g].y, S[kt], 0, 0, 0);
                S[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qreg[g].z, S[kt], 0, 0, 0);
                S[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qreg[g].w, S[kt], 0, 0, 0);
                dP[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, dareg[g].x, dP[kt], 0, 0, 0);
                dP[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, dareg[g].y, dP[kt], 0, 0, 0);
                dP[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, dareg[g].z, dP[kt], 0, 0, 0);
                dP[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, dareg[g].w, dP[kt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float pv = (key < N) ? __expf(S[kt][r] - lse) : 0.f;
                S[kt][r] = pv * (dP[kt][r] - Dq);            // dS^T
            }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float k0v = ldsP[key * 64 + li];
                const float k1v = ldsP[key * 64 + 32 + li];
                dQ[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(k0v, S[kt][r], dQ[0], 0, 0, 0);
                dQ[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(k1v, S[kt][r], dQ[1], 0, 0, 0);
            }
    }
    if (query < N) {
        float* orow = dqkv + ((size_t)b * N + query) * C3 + hd * 64;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                float4 v;
                v.x = dQ[ct][4 * rq + 0] * 0.125f; v.y = dQ[ct][4 * rq + 1] * 0.125f;
                v.z = dQ[ct][4 * rq + 2] * 0.125f; v.w = dQ[ct][4 * rq + 3] * 0.125f;
                *reinterpret_cast<float4*>(orow + 32 * ct + 8 * rq + 4 * lh) = v;
            }
    }
}

__global__ __launch_bounds__(256, 2) void attention_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ da,
                                                                  float* __restrict__ dqkv, const float* __restrict__ stats, int N, int C) {
    __shared__ float4 ldsQ4[64 * 16];   // (Q / 8) tile, [query][slot ^ (query & 15)]: A operand of S = Q . K^T
    __shared__ float4 ldsA4[64 * 16];   // dA tile, same layout: A operand of dP = dA . V^T
    __shared__ float ldsQp[64 * 64];    // plain [query][channel] copies: A operands of dK^T += Q^T dS and dV^T += dA^T P
    __shared__ float ldsAp[64 * 64];
    __shared__ float ldsL[64], ldsD[64];   // per query: m + log l, D
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int heads = C >> 6;
    const int nkb = (N + 127) >> 7;
    int bid = blockIdx.x;
    const int kb = bid % nkb;
    bid /= nkb;
    const int hd = bid % heads;
    const int b = bid / heads;
    const int C3 = 3 * C;
    const float* base = qkv + (size_t)b * N * C3;
    const int keyi = kb * 128 + wave * 32 + li;
    const int krow = min(keyi, N - 1);
    float4 kreg[8], vreg[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        kreg[g] = *reinterpret_cast<const float4*>(base + (size_t)krow * C3 + C + hd * 64 + 8 * g + 4 * lh);
        vreg[g] = *reinterpret_cast<const float4*>(base + (size_t)krow * C3 + 2 * C + hd * 64 + 8 * g + 4 * lh);
    }
    f32x16 dK[2], dV[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dK[ct][r] = 0.f; dV[ct][r] = 0.f; }
    const float* st = stats + ((size_t)b * heads + hd) * N * 2;
    for (int q0 = 0; q0 < N; q0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int qq = e >> 4, slot = e & 15;
            float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), av = qv;
            if (q0 + qq < N) {
                qv = *reinterpret_cast<const float4*>(base + (size_t)(q0 + qq) * C3 + hd * 64 + 4 * slot);
                qv.x *= 0.125f; qv.y *= 0.125f; qv.z *= 0.125f; qv.w *= 0.125f;
                av = *reinterpret_cast<const float4*>(da + ((size_t)b * N + q0 + qq) * C + hd * 64 + 4 * slot);
            }
            ldsQ4[qq * 16 + (slot ^ (qq & 15))] = qv;
            ldsA4[qq * 16 + (slot ^ (qq & 15))] = av;
            *reinterpret_cast<float4*>(&ldsQp[qq * 64 + 4 * slot]) = qv;
            *reinterpret_cast<float4*>(&ldsAp[qq * 64 + 4 * slot]) = av;
        }
        if (tid < 64) {
            const bool ok = q0 + tid < N;
            ldsL[tid] = ok ? st[(size_t)(q0 + tid) * 2] : INFINITY;      // exp(S - inf) = 0: queries past the end contribute nothing
            ldsD[tid] = ok ? st[(size_t)(q0 + tid) * 2 + 1] : 0.f;
        }
        __syncthreads();
        // S[query][key] = (Q/8) . K^T and dP[query][key] = dA . V^T: queries in the accumulator registers, the key on the lane;
        // one 32-query sub-tile at a time (32 instead of 64 live accumulator registers for S and dP: no spills)
#pragma unroll 1
        for (int qt = 0; qt < 2; ++qt) {
            f32x16 S, dP;
#pragma unroll
            for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
            const int row = 32 * qt + li;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 qf = ldsQ4[row * 16 + ((2 * g + lh) ^ (row & 15))];
                const float4 af = ldsA4[row * 16 + ((2 * g + lh) ^ (row & 15))];
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.x, kreg[g].x, S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.y, kreg[g].y, S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.z, kreg[g].z, S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.w, kreg[g].w, S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, vreg[g].x, dP, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, vreg[g].y, dP, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, vreg[g].z, dP, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, vreg[g].w, dP, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = 32 * qt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float pv = __expf(S[r] - ldsL[qq]);
                const float ds = pv * (dP[r] - ldsD[qq]);
                const float a0 = ldsAp[qq * 64 + li], a1 = ldsAp[qq * 64 + 32 + li];
                const float x0 = ldsQp[qq * 64 + li], x1 = ldsQp[qq * 64 + 32 + li];
                dV[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, pv, dV[0], 0, 0, 0);
                dV[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, pv, dV[1], 0, 0, 0);
                dK[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, ds, dK[0], 0, 0, 0);
                dK[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, ds, dK[1], 0, 0, 0);
            }
        }
    }
    if (keyi < N) {
        float* krow_o = dqkv + ((size_t)b * N + keyi) * C3 + C + hd * 64;
        float* vrow_o = krow_o + C;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                *reinterpret_cast<float4*>(krow_o + 32 * ct + 8 * rq + 4 * lh) =
                    make_float4(dK[ct][4 * rq + 0], dK[ct][4 * rq + 1], dK[ct][4 * rq + 2], dK[ct][4 * rq + 3]);
                *reinterpret_cast<float4*>(vrow_o + 32 * ct + 8 * rq + 4 * lh) =
                    make_float4(dV[ct][4 * rq + 0], dV[ct][4 * rq + 1], dV[ct][4 * rq + 2], dV[ct][4 * rq + 3]);
            }
    }
}

// stats: [B][heads][N][2] floats of scratch
void launch_attention_backward_flash(const float* qkv, const float* da, float* dqkv, float* stats, int B, int N, int C, hipStream_t stream) {
    const int heads = C / 64;
    const int nb = (N + 127) / 128;
    hipLaunchKernelGGL(attention_bwd_q_kernel, dim3((unsigned)(B * heads * nb)), dim3(256), 0, stream, qkv, da, dqkv, stats, N, C);
    hipLaunchKernelGGL(attention_bwd_kv_kernel, dim3((unsigned)(B * heads * nb)), dim3(256), 0, stream, qkv, da, dqkv, stats, N, C);
}

}  // namespace cddpm
