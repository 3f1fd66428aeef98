// Context encoder of the cDDPM: timm `resnet50(in_chans=1, num_classes=cond_dim)` in
// inference mode, as built by the reference's get_encoder (src/models/modules/DDPM_encoder.py:6-29; the SparK variant
// src/models/modules/spark/models.py:89-109 creates the same network). One call per slice BATCH, 1.3 GMAC per 128x128
// slice against 132 790 GMAC for its 1000 reverse steps: performance-irrelevant, so these kernels are plain fp32 FMA
// tiles, written for clarity and exact fp32 arithmetic, not for the matrix pipe.
//
// PARITY UNPINNED: timm 0.6.7 is not installed in the build image, so the architecture below follows timm's published
// ResNet-50 ("v1.5": stride on the 3x3 conv of a bottleneck, 7x7/2 stem, 3x3/2 max-pool, avg-pool downsample NOT used,
// BatchNorm eps 1e-5, global average pool, fc) and its state_dict names; it is checked against a torch restatement of
// that description (oracle/encoder_oracle.py), not against timm itself.
//
// Layout: NHWC fp32. BatchNorm (eval) is folded into a per-channel (scale, shift) at load time, in double.
#include "kernels.h"
#include "../../include/cddpm.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace cddpm {
namespace {

// ---- stem: conv 7x7 stride 2 pad 3, 1 -> 64 channels, + BN + ReLU.  x [B,H,W] -> y [B,H/2,W/2,64]
__global__ __launch_bounds__(256) void enc_stem_kernel(const float* __restrict__ x, const float* __restrict__ w /*[49][64]*/,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       float* __restrict__ y, int B, int H, int W, int Ho, int Wo) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;       // one thread per (pixel, 4 channels)
    if (e >= (long long)B * Ho * Wo * 16) return;
    const int c4 = (int)(e & 15);
    long long p = e >> 4;
    const int xo = (int)(p % Wo); p /= Wo;
    const int yo = (int)(p % Ho);
    const int b = (int)(p / Ho);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < 7; ++ky) {
        const int yi = 2 * yo + ky - 3;
        if (yi < 0 || yi >= H) continue;
        for (int kx = 0; kx < 7; ++kx) {
            const int xi = 2 * xo + kx - 3;
            if (xi < 0 || xi >= W) continue;
            const float v = x[((size_t)b * H + yi) * W + xi];
            const float4 wv = *reinterpret_cast<const float4*>(w + (ky * 7 + kx) * 64 + 4 * c4);
            acc[0] = fmaf(v, wv.x, acc[0]); acc[1] = fmaf(v, wv.y, acc[1]);
            acc[2] = fmaf(v, wv.z, acc[2]); acc[3] = fmaf(v, wv.w, acc[3]);
        }
    }
    const float4 sc = *reinterpret_cast<const float4*>(scale + 4 * c4), sh = *reinterpret_cast<const float4*>(shift + 4 * c4);
    float4 o;
    o.x = fmaxf(fmaf(acc[0], sc.x, sh.x), 0.f); o.y = fmaxf(fmaf(acc[1], sc.y, sh.y), 0.f);
    o.z = fmaxf(fmaf(acc[2], sc.z, sh.z), 0.f); o.w = fmaxf(fmaf(acc[3], sc.w, sh.w), 0.f);
    *reinterpret_cast<float4*>(y + (((size_t)b * Ho + yo) * Wo + xo) * 64 + 4 * c4) = o;
}

// ---- max-pool 3x3 stride 2 pad 1 on NHWC
__global__ __launch_bounds__(256) void enc_maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W,
                                                          int C, int Ho, int Wo) {
    const int nq = C >> 2;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)B * Ho * Wo * nq) return;
    const int c4 = (int)(e % nq);
    long long p = e / nq;
    const int xo = (int)(p % Wo); p /= Wo;
    const int yo = (int)(p % Ho);
    const int b = (int)(p / Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int ky = 0; ky < 3; ++ky) {
        const int yi = 2 * yo + ky - 1;
        if (yi < 0 || yi >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int xi = 2 * xo + kx - 1;
            if (xi < 0 || xi >= W) continue;
            const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)b * H + yi) * W + xi) * C + 4 * c4);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    *reinterpret_cast<float4*>(y + (((size_t)b * Ho + yo) * Wo + xo) * C + 4 * c4) = m;
}

// ---- generic convolution (K = 1 or 3, stride 1 or 2, pad K/2) + folded BN + optional residual + optional ReLU.
//      Tile: 64 output pixels x 64 output channels per 256-thread workgroup, 16 input channels x 1 tap per LDS step,
//      4 x 4 outputs per thread. w packed [taps][Cin][Cout]. Cin % 16 == 0, Cout % 64 == 0.
struct EncConv {
    const float* x; const float* w; const float* scale; const float* shift; const float* res; float* y;
    int B, H, W, Cin, Ho, Wo, Cout, K, stride, relu;
};
__global__ __launch_bounds__(256) void enc_conv_kernel(const EncConv a) {
    __shared__ float As[16][64 + 4];
    __shared__ float Bs[16][64 + 4];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;                 // 4 couts / 4 pixels per thread
    const long long M = (long long)a.B * a.Ho * a.Wo;
    const long long p0 = (long long)blockIdx.x * 64;
    const int co0 = blockIdx.y * 64;
    // loader roles
    const int lp = tid >> 2, lc4 = tid & 3;                 // A: pixel lp of the tile, 4 channels
    const long long pl = p0 + lp;
    int lb = 0, lyo = 0, lxo = 0;
    const bool lvalid = pl < M;
    if (lvalid) { long long t = pl; lxo = (int)(t % a.Wo); t /= a.Wo; lyo = (int)(t % a.Ho); lb = (int)(t / a.Ho); }
    const int bk = tid >> 4, bc4 = tid & 15;                // B: input channel bk of the step, 4 couts
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int pad = a.K / 2, taps = a.K * a.K;
    for (int t = 0; t < taps; ++t) {
        const int ky = t / a.K, kx = t - ky * a.K;
        const int yi = lyo * a.stride + ky - pad, xi = lxo * a.stride + kx - pad;
        const bool inb = lvalid && yi >= 0 && yi < a.H && xi >= 0 && xi < a.W;
        const float* xp = a.x + (((size_t)lb * a.H + (inb ? yi : 0)) * a.W + (inb ? xi : 0)) * a.Cin;
        const float* wp = a.w + (size_t)t * a.Cin * a.Cout;
        for (int c0 = 0; c0 < a.Cin; c0 += 16) {
            float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
            if (inb) av = *reinterpret_cast<const float4*>(xp + c0 + 4 * lc4);
            const float4 bv = *reinterpret_cast<const float4*>(wp + (size_t)(c0 + bk) * a.Cout + co0 + 4 * bc4);
            __syncthreads();
            As[4 * lc4 + 0][lp] = av.x; As[4 * lc4 + 1][lp] = av.y; As[4 * lc4 + 2][lp] = av.z; As[4 * lc4 + 3][lp] = av.w;
            *reinterpret_cast<float4*>(&Bs[bk][4 * bc4]) = bv;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float4 a4 = *reinterpret_cast<const float4*>(&As[k][4 * ty]);
                const float4 b4 = *reinterpret_cast<const float4*>(&Bs[k][4 * tx]);
                const float aa[4] = {a4.x, a4.y, a4.z, a4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(aa[i], bb[j], acc[i][j]);
            }
        }
    }
    const int co = co0 + 4 * tx;
    const float4 sc = *reinterpret_cast<const float4*>(a.scale + co), sh = *reinterpret_cast<const float4*>(a.shift + co);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long p = p0 + 4 * ty + i;
        if (p >= M) continue;
        float4 o;
        o.x = fmaf(acc[i][0], sc.x, sh.x); o.y = fmaf(acc[i][1], sc.y, sh.y);
        o.z = fmaf(acc[i][2], sc.z, sh.z); o.w = fmaf(acc[i][3], sc.w, sh.w);
        if (a.res) {
            const float4 r = *reinterpret_cast<const float4*>(a.res + (size_t)p * a.Cout + co);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (a.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<float4*>(a.y + (size_t)p * a.Cout + co) = o;
    }
}

// ---- global average pool: x [B,HW,C] -> g [B,C]
__global__ __launch_bounds__(256) void enc_avgpool_kernel(const float* __restrict__ x, float* __restrict__ g, int B, int HW, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += x[((size_t)b * HW + p) * C + c];
    g[i] = s / (float)HW;
}
// ---- fc: y[b][n] = sum_k g[b][k] W[n][k] + bias[n]   (one wave per output)
__global__ __launch_bounds__(64) void enc_fc_kernel(const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ bias,
                                                    float* __restrict__ y, int B, int K, int N) {
    const int o = blockIdx.x;
    if (o >= B * N) return;
    const int b = o / N, n = o - b * N;
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 64) s = fmaf(g[(size_t)b * K + k], w[(size_t)n * K + k], s);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (threadIdx.x == 0) y[o] = s + bias[n];
}

struct ConvBN { float* w = nullptr; float* scale = nullptr; float* shift = nullptr; int Cin = 0, Cout = 0, K = 0, stride = 1; };
struct Bottleneck { ConvBN c1, c2, c3, down; bool has_down = false; };

}  // namespace
}  // namespace cddpm

using namespace cddpm;

struct cddpm_encoder_ctx {
    int device = 0, num_classes = 0, max_batch = 0, max_h = 0, max_w = 0;
    bool loaded = false;
    std::string err;
    std::vector<void*> allocs;
    ConvBN stem;                          // w [49][64]
    std::vector<Bottleneck> blocks;       // 3 + 4 + 6 + 3
    float* fc_w = nullptr; float* fc_b = nullptr;
    float *buf0 = nullptr, *buf1 = nullptr, *buf2 = nullptr, *buf3 = nullptr, *gap = nullptr;
    size_t buf_elems = 0;
};

static thread_local std::string g_enc_err;
static int efail(cddpm_encoder_ctx* h, const char* fmt, ...) {
    char b[512];
    va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap);
    if (h) h->err = b; else g_enc_err = b;
    return -1;
}
#define ECHECK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return efail(h, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

static const int kStageBlocks[4] = {3, 4, 6, 3};
static const int kStagePlanes[4] = {64, 128, 256, 512};

extern "C" {

const char* cddpm_encoder_last_error(cddpm_encoder_handle h) { return h ? h->err.c_str() : g_enc_err.c_str(); }

int cddpm_encoder_create(cddpm_encoder_handle* out, int num_classes, int max_batch, int max_h, int max_w, int device) {
    if (!out) return efail(nullptr, "out handle is NULL");
    *out = nullptr;
    if (num_classes < 1 || max_batch < 1 || max_h < 32 || max_w < 32)
        return efail(nullptr, "cddpm_encoder_create: need num_classes >= 1, max_batch >= 1, max_h, max_w >= 32 (got %d, %d, %dx%d)",
                     num_classes, max_batch, max_h, max_w);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return efail(nullptr, "no HIP device (this library needs an MI355X / gfx950)");
    if (device < 0 || device >= ndev) return efail(nullptr, "device %d out of range (%d devices)", device, ndev);
    cddpm_encoder_ctx* h = new cddpm_encoder_ctx;
    h->device = device; h->num_classes = num_classes; h->max_batch = max_batch; h->max_h = max_h; h->max_w = max_w;
    // largest activation: stem output [B, H/2, W/2, 64] = layer1 output [B, H/4, W/4, 256]
    h->buf_elems = (size_t)max_batch * ((max_h + 1) / 2) * ((max_w + 1) / 2) * 64;
    *out = h;
    return 0;
}

void cddpm_encoder_destroy(cddpm_encoder_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
}

int cddpm_encoder_num_weights(void) { return 1 + 5 + 16 * 15 + 4 * 5 + 2; }   // informational: tensors read by load_weights

/* names: timm ResNet state_dict keys. */
int cddpm_encoder_load_weights(cddpm_encoder_handle h, const char* const* names, const float* const* host_ptrs,
                               const int64_t* numels, int n) {
    if (!h) return -1;
    ECHECK(h, hipSetDevice(h->device));
    std::map<std::string, std::pair<const float*, int64_t>> m;
    for (int i = 0; i < n; ++i) m[names[i]] = {host_ptrs[i], numels[i]};
    auto get = [&](const std::string& k, int64_t want) -> const float* {
        auto it = m.find(k);
        if (it == m.end()) { efail(h, "missing weight '%s'", k.c_str()); return nullptr; }
        if (it->second.second != want) { efail(h, "weight '%s' has %lld elements, expected %lld", k.c_str(), (long long)it->second.second, (long long)want); return nullptr; }
        return it->second.first;
    };
    auto upload = [&](float** dst, const std::vector<float>& v) -> int {
        void* p = nullptr;
        if (hipMalloc(&p, v.size() * sizeof(float)) != hipSuccess) return efail(h, "hipMalloc failed");
        h->allocs.push_back(p);
        if (hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return efail(h, "hipMemcpy failed");
        *dst = static_cast<float*>(p);
        return 0;
    };
    // conv weight [Cout][Cin][K][K] -> [taps][Cin][Cout]; BN (eval) folded: scale = g / sqrt(var + eps), shift = b - mean * scale
    auto load_convbn = [&](const std::string& conv, const std::string& bn, int Cin, int Cout, int K, int stride, ConvBN* c) -> int {
        const float* w = get(conv + ".weight", (int64_t)Cout * Cin * K * K);
        const float* g = get(bn + ".weight", Cout); const float* b = get(bn + ".bias", Cout);
        const float* mu = get(bn + ".running_mean", Cout); const float* var = get(bn + ".running_var", Cout);
        if (!w || !g || !b || !mu || !var) return -1;
        std::vector<float> pk((size_t)K * K * Cin * Cout), sc(Cout), sh(Cout);
        for (int co = 0; co < Cout; ++co)
            for (int ci = 0; ci < Cin; ++ci)
                for (int t = 0; t < K * K; ++t) pk[((size_t)t * Cin + ci) * Cout + co] = w[((size_t)co * Cin + ci) * K * K + t];
        for (int co = 0; co < Cout; ++co) {
            const double s = (double)g[co] / std::sqrt((double)var[co] + 1e-5);
            sc[co] = (float)s; sh[co] = (float)((double)b[co] - (double)mu[co] * s);
        }
        c->Cin = Cin; c->Cout = Cout; c->K = K; c->stride = stride;
        if (upload(&c->w, pk) || upload(&c->scale, sc) || upload(&c->shift, sh)) return -1;
        return 0;
    };
    if (load_convbn("conv1", "bn1", 1, 64, 7, 2, &h->stem)) return -1;
    h->blocks.clear();
    int inplanes = 64;
    for (int s = 0; s < 4; ++s)
        for (int i = 0; i < kStageBlocks[s]; ++i) {
            Bottleneck bk;
            const int planes = kStagePlanes[s], stride = (i == 0 && s > 0) ? 2 : 1;
            const std::string p = "layer" + std::to_string(s + 1) + "." + std::to_string(i);
            if (load_convbn(p + ".conv1", p + ".bn1", inplanes, planes, 1, 1, &bk.c1)) return -1;
            if (load_convbn(p + ".conv2", p + ".bn2", planes, planes, 3, stride, &bk.c2)) return -1;
            if (load_convbn(p + ".conv3", p + ".bn3", planes, planes * 4, 1, 1, &bk.c3)) return -1;
            bk.has_down = (i == 0);
            if (bk.has_down && load_convbn(p + ".downsample.0", p + ".downsample.1", inplanes, planes * 4, 1, stride, &bk.down)) return -1;
            inplanes = planes * 4;
            h->blocks.push_back(bk);
        }
    const float* fw = get("fc.weight", (int64_t)h->num_classes * 2048);
    const float* fb = get("fc.bias", h->num_classes);
    if (!fw || !fb) return -1;
    if (upload(&h->fc_w, std::vector<float>(fw, fw + (size_t)h->num_classes * 2048)) || upload(&h->fc_b, std::vector<float>(fb, fb + h->num_classes))) return -1;
    for (float** b : {&h->buf0, &h->buf1, &h->buf2, &h->buf3}) {
        void* p = nullptr;
        if (hipMalloc(&p, h->buf_elems * sizeof(float)) != hipSuccess) return efail(h, "hipMalloc(workspace) failed");
        h->allocs.push_back(p);
        *b = static_cast<float*>(p);
    }
    {
        void* p = nullptr;
        if (hipMalloc(&p, (size_t)h->max_batch * 2048 * sizeof(float)) != hipSuccess) return efail(h, "hipMalloc failed");
        h->allocs.push_back(p);
        h->gap = static_cast<float*>(p);
    }
    h->loaded = true;
    return 0;
}

static void enc_conv(const ConvBN& c, const float* x, const float* res, float* y, int B, int H, int W, int relu, hipStream_t s) {
    EncConv a;
    a.x = x; a.w = c.w; a.scale = c.scale; a.shift = c.shift; a.res = res; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = c.Cin; a.Ho = (H + c.stride - 1) / c.stride; a.Wo = (W + c.stride - 1) / c.stride;
    a.Cout = c.Cout; a.K = c.K; a.stride = c.stride; a.relu = relu;
    const long long M = (long long)B * a.Ho * a.Wo;
    hipLaunchKernelGGL(enc_conv_kernel, dim3((unsigned)((M + 63) / 64), (unsigned)(c.Cout / 64)), dim3(256), 0, s, a);
}

/* x_dev [B,1,H,W] fp32 -> out_dev [B, num_classes] */
int cddpm_encoder_forward(cddpm_encoder_handle h, const float* x_dev, float* out_dev, int B, int H, int W, void* stream) {
    if (!h) return -1;
    if (!h->loaded) return efail(h, "encoder weights not loaded (cddpm_encoder_load_weights)");
    if (!x_dev || !out_dev) return efail(h, "NULL tensor");
    if (B < 1 || B > h->max_batch || H < 32 || W < 32 || H > h->max_h || W > h->max_w)
        return efail(h, "cddpm_encoder_forward: B=%d H=%d W=%d outside the handle's limits (%d, 32..%d x 32..%d)", B, H, W,
                     h->max_batch, h->max_h, h->max_w);
    hipStream_t s = (hipStream_t)stream;
    ECHECK(h, hipSetDevice(h->device));
    // every strided op here (7x7/2 pad 3, 3x3/2 pad 1, 1x1/2 pad 0) maps n -> ceil(n / 2)
    int Hc = (H + 1) / 2, Wc = (W + 1) / 2;
    {
        const long long total = (long long)B * Hc * Wc * 16;
        hipLaunchKernelGGL(enc_stem_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x_dev, h->stem.w, h->stem.scale,
                           h->stem.shift, h->buf0, B, H, W, Hc, Wc);
    }
    {
        const int Ho = (Hc + 1) / 2, Wo = (Wc + 1) / 2;
        const long long total = (long long)B * Ho * Wo * 16;
        hipLaunchKernelGGL(enc_maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h->buf0, h->buf1, B, Hc, Wc, 64, Ho, Wo);
        Hc = Ho; Wc = Wo;
    }
    float* cur = h->buf1;      // block input
    float* t1 = h->buf0; float* t2 = h->buf2; float* t3 = h->buf3;
    for (const Bottleneck& bk : h->blocks) {
        const int st = bk.c2.stride;
        const int Ho = (Hc + st - 1) / st, Wo = (Wc + st - 1) / st;
        enc_conv(bk.c1, cur, nullptr, t1, B, Hc, Wc, 1, s);                 // 1x1 + BN + ReLU
        enc_conv(bk.c2, t1, nullptr, t2, B, Hc, Wc, 1, s);                  // 3x3 (stride) + BN + ReLU
        const float* shortcut = cur;
        if (bk.has_down) { enc_conv(bk.down, cur, nullptr, t1, B, Hc, Wc, 0, s); shortcut = t1; }   // 1x1 (stride) + BN
        enc_conv(bk.c3, t2, shortcut, t3, B, Ho, Wo, 1, s);                 // 1x1 + BN + shortcut + ReLU
        float* tmp = cur; cur = t3; t3 = tmp;
        Hc = Ho; Wc = Wo;
    }
    hipLaunchKernelGGL(enc_avgpool_kernel, dim3((unsigned)((B * 2048 + 255) / 256)), dim3(256), 0, s, cur, h->gap, B, Hc * Wc, 2048);
    hipLaunchKernelGGL(enc_fc_kernel, dim3((unsigned)(B * h->num_classes)), dim3(64), 0, s, h->gap, h->fc_w, h->fc_b, out_dev, B, 2048, h->num_classes);
    ECHECK(h, hipGetLastError());
    return 0;
}

}  // extern "C"
