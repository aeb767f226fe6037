// K1: fused MM-RCA fusion head, forward and backward (CVPR_code/multimodal_model.py:662-726; SelfAttention :39-68;
// ReverseCrossAttention :71-108).  One 256-thread workgroup per sample; every intermediate of the sample lives in
// LDS (<= ~125 KiB), weights (94,820 fp32 = 380 KB for the reference dims) stream from L2.  The whole head is
// 2.9 MFLOP/sample, i.e. launch/HBM-bound: the design goal is ONE launch per direction instead of the reference's
// ~15 eager kernels, and no HBM round trip of the 16x16 attention matrices.
//
// Math is fp32 throughout (the attention tiles are 16x16xd with d in {48..128}: too small to amortise an MFMA
// fragment shuffle, and the head must agree with the reference to <=1e-3).
//
// The backward recomputes each attention block's forward into a shared LDS workspace right before differentiating it
// (cheap; keeps the LDS footprint under 160 KiB), and accumulates parameter gradients with fp32 atomics.
#include "common.h"

#define HP 16          // pseudo-patches (multimodal_model.py:250)
#define SA_HID 128
#define SA_OUT 96
#define CA_HID 64
#define CA_OUT 48
#define CA_FLAT (2 * HP * CA_OUT)   // 1536
#define MAX_CLASSES 16

struct AttnW { const float *wq, *bq, *wk, *bk, *wv, *bv, *g, *b; };
struct AttnG { float *wq, *bq, *wk, *bk, *wv, *bv, *g, *b; };

// workspace of one attention block
struct AttnWS { float *Q, *K, *V, *A, *C, *mean, *rstd; };
__device__ __forceinline__ int attn_ws_floats(int dkq, int dv) { return 2 * HP * dkq + 2 * HP * dv + 256 + 32; }
__device__ __forceinline__ AttnWS carve_ws(float* p, int dkq, int dv) {
  AttnWS w; w.Q = p; w.K = w.Q + HP * dkq; w.V = w.K + HP * dkq; w.A = w.V + HP * dv; w.C = w.A + 256; w.mean = w.C + HP * dv; w.rstd = w.mean + 16; return w;
}

// y[16][out] = x[16][in] W^T + b
__device__ void lin16_fwd(const float* __restrict__ x, int in, const float* __restrict__ W, const float* __restrict__ b,
                          float* __restrict__ y, int out) {
  for (int c = threadIdx.x; c < out; c += blockDim.x) {
    float acc[HP];
    const float bb = b[c];
#pragma unroll
    for (int r = 0; r < HP; ++r) acc[r] = bb;
    const float* w = W + (int64_t)c * in;
    for (int k = 0; k < in; k += 4) {
      const float4 wv = *reinterpret_cast<const float4*>(w + k);
#pragma unroll
      for (int r = 0; r < HP; ++r) {
        const float* xr = x + r * in + k;
        acc[r] += wv.x * xr[0] + wv.y * xr[1] + wv.z * xr[2] + wv.w * xr[3];
      }
    }
#pragma unroll
    for (int r = 0; r < HP; ++r) y[r * out + c] = acc[r];
  }
}

// forward of one attention block into ws; O[16][dv] = relu(LN(A' V)).  Caller syncs before and after.
__device__ void attn_fwd(const float* x1, int din1, const float* x2, int din2, const AttnW& w, int dkq, int dv,
                         bool reverse, const AttnWS& ws, float* O) {
  lin16_fwd(x1, din1, w.wq, w.bq, ws.Q, dkq);
  lin16_fwd(x2, din2, w.wk, w.bk, ws.K, dkq);
  lin16_fwd(x2, din2, w.wv, w.bv, ws.V, dv);
  __syncthreads();
  const int t = threadIdx.x, i = t >> 4, j = t & 15;
  {
    float s = 0.f;
    for (int k = 0; k < dkq; ++k) s += ws.Q[i * dkq + k] * ws.K[j * dkq + k];
    ws.A[t] = s * rsqrtf((float)dkq);
  }
  __syncthreads();
  float a;
  {
    float m = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) m = fmaxf(m, ws.A[i * 16 + jj]);
    float l = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) l += expf(ws.A[i * 16 + jj] - m);
    a = expf(ws.A[t] - m) / l;
  }
  __syncthreads();
  ws.A[t] = a;                       // softmax probabilities (before the reverse map)
  __syncthreads();
  for (int o = t; o < HP * dv; o += 256) {
    const int ii = o / dv, c = o % dv;
    float acc = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      float p = ws.A[ii * 16 + jj];
      if (reverse) p = (1.0f - p) * (1.0f / 15.0f);      // (1-A)/(n-1), n = 16  (multimodal_model.py:95-99)
      acc += p * ws.V[jj * dv + c];
    }
    ws.C[o] = acc;
  }
  __syncthreads();
  if (t < 16) {
    float mu = 0.f;
    for (int c = 0; c < dv; ++c) mu += ws.C[t * dv + c];
    mu /= (float)dv;
    float var = 0.f;
    for (int c = 0; c < dv; ++c) { const float d = ws.C[t * dv + c] - mu; var += d * d; }
    ws.mean[t] = mu;
    ws.rstd[t] = rsqrtf(var / (float)dv + 1e-5f);
  }
  __syncthreads();
  for (int o = t; o < HP * dv; o += 256) {
    const int ii = o / dv, c = o % dv;
    const float y = (ws.C[o] - ws.mean[ii]) * ws.rstd[ii] * w.g[c] + w.b[c];
    O[o] = fmaxf(y, 0.f);
  }
  __syncthreads();
}

// dW[c][k] += sum_r dy[r][c] x[r][k];  db[c] += sum_r dy[r][c];  dx[r][k] += sum_c dy[r][c] W[c][k]
__device__ void lin16_bwd(const float* __restrict__ x, int in, const float* __restrict__ W, const float* __restrict__ dy, int out,
                          float* __restrict__ dx, float* __restrict__ dW, float* __restrict__ db) {
  for (int o = threadIdx.x; o < out * in; o += blockDim.x) {
    const int c = o / in, k = o % in;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < HP; ++r) acc += dy[r * out + c] * x[r * in + k];
    atomicAdd(dW + o, acc);
  }
  for (int c = threadIdx.x; c < out; c += blockDim.x) {
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < HP; ++r) acc += dy[r * out + c];
    atomicAdd(db + c, acc);
  }
  if (dx) {
    for (int o = threadIdx.x; o < HP * in; o += blockDim.x) {
      const int r = o / in, k = o % in;
      float acc = 0.f;
      for (int c = 0; c < out; ++c) acc += dy[r * out + c] * W[(int64_t)c * in + k];
      dx[o] += acc;
    }
  }
}

// backward of one attention block.  ws holds its (recomputed) forward; dO = grad wrt O; G = scratch:
// dC[16*dv] | dS[256] | dQ[16*dkq] | dK[16*dkq] | dV[16*dv] | s1[16] | s2[16].  dx1/dx2 are accumulated.
__device__ void attn_bwd(const float* x1, int din1, const float* x2, int din2, const AttnW& w, const AttnG& gw, int dkq, int dv,
                         bool reverse, const AttnWS& ws, const float* dO, float* G, float* dx1, float* dx2) {
  float* dC = G; float* dS = dC + HP * dv; float* dQ = dS + 256; float* dK = dQ + HP * dkq; float* dV = dK + HP * dkq;
  float* s1 = dV + HP * dv; float* s2 = s1 + 16;
  const int t = threadIdx.x;
  // LayerNorm + ReLU backward.  dC first holds dY = dO * [y>0]
  for (int o = t; o < HP * dv; o += 256) {
    const int ii = o / dv, c = o % dv;
    const float xh = (ws.C[o] - ws.mean[ii]) * ws.rstd[ii];
    const float y = xh * w.g[c] + w.b[c];
    dC[o] = y > 0.f ? dO[o] : 0.f;
  }
  __syncthreads();
  for (int c = t; c < dv; c += 256) {
    float ag = 0.f, ab = 0.f;
#pragma unroll
    for (int ii = 0; ii < HP; ++ii) {
      const float xh = (ws.C[ii * dv + c] - ws.mean[ii]) * ws.rstd[ii];
      ag += dC[ii * dv + c] * xh; ab += dC[ii * dv + c];
    }
    atomicAdd(gw.g + c, ag); atomicAdd(gw.b + c, ab);
  }
  if (t < 16) {
    float a1 = 0.f, a2 = 0.f;
    for (int c = 0; c < dv; ++c) {
      const float xh = (ws.C[t * dv + c] - ws.mean[t]) * ws.rstd[t];
      const float gg = dC[t * dv + c] * w.g[c];
      a1 += gg; a2 += gg * xh;
    }
    s1[t] = a1 / (float)dv; s2[t] = a2 / (float)dv;
  }
  __syncthreads();
  for (int o = t; o < HP * dv; o += 256) {
    const int ii = o / dv, c = o % dv;
    const float xh = (ws.C[o] - ws.mean[ii]) * ws.rstd[ii];
    dC[o] = ws.rstd[ii] * (dC[o] * w.g[c] - s1[ii] - xh * s2[ii]);
  }
  __syncthreads();
  // dA' = dC V^T ; dV = A'^T dC
  const int i = t >> 4, j = t & 15;
  float dA;
  {
    float acc = 0.f;
    for (int c = 0; c < dv; ++c) acc += dC[i * dv + c] * ws.V[j * dv + c];
    dA = reverse ? -acc * (1.0f / 15.0f) : acc;
  }
  dS[t] = dA;
  for (int o = t; o < HP * dv; o += 256) {
    const int jj = o / dv, c = o % dv;
    float acc = 0.f;
#pragma unroll
    for (int ii = 0; ii < HP; ++ii) {
      float p = ws.A[ii * 16 + jj];
      if (reverse) p = (1.0f - p) * (1.0f / 15.0f);
      acc += p * dC[ii * dv + c];
    }
    dV[o] = acc;
  }
  __syncthreads();
  float ds;
  {
    float dot = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) dot += dS[i * 16 + jj] * ws.A[i * 16 + jj];
    ds = ws.A[t] * (dA - dot) * rsqrtf((float)dkq);
  }
  __syncthreads();
  dS[t] = ds;
  __syncthreads();
  for (int o = t; o < HP * dkq; o += 256) {
    const int r = o / dkq, k = o % dkq;
    float aq = 0.f, ak = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) { aq += dS[r * 16 + jj] * ws.K[jj * dkq + k]; ak += dS[jj * 16 + r] * ws.Q[jj * dkq + k]; }
    dQ[o] = aq; dK[o] = ak;
  }
  __syncthreads();
  lin16_bwd(x1, din1, w.wq, dQ, dkq, dx1, gw.wq, gw.bq);
  __syncthreads();          // dx1 and dx2 may alias (self-attention): serialise the read-modify-writes
  lin16_bwd(x2, din2, w.wk, dK, dkq, dx2, gw.wk, gw.bk);
  __syncthreads();
  lin16_bwd(x2, din2, w.wv, dV, dv, dx2, gw.wv, gw.bv);
  __syncthreads();
}

struct HeadDims { int d_img, d_txt, pi, pt, wfull, n_classes, mode, reverse; };

// column range of the concatenated vector [O_c1 | O_c2 | img | txt] that feeds the active classifier, and the
// offset that maps a full column to a classifier-weight column (multimodal_model.py:694-726)
__device__ __forceinline__ void active_cols(const HeadDims& d, int& c0, int& c1, int& woff) {
  if (d.mode == 1) { c0 = CA_FLAT; c1 = d.wfull; woff = CA_FLAT; }       // features_only: [img | txt]
  else if (d.mode == 2) { c0 = 0; c1 = CA_FLAT; woff = 0; }               // cross_attention_only
  else { c0 = 0; c1 = d.wfull; woff = 0; }
}

template <typename T>
__device__ void load_normalised(const T* __restrict__ src, int n, float* dst, float* red, float* norm_out) {
  float ss = 0.f;
  for (int k = threadIdx.x; k < n; k += 256) { const float v = to_f(src[k]); dst[k] = v; ss += v * v; }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
  __syncthreads();
  for (int k = threadIdx.x; k < n; k += 256) dst[k] = dst[k] / nrm;      // x / ||x||_2, no epsilon (:662-665)
  if (threadIdx.x == 0) *norm_out = nrm;
  __syncthreads();
}

// LDS plan (floats): cat[wfull] | O_sai[1536] | O_sat[1536] | WS[max ws] | red[8] | (bwd only:) dcat[wfull] | dO_sai | dO_sat | G
__device__ __forceinline__ int ws_max_floats() { return attn_ws_floats(SA_HID, SA_OUT); }
__device__ __forceinline__ int g_floats() { return 2 * HP * SA_OUT + 256 + 2 * HP * SA_HID + 32; }

template <typename T>
__device__ void head_forward(const T* img, const T* txt, const MmrcaHeadWeights& w, const HeadDims& d, float* cat,
                             float* O_sai, float* O_sat, float* wsbuf, float* red, float* norms) {
  float* xi = cat + CA_FLAT; float* xt = xi + d.d_img;
  load_normalised(img, d.d_img, xi, red, norms + 0);
  load_normalised(txt, d.d_txt, xt, red, norms + 1);
  AttnW sai = {w.sai_wq, w.sai_bq, w.sai_wk, w.sai_bk, w.sai_wv, w.sai_bv, w.sai_g, w.sai_b};
  AttnW sat = {w.sat_wq, w.sat_bq, w.sat_wk, w.sat_bk, w.sat_wv, w.sat_bv, w.sat_g, w.sat_b};
  AttnW c1 = {w.c1_wq, w.c1_bq, w.c1_wk, w.c1_bk, w.c1_wv, w.c1_bv, w.c1_g, w.c1_b};
  AttnW c2 = {w.c2_wq, w.c2_bq, w.c2_wk, w.c2_bk, w.c2_wv, w.c2_bv, w.c2_g, w.c2_b};
  attn_fwd(xt, d.pt, xt, d.pt, sat, SA_HID, SA_OUT, false, carve_ws(wsbuf, SA_HID, SA_OUT), O_sat);     // :677-678
  attn_fwd(xi, d.pi, xi, d.pi, sai, SA_HID, SA_OUT, false, carve_ws(wsbuf, SA_HID, SA_OUT), O_sai);     // :679-680
  attn_fwd(O_sat, SA_OUT, O_sai, SA_OUT, c1, CA_HID, CA_OUT, d.reverse, carve_ws(wsbuf, CA_HID, CA_OUT), cat);                 // T->I :683-684
  attn_fwd(O_sai, SA_OUT, O_sat, SA_OUT, c2, CA_HID, CA_OUT, d.reverse, carve_ws(wsbuf, CA_HID, CA_OUT), cat + HP * CA_OUT);   // I->T :685-686
}

__device__ __forceinline__ float drop_scale(float p, uint64_t seed, int64_t sample, int wcol, int wactive) {
  if (p <= 0.f) return 1.f;
  return mmrca_uniform(seed, (uint64_t)sample * (uint64_t)wactive + (uint64_t)wcol) >= p ? 1.f / (1.f - p) : 0.f;
}

template <typename T>
__global__ void __launch_bounds__(256)
head_fwd_k(const T* __restrict__ img, const T* __restrict__ txt, MmrcaHeadWeights w, float* __restrict__ logits,
           HeadDims d, float drop_p, uint64_t seed) {
  extern __shared__ __attribute__((aligned(16))) float hs[];
  float* cat = hs; float* O_sai = cat + d.wfull; float* O_sat = O_sai + HP * SA_OUT; float* wsbuf = O_sat + HP * SA_OUT;
  float* red = wsbuf + ws_max_floats(); float* norms = red + 4;
  float* part = norms + 4;     // [MAX_CLASSES][4 waves]
  const int64_t b = blockIdx.x;
  head_forward(img + b * d.d_img, txt + b * d.d_txt, w, d, cat, O_sai, O_sat, wsbuf, red, norms);
  int c0, c1, woff; active_cols(d, c0, c1, woff);
  const int wact = c1 - c0;
  float acc[MAX_CLASSES];
#pragma unroll
  for (int k = 0; k < MAX_CLASSES; ++k) acc[k] = 0.f;
  for (int c = c0 + threadIdx.x; c < c1; c += 256) {
    const float v = cat[c] * drop_scale(drop_p, seed, b, c - woff, wact);
#pragma unroll
    for (int k = 0; k < MAX_CLASSES; ++k) if (k < d.n_classes) acc[k] += v * w.fin_w[(int64_t)k * wact + (c - woff)];
  }
#pragma unroll
  for (int k = 0; k < MAX_CLASSES; ++k) {
    if (k < d.n_classes) {
      const float s = wave_sum(acc[k]);
      if ((threadIdx.x & 63) == 0) part[k * 4 + (threadIdx.x >> 6)] = s;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < d.n_classes) {
    const int k = threadIdx.x;
    logits[b * d.n_classes + k] = part[k * 4] + part[k * 4 + 1] + part[k * 4 + 2] + part[k * 4 + 3] + w.fin_b[k];
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
head_bwd_k(const float* __restrict__ dlogits, const T* __restrict__ img, const T* __restrict__ txt, MmrcaHeadWeights w,
           MmrcaHeadGrads gw, T* __restrict__ dimg, T* __restrict__ dtxt, HeadDims d, float drop_p, uint64_t seed) {
  extern __shared__ __attribute__((aligned(16))) float hs[];
  float* cat = hs; float* O_sai = cat + d.wfull; float* O_sat = O_sai + HP * SA_OUT; float* wsbuf = O_sat + HP * SA_OUT;
  float* red = wsbuf + ws_max_floats(); float* norms = red + 4; float* dl = norms + 4;   // dl[MAX_CLASSES]
  float* dcat = dl + MAX_CLASSES; float* dO_sai = dcat + d.wfull; float* dO_sat = dO_sai + HP * SA_OUT; float* G = dO_sat + HP * SA_OUT;
  const int64_t b = blockIdx.x;
  const int t = threadIdx.x;
  head_forward(img + b * d.d_img, txt + b * d.d_txt, w, d, cat, O_sai, O_sat, wsbuf, red, norms);
  if (t < d.n_classes) dl[t] = dlogits[b * d.n_classes + t];
  for (int c = t; c < d.wfull; c += 256) dcat[c] = 0.f;
  for (int c = t; c < HP * SA_OUT; c += 256) { dO_sai[c] = 0.f; dO_sat[c] = 0.f; }
  __syncthreads();
  // classifier backward (dropout mask regenerated from the counter)
  int c0, c1, woff; active_cols(d, c0, c1, woff);
  const int wact = c1 - c0;
  for (int c = c0 + t; c < c1; c += 256) {
    const float sc = drop_scale(drop_p, seed, b, c - woff, wact);
    const float v = cat[c] * sc;
    float g = 0.f;
    for (int k = 0; k < d.n_classes; ++k) {
      g += dl[k] * w.fin_w[(int64_t)k * wact + (c - woff)];
      atomicAdd(gw.fin_w + (int64_t)k * wact + (c - woff), dl[k] * v);
    }
    dcat[c] = g * sc;
  }
  if (t < d.n_classes) atomicAdd(gw.fin_b + t, dl[t]);
  __syncthreads();
  float* xi = cat + CA_FLAT; float* xt = xi + d.d_img;
  float* dxi = dcat + CA_FLAT; float* dxt = dxi + d.d_img;
  AttnW sai = {w.sai_wq, w.sai_bq, w.sai_wk, w.sai_bk, w.sai_wv, w.sai_bv, w.sai_g, w.sai_b};
  AttnW sat = {w.sat_wq, w.sat_bq, w.sat_wk, w.sat_bk, w.sat_wv, w.sat_bv, w.sat_g, w.sat_b};
  AttnW c1w = {w.c1_wq, w.c1_bq, w.c1_wk, w.c1_bk, w.c1_wv, w.c1_bv, w.c1_g, w.c1_b};
  AttnW c2w = {w.c2_wq, w.c2_bq, w.c2_wk, w.c2_bk, w.c2_wv, w.c2_bv, w.c2_g, w.c2_b};
  AttnG gsai = {gw.sai_wq, gw.sai_bq, gw.sai_wk, gw.sai_bk, gw.sai_wv, gw.sai_bv, gw.sai_g, gw.sai_b};
  AttnG gsat = {gw.sat_wq, gw.sat_bq, gw.sat_wk, gw.sat_bk, gw.sat_wv, gw.sat_bv, gw.sat_g, gw.sat_b};
  AttnG gc1 = {gw.c1_wq, gw.c1_bq, gw.c1_wk, gw.c1_bk, gw.c1_wv, gw.c1_bv, gw.c1_g, gw.c1_b};
  AttnG gc2 = {gw.c2_wq, gw.c2_bq, gw.c2_wk, gw.c2_bk, gw.c2_wv, gw.c2_bv, gw.c2_g, gw.c2_b};
  if (d.mode != 1) {     // features_only never sees the attention blocks: their gradients are zero
    AttnWS wc = carve_ws(wsbuf, CA_HID, CA_OUT);
    // cross_attention_2 (I->T): recompute into the workspace (its output lands in G's tail as a dump), then differentiate
    attn_fwd(O_sai, SA_OUT, O_sat, SA_OUT, c2w, CA_HID, CA_OUT, d.reverse, wc, G);
    attn_bwd(O_sai, SA_OUT, O_sat, SA_OUT, c2w, gc2, CA_HID, CA_OUT, d.reverse, wc, dcat + HP * CA_OUT, G, dO_sai, dO_sat);
    attn_fwd(O_sat, SA_OUT, O_sai, SA_OUT, c1w, CA_HID, CA_OUT, d.reverse, wc, G);
    attn_bwd(O_sat, SA_OUT, O_sai, SA_OUT, c1w, gc1, CA_HID, CA_OUT, d.reverse, wc, dcat, G, dO_sat, dO_sai);
    AttnWS wsa = carve_ws(wsbuf, SA_HID, SA_OUT);
    attn_fwd(xt, d.pt, xt, d.pt, sat, SA_HID, SA_OUT, false, wsa, G);
    attn_bwd(xt, d.pt, xt, d.pt, sat, gsat, SA_HID, SA_OUT, false, wsa, dO_sat, G, dxt, dxt);
    attn_fwd(xi, d.pi, xi, d.pi, sai, SA_HID, SA_OUT, false, wsa, G);
    attn_bwd(xi, d.pi, xi, d.pi, sai, gsai, SA_HID, SA_OUT, false, wsa, dO_sai, G, dxi, dxi);
  }
  // y = x/||x||  ->  dx = (dy - y (y.dy)) / ||x||
  for (int which = 0; which < 2; ++which) {
    const int n = which ? d.d_txt : d.d_img;
    const float* y = which ? xt : xi; const float* dy = which ? dxt : dxi;
    T* dst = which ? dtxt : dimg;
    float dot = 0.f;
    for (int k = t; k < n; k += 256) dot += y[k] * dy[k];
    dot = wave_sum(dot);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = dot;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.f / norms[which];
    if (dst) for (int k = t; k < n; k += 256) dst[b * n + k] = from_f<T>((dy[k] - y[k] * tot) * inv);
  }
}

static int head_check(int B, int d_img, int d_txt, int n_classes, int mode) {
  if (B <= 0) return mmrca_fail(-1, "head: B must be positive");
  if (d_img % 64 != 0 || d_txt % 64 != 0 || d_img / HP > 128 || d_txt / HP > 128 || d_img <= 0 || d_txt <= 0)
    return mmrca_fail(-1, "head: feature widths must be multiples of 64 and <= 2048 (d_img=%d d_txt=%d)", d_img, d_txt);
  if (n_classes < 1 || n_classes > MAX_CLASSES) return mmrca_fail(-1, "head: n_classes=%d unsupported (1..%d)", n_classes, MAX_CLASSES);
  if (mode < 0 || mode > 2) return mmrca_fail(-1, "head: bad mode %d", mode);
  return 0;
}

static bool head_weights_ok(const MmrcaHeadWeights* w) {
  const float* const* p = (const float* const*)w;
  for (size_t i = 0; i < sizeof(MmrcaHeadWeights) / sizeof(float*); ++i) if (!p[i]) return false;
  return true;
}

static size_t head_lds_fwd(int wfull) { return (size_t)(wfull + 2 * HP * SA_OUT + (2 * HP * SA_HID + 2 * HP * SA_OUT + 256 + 32) + 8 + 4 * MAX_CLASSES) * 4; }
static size_t head_lds_bwd(int wfull) {
  return (size_t)(wfull + 2 * HP * SA_OUT + (2 * HP * SA_HID + 2 * HP * SA_OUT + 256 + 32) + 8 + MAX_CLASSES + wfull + 2 * HP * SA_OUT +
                  (2 * HP * SA_OUT + 256 + 2 * HP * SA_HID + 32)) * 4;
}

extern "C" int mmrca_head_fwd(const void* img, const void* txt, const MmrcaHeadWeights* w, float* logits,
                              int B, int d_img, int d_txt, int n_classes, int reverse, int mode, float drop_p,
                              uint64_t seed, int dtype, void* stream) {
  MMRCA_REQUIRE(img && txt && w && logits, "head_fwd: null pointer");
  if (int rc = head_check(B, d_img, d_txt, n_classes, mode)) return rc;
  MMRCA_REQUIRE(head_weights_ok(w), "head_fwd: null weight pointer");
  MMRCA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "head_fwd: dropout p must be in [0,1)");
  HeadDims d = {d_img, d_txt, d_img / HP, d_txt / HP, CA_FLAT + d_img + d_txt, n_classes, mode, reverse ? 1 : 0};
  const size_t lds = head_lds_fwd(d.wfull);
  MMRCA_REQUIRE(lds <= 160 * 1024, "head_fwd: LDS budget exceeded");
  MMRCA_DISPATCH_DTYPE(dtype, "head_fwd",
    hipFuncSetAttribute((const void*)head_fwd_k<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(head_fwd_k<T>, dim3(B), dim3(256), lds, (hipStream_t)stream, (const T*)img, (const T*)txt, *w, logits, d, drop_p, seed);)
  MMRCA_CHECK_LAUNCH("head_fwd");
  return 0;
}

extern "C" int mmrca_head_bwd(const float* dlogits, const void* img, const void* txt, const MmrcaHeadWeights* w,
                              const MmrcaHeadGrads* g, void* dimg, void* dtxt, int B, int d_img, int d_txt, int n_classes,
                              int reverse, int mode, float drop_p, uint64_t seed, int dtype, void* stream) {
  MMRCA_REQUIRE(dlogits && w && g && img && txt, "head_bwd: null pointer");
  if (int rc = head_check(B, d_img, d_txt, n_classes, mode)) return rc;
  MMRCA_REQUIRE(head_weights_ok(w), "head_bwd: null weight pointer");
  {
    float* const* p = (float* const*)g;
    for (size_t i = 0; i < sizeof(MmrcaHeadGrads) / sizeof(float*); ++i) MMRCA_REQUIRE(p[i], "head_bwd: null gradient pointer");
  }
  HeadDims d = {d_img, d_txt, d_img / HP, d_txt / HP, CA_FLAT + d_img + d_txt, n_classes, mode, reverse ? 1 : 0};
  const size_t lds = head_lds_bwd(d.wfull);
  MMRCA_REQUIRE(lds <= 160 * 1024, "head_bwd: LDS budget exceeded");
  MMRCA_DISPATCH_DTYPE(dtype, "head_bwd",
    hipFuncSetAttribute((const void*)head_bwd_k<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(head_bwd_k<T>, dim3(B), dim3(256), lds, (hipStream_t)stream, dlogits, (const T*)img, (const T*)txt, *w, *g,
                       (T*)dimg, (T*)dtxt, d, drop_p, seed);)
  MMRCA_CHECK_LAUNCH("head_bwd");
  return 0;
}
