// K1: fused MM-RCA fusion head, forward and backward (CVPR_code/multimodal_model.py:662-726; SelfAttention :39-68;
// ReverseCrossAttention :71-108).
//
// One 512-thread workgroup per sample.  Waves 0-3 ("text side") and 4-7 ("image side") run the two independent
// attention blocks of each stage side by side (self_attention_text || self_attention_image, then cross_attention_1 ||
// cross_attention_2), so the dependent chain is two blocks long instead of four.  Every intermediate of the sample
// lives in LDS; the weights (94,820 fp32 = 380 KB for the reference dims) stream from L2.
//
// Math is fp32 throughout (the head must agree with the reference to <= 1e-3), on the fp32 matrix cores:
// v_mfma_f32_16x16x4_f32 -- a sample's 16 pseudo-patches are exactly one MFMA tile high, so each projection,
// score matrix, A'V product and their transposes in the backward is a handful of 16x16 tiles.  The softmax, the
// LayerNorm + ReLU and their backward run in the accumulator layout of one wave (a row is 16 lanes wide: DPP-range
// shuffles), so an attention block costs two workgroup barriers in the forward and three in the backward.
//
// Backward = two launches.  head_bwd_k (per sample) recomputes the forward in LDS, differentiates it, and writes the
// per-sample projection gradients dQ|dK|dV, the projection inputs and the dropped classifier input to a workspace;
// head_wgrad_k then forms every weight gradient as a [out x in] GEMM over all B*16 rows (64x64 blocks of 16 MFMA
// tiles per wave, 4 waves of a workgroup splitting 256 rows and reducing in LDS, one atomic per address per 256 rows)
// instead of one atomic per weight per SAMPLE (24 M atomics at B=256) as the first version did.
#include "common.h"
#include <stdlib.h>

#define HP 16          // pseudo-patches (multimodal_model.py:250)
#define SA_HID 128
#define SA_OUT 96
#define CA_HID 64
#define CA_OUT 48
#define CA_FLAT (2 * HP * CA_OUT)   // 1536
#define SA_FLAT (HP * SA_OUT)       // 1536
#define MAX_CLASSES 16
#define HEAD_THREADS 512
#define MAXKB 8                     // projection inputs are at most 128 wide (feature widths <= 2048)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct AttnW { const float *wq, *bq, *wk, *bk, *wv, *bv, *g, *b; };
struct AttnG { float *wq, *bq, *wk, *bk, *wv, *bv, *g, *b; };

// attention workspace of one wave group (floats; sized for the self-attention dims, padded row strides)
#define WS_Q 0
#define WS_K (HP * (SA_HID + 4))
#define WS_V (2 * HP * (SA_HID + 4))
#define WS_P (WS_V + HP * (SA_OUT + 4))
#define WS_C (WS_P + 256)
#define WS_MEAN (WS_C + HP * (SA_OUT + 4))
#define WS_RSTD (WS_MEAN + 16)
#define WS_FLOATS (WS_RSTD + 16)
// backward scratch of one wave group
#define G_DQ 0
#define G_DK (HP * (SA_HID + 4))
#define G_DV (2 * HP * (SA_HID + 4))
#define G_DS (G_DV + HP * (SA_OUT + 4))
#define G_FLOATS (G_DS + 256)
#define MISC_FLOATS 256             // red[16] | norms[2] | pad | dl[16] @32 | part[16*8] @64

struct HeadDims { int d_img, d_txt, pi, pt, wfull, n_classes, mode, reverse, stop; };   // stop: timing ablation only (MMRCA_HEAD_STOP)

// LDS plan, offsets in floats
struct HeadLayout {
  int oc, doc;       // O_c = [O_c1 | O_c2] (1536) and its gradient; in the backward dO_sa = [text | image] aliases both
  int x, dx;         // normalised [img | txt] features and their gradient
  int osa;           // O_sa = [text | image], 16 x 96 each
  int ws, g, misc;
  int total;
};
static HeadLayout head_layout(const HeadDims& d, bool bwd) {
  HeadLayout l; int o = 0;
  l.oc = o; o += CA_FLAT;
  l.doc = o; if (bwd) o += CA_FLAT;
  l.x = o; o += d.d_img + d.d_txt;
  l.dx = o; if (bwd) o += d.d_img + d.d_txt;
  l.osa = o; o += 2 * SA_FLAT;
  l.ws = o; o += 2 * WS_FLOATS;
  l.g = o; if (bwd) o += 2 * G_FLOATS;
  l.misc = o; o += MISC_FLOATS;
  l.total = o;
  return l;
}

// workspace of the backward (floats per array; R = 16 B rows)
struct HeadScratch {
  float *xt, *xi, *osat, *osai, *v;              // [R][pt] [R][pi] [R][96] [R][96] [B][wfull]
  float *dy_sa[2];                               // [R][2*128+96]  (text, image)
  float *dq_ca[2], *dkv_ca[2];                   // [R][64], [R][64+48]  (cross_attention_1, _2)
};
#define SA_DY (2 * SA_HID + SA_OUT)   // 352
#define CA_DKV (CA_HID + CA_OUT)      // 112
#define WG_ROWS 256                // rows of one weight-gradient workgroup (64 per wave)
// 64x64 blocks of the six weight-gradient GEMMs: 2 x 6 x ceil(p/64) (self) + 2 x (1 + 2) x 2 (cross)
static int wg_blocks(int d_img, int d_txt) { return 6 * ((d_img / HP + 63) / 64) + 6 * ((d_txt / HP + 63) / 64) + 12; }
static int wg_ksplit(int B) { return (B * HP + WG_ROWS - 1) / WG_ROWS; }
static int64_t head_scratch_floats(int B, int d_img, int d_txt) {
  return (int64_t)B * (d_img + d_txt + 2 * SA_FLAT + (CA_FLAT + d_img + d_txt) + HP * (2 * SA_DY + 2 * CA_HID + 2 * CA_DKV));
}
static HeadScratch carve_scratch(float* p, int B, const HeadDims& d) {
  HeadScratch s; const int64_t R = (int64_t)B * HP;
  s.xt = p; p += (int64_t)B * d.d_txt;
  s.xi = p; p += (int64_t)B * d.d_img;
  s.osat = p; p += R * SA_OUT;
  s.osai = p; p += R * SA_OUT;
  s.v = p; p += (int64_t)B * d.wfull;
  for (int m = 0; m < 2; ++m) { s.dy_sa[m] = p; p += R * SA_DY; }
  for (int m = 0; m < 2; ++m) { s.dq_ca[m] = p; p += R * CA_HID; s.dkv_ca[m] = p; p += R * CA_DKV; }
  return s;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// v_mfma_f32_16x16x4_f32 operand layout: lane l supplies A[l%16][l/16] and B[l/16][l%16]; it receives
// D[4*(l/16) + r][l%16] in register r.  A lane that holds four consecutive k of its row (a 16-byte load) feeds four
// MFMAs: the k slot of MFMA m is 4*(l/16)+m for both operands, which is a permutation of the 16 k of the chunk.
__device__ __forceinline__ f32x4 mfma_k16(f32x4 a, f32x4 b, f32x4 c) {
#pragma unroll
  for (int m = 0; m < 4; ++m) c = mfma4(a[m], b[m], c);
  return c;
}
// reductions over the 16 lanes that share an accumulator row (= one DPP row): quad swaps, half-row mirror, row mirror.
// Every lane ends with the row's result; these are VALU-rate, where ds_bpermute shuffles cost an LDS round trip each.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum(float v) {
  v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);     // row_half_mirror
  v += dpp_f<0x140>(v);     // row_mirror
  return v;
}
__device__ __forceinline__ float row_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  return v;
}

// A-operand fragments of a row-major [16][in] matrix: a[kb] = x[lane%16][16 kb + 4 (lane/16) .. +3], zero past `in`
template <int NKB>
__device__ __forceinline__ void load_rows16(const float* x, int ld, int in, int lane, f32x4 (&a)[NKB]) {
  const float* p = x + (lane & 15) * ld + 4 * (lane >> 4);
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    a[kb] = 16 * kb + 4 * (lane >> 4) < in ? ld4(p + 16 * kb) : z;
  }
}

struct ProjTile { const float* W; const float* bias; float* dst; int ld, c0, from_x1; };
template <int DKQ, int DV>
__device__ __forceinline__ ProjTile proj_tile(int t, const AttnW& w, float* ws) {
  constexpr int NQ = DKQ / 16;
  ProjTile p;
  if (t < NQ) { p.W = w.wq; p.bias = w.bq; p.dst = ws + WS_Q; p.ld = DKQ + 4; p.c0 = 16 * t; p.from_x1 = 1; }
  else if (t < 2 * NQ) { p.W = w.wk; p.bias = w.bk; p.dst = ws + WS_K; p.ld = DKQ + 4; p.c0 = 16 * (t - NQ); p.from_x1 = 0; }
  else { p.W = w.wv; p.bias = w.bv; p.dst = ws + WS_V; p.ld = DV + 4; p.c0 = 16 * (t - 2 * NQ); p.from_x1 = 0; }
  return p;
}

// Q = x1 Wq^T + bq, K = x2 Wk^T + bk, V = x2 Wv^T + bv into the workspace: (2 DKQ + DV)/16 column tiles over the four
// waves of the group.  The phase is L2-latency bound (12-24 MFMAs per tile against a ~1 us weight fetch), so a wave
// puts the weight rows of TPB of its tiles in flight at once.  NKB = 16-wide chunks of the input width.
template <int DKQ, int DV, int NKB, int TPB, bool SAME>
__device__ __forceinline__ void proj_qkv_t(const float* x1, const float* x2, int in, const AttnW& w, float* ws, int gwave, int lane) {
  constexpr int NTILE = (2 * DKQ + DV) / 16;
  f32x4 a1[NKB], a2[SAME ? 1 : NKB];
  load_rows16<NKB>(x1, in, in, lane, a1);
  if (!SAME) load_rows16<SAME ? 1 : NKB>(x2, in, in, lane, a2);
  for (int t0 = gwave; t0 < NTILE; t0 += 4 * TPB) {
    f32x4 bw[TPB][NKB]; float bias[TPB];
#pragma unroll
    for (int i = 0; i < TPB; ++i) {
      const int t = t0 + 4 * i;
      if (t < NTILE) {
        const ProjTile p = proj_tile<DKQ, DV>(t, w, ws);
        load_rows16<NKB>(p.W + (int64_t)p.c0 * in, in, in, lane, bw[i]);
        bias[i] = p.bias[p.c0 + (lane & 15)];
      }
    }
#pragma unroll
    for (int i = 0; i < TPB; ++i) {
      const int t = t0 + 4 * i;
      if (t < NTILE) {
        const ProjTile p = proj_tile<DKQ, DV>(t, w, ws);
        f32x4 acc4[4] = {};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          const f32x4 a = (SAME || p.from_x1) ? a1[kb] : a2[SAME ? 0 : kb];
#pragma unroll
          for (int m = 0; m < 4; ++m) acc4[m] = mfma4(a[m], bw[i][kb][m], acc4[m]);
        }
        const f32x4 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]) + bias[i];
        float* dst = p.dst + (4 * (lane >> 4)) * p.ld + p.c0 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[r * p.ld] = acc[r];
      }
    }
  }
}
// self-attention: x1 = x2, input width d/16 in {48, 64, 80, 128, ...}
__device__ void proj_self(const float* x, int in, const AttnW& w, float* ws, int gwave, int lane) {
  const int nkb = (in + 15) >> 4;
  if (nkb <= 3) proj_qkv_t<SA_HID, SA_OUT, 3, 6, true>(x, x, in, w, ws, gwave, lane);
  else if (nkb == 4) proj_qkv_t<SA_HID, SA_OUT, 4, 6, true>(x, x, in, w, ws, gwave, lane);
  else if (nkb <= 6) proj_qkv_t<SA_HID, SA_OUT, 6, 3, true>(x, x, in, w, ws, gwave, lane);
  else proj_qkv_t<SA_HID, SA_OUT, 8, 3, true>(x, x, in, w, ws, gwave, lane);
}
__device__ void proj_cross(const float* x1, const float* x2, const AttnW& w, float* ws, int gwave, int lane) {
  proj_qkv_t<CA_HID, CA_OUT, SA_OUT / 16, 3, false>(x1, x2, SA_OUT, w, ws, gwave, lane);
}

// scores, softmax, (reverse map,) A'V, LayerNorm, ReLU of one attention block in ONE wave; O: [16][DV] row-major.
template <int DKQ, int DV, bool SAVE>
__device__ void attn_core(float* ws, const AttnW& w, bool reverse, float* O, int lane) {
  constexpr int QS = DKQ + 4, VS = DV + 4, NV = DV / 16;
  const int x = lane & 15, g = lane >> 4;
  const float* Q = ws + WS_Q; const float* K = ws + WS_K; const float* V = ws + WS_V; float* P = ws + WS_P;
  f32x4 s4[4] = {};                          // four independent accumulators: a dependent MFMA chain idles the pipe
#pragma unroll
  for (int kb = 0; kb < DKQ / 16; ++kb) {
    const f32x4 a = ld4(Q + x * QS + 16 * kb + 4 * g), bk = ld4(K + x * QS + 16 * kb + 4 * g);
#pragma unroll
    for (int m = 0; m < 4; ++m) s4[m] = mfma4(a[m], bk[m], s4[m]);
  }
  const f32x4 s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  const float scale = rsqrtf((float)DKQ);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = s[r] * scale;
    const float m = row_max(v);
    const float e = expf(v - m);
    const float l = row_sum(e);
    P[(4 * g + r) * 16 + x] = e / l;           // softmax probabilities (before the reverse map)
  }
  f32x4 pa = ld4(P + x * 16 + 4 * g);          // same wave wrote it: LDS operations of a wave complete in order
  if (reverse) pa = (1.0f - pa) * (1.0f / 15.0f);      // (1-A)/(n-1), n = 16  (multimodal_model.py:95-99)
  f32x4 c[NV];
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) acc = mfma4(pa[m], V[(4 * g + m) * VS + 16 * t + x], acc);
    c[t] = acc;
  }
  float gam[NV], bet[NV];
#pragma unroll
  for (int t = 0; t < NV; ++t) { gam[t] = w.g[16 * t + x]; bet[t] = w.b[16 * t + x]; }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float su = 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) su += c[t][r];
    const float mu = row_sum(su) * (1.0f / DV);
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) { const float dlt = c[t][r] - mu; sq += dlt * dlt; }
    const float rstd = rsqrtf(row_sum(sq) * (1.0f / DV) + 1e-5f);
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      const float y = (c[t][r] - mu) * rstd * gam[t] + bet[t];
      O[(4 * g + r) * DV + 16 * t + x] = fmaxf(y, 0.f);
      if (SAVE) ws[WS_C + (4 * g + r) * VS + 16 * t + x] = c[t][r];
    }
    if (SAVE && x == 0) { ws[WS_MEAN + 4 * g + r] = mu; ws[WS_RSTD + 4 * g + r] = rstd; }
  }
}

// backward, step 1 (one wave): LayerNorm + ReLU backward (dC overwrites C in the workspace), d gamma / d beta,
// dA' = dC V^T, softmax (and reverse map) backward -> dS in G.
template <int DKQ, int DV>
__device__ void attn_bwd_core(float* ws, float* G, const AttnW& w, const AttnG& gw, bool reverse, const float* dO, int lane) {
  constexpr int VS = DV + 4, NV = DV / 16;
  const int x = lane & 15, g = lane >> 4;
  float* C = ws + WS_C; const float* V = ws + WS_V; const float* P = ws + WS_P;
  float gam[NV], dgam[NV], dbet[NV];
#pragma unroll
  for (int t = 0; t < NV; ++t) { gam[t] = w.g[16 * t + x]; dgam[t] = 0.f; dbet[t] = 0.f; }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * g + r;
    const float mu = ws[WS_MEAN + row], rstd = ws[WS_RSTD + row];
    float xh[NV], gg[NV];
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      const int col = 16 * t + x;
      xh[t] = (C[row * VS + col] - mu) * rstd;
      const float y = xh[t] * gam[t] + w.b[col];
      const float dy = y > 0.f ? dO[row * DV + col] : 0.f;
      dgam[t] += dy * xh[t]; dbet[t] += dy;
      gg[t] = dy * gam[t];
      a1 += gg[t]; a2 += gg[t] * xh[t];
    }
    const float s1 = row_sum(a1) * (1.0f / DV), s2 = row_sum(a2) * (1.0f / DV);
#pragma unroll
    for (int t = 0; t < NV; ++t) C[row * VS + 16 * t + x] = rstd * (gg[t] - s1 - xh[t] * s2);
  }
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    float a = dgam[t], b = dbet[t];
    a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
    b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
    if (g == 0) { atomicAdd(gw.g + 16 * t + x, a); atomicAdd(gw.b + 16 * t + x, b); }
  }
  f32x4 dp4[4] = {};
#pragma unroll
  for (int kb = 0; kb < NV; ++kb) {
    const f32x4 a = ld4(C + x * VS + 16 * kb + 4 * g), bv = ld4(V + x * VS + 16 * kb + 4 * g);
#pragma unroll
    for (int m = 0; m < 4; ++m) dp4[m] = mfma4(a[m], bv[m], dp4[m]);
  }
  const f32x4 dp = (dp4[0] + dp4[1]) + (dp4[2] + dp4[3]);
  const float scale = rsqrtf((float)DKQ);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float dA = reverse ? -dp[r] * (1.0f / 15.0f) : dp[r];
    const float p = P[(4 * g + r) * 16 + x];
    const float dot = row_sum(dA * p);
    G[G_DS + (4 * g + r) * 16 + x] = p * (dA - dot) * scale;
  }
}

// backward, step 2 (four waves): dV = A'^T dC, dQ = dS K, dK = dS^T Q as 16x16 tiles into G and the global workspace.
template <int DKQ, int DV>
__device__ void attn_bwd_qkv(const float* ws, float* G, bool reverse, float* gq, int ldq, float* gk, int ldk, float* gv, int ldv,
                             int gwave, int lane) {
  constexpr int QS = DKQ + 4, VS = DV + 4, NQ = DKQ / 16, NV = DV / 16;
  const int x = lane & 15, g = lane >> 4;
  const float* P = ws + WS_P; const float* dS = G + G_DS;
  for (int t = gwave; t < 2 * NQ + NV; t += 4) {
    f32x4 a; const float* Bm; int ldb, c0; float* dst; int ldd; float* gdst; int ldg;
    if (t < NV) {              // dV[j][c] = sum_i A'[i][j] dC[i][c]
#pragma unroll
      for (int m = 0; m < 4; ++m) { const float p = P[(4 * g + m) * 16 + x]; a[m] = reverse ? (1.0f - p) * (1.0f / 15.0f) : p; }
      Bm = ws + WS_C; ldb = VS; c0 = 16 * t; dst = G + G_DV; ldd = VS; gdst = gv; ldg = ldv;
    } else if (t < NV + NQ) {  // dQ[i][c] = sum_j dS[i][j] K[j][c]
      a = ld4(dS + x * 16 + 4 * g);
      Bm = ws + WS_K; ldb = QS; c0 = 16 * (t - NV); dst = G + G_DQ; ldd = QS; gdst = gq; ldg = ldq;
    } else {                   // dK[j][c] = sum_i dS[i][j] Q[i][c]
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = dS[(4 * g + m) * 16 + x];
      Bm = ws + WS_Q; ldb = QS; c0 = 16 * (t - NV - NQ); dst = G + G_DK; ldd = QS; gdst = gk; ldg = ldk;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) acc = mfma4(a[m], Bm[(4 * g + m) * ldb + c0 + x], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dst[(4 * g + r) * ldd + c0 + x] = acc[r];
      gdst[(int64_t)(4 * g + r) * ldg + c0 + x] = acc[r];
    }
  }
}

// backward, step 3 (four waves): dx1 += dQ Wq, dx2 += dK Wk + dV Wv.  A wave OWNS 16-column tiles of the result and
// walks the whole weight column block of every projection that feeds it, so the sum needs no atomics (LDS float
// atomics retire one lane at a time: ~0.5 us per instruction, measured) and no barrier between the projections.
template <int NKB>
__device__ __forceinline__ void dx_acc(const float* dY, int ldy, const float* W, int in, int k0, int lane, f32x4 (&acc)[4]) {
  const int x = lane & 15, g = lane >> 4;
  const bool ok = k0 + x < in;
  float b[NKB][4];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int m = 0; m < 4; ++m) b[kb][m] = ok ? W[(int64_t)(16 * kb + 4 * g + m) * in + k0 + x] : 0.f;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const f32x4 a = ld4(dY + x * ldy + 16 * kb + 4 * g);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = mfma4(a[m], b[kb][m], acc[m]);
  }
}
__device__ __forceinline__ void dx_store(float* dx, int in, int k0, int lane, const f32x4 (&acc)[4]) {
  const int x = lane & 15, g = lane >> 4;
  if (k0 + x < in) {
    const f32x4 v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) dx[(4 * g + r) * in + k0 + x] += v[r];
  }
}
// which: 1 = query projection -> dx1, 2 = key + value projections -> dx2, 3 = all three into dx1 (self-attention)
template <int DKQ, int DV>
__device__ void attn_bwd_dx(const float* G, const AttnW& w, int in, float* dx, int which, int gwave, int lane) {
  const int nt = (in + 15) >> 4;
  for (int t = gwave; t < nt; t += 4) {
    f32x4 acc[4] = {};
    if (which & 1) dx_acc<DKQ / 16>(G + G_DQ, DKQ + 4, w.wq, in, 16 * t, lane, acc);
    if (which & 2) {
      dx_acc<DKQ / 16>(G + G_DK, DKQ + 4, w.wk, in, 16 * t, lane, acc);
      dx_acc<DV / 16>(G + G_DV, DV + 4, w.wv, in, 16 * t, lane, acc);
    }
    dx_store(dx, in, 16 * t, lane, acc);
  }
}

// column range of the concatenated vector [O_c1 | O_c2 | img | txt] that feeds the active classifier, and the
// offset that maps a full column to a classifier-weight column (multimodal_model.py:694-726)
__device__ __forceinline__ void active_cols(const HeadDims& d, int& c0, int& c1, int& woff) {
  if (d.mode == 1) { c0 = CA_FLAT; c1 = d.wfull; woff = CA_FLAT; }       // features_only: [img | txt]
  else if (d.mode == 2) { c0 = 0; c1 = CA_FLAT; woff = 0; }               // cross_attention_only
  else { c0 = 0; c1 = d.wfull; woff = 0; }
}

__device__ __forceinline__ float drop_scale(float p, uint64_t seed, int64_t sample, int wcol, int wactive) {
  if (p <= 0.f) return 1.f;
  return mmrca_uniform(seed, (uint64_t)sample * (uint64_t)wactive + (uint64_t)wcol) >= p ? 1.f / (1.f - p) : 0.f;
}

// [0] text side: self_attention_text, cross_attention_1; [1] image side: self_attention_image, cross_attention_2.
// Selected field by field (an array indexed by the wave's side would live in scratch memory).
#define SIDE_SEL(T, pre0, pre1) T{side ? w.pre1##_wq : w.pre0##_wq, side ? w.pre1##_bq : w.pre0##_bq, side ? w.pre1##_wk : w.pre0##_wk, \
                                  side ? w.pre1##_bk : w.pre0##_bk, side ? w.pre1##_wv : w.pre0##_wv, side ? w.pre1##_bv : w.pre0##_bv, \
                                  side ? w.pre1##_g : w.pre0##_g, side ? w.pre1##_b : w.pre0##_b}
struct HeadBlocks { AttnW sa, ca; };
__device__ __forceinline__ HeadBlocks head_blocks(const MmrcaHeadWeights& w, int side) {
  HeadBlocks b;
  b.sa = SIDE_SEL(AttnW, sat, sai);
  b.ca = SIDE_SEL(AttnW, c1, c2);
  return b;
}
__device__ __forceinline__ AttnG grad_block(const MmrcaHeadGrads& w, int stage, int side) {
  return stage == 0 ? SIDE_SEL(AttnG, sat, sai) : SIDE_SEL(AttnG, c1, c2);
}

// x / ||x||_2 (no epsilon, :662-665) of both modalities: the text side loads txt, the image side img
template <typename T>
__device__ void load_normalised(const T* img, const T* txt, const HeadDims& d, float* xs, float* misc, int side, int gt) {
  const T* src = side ? img : txt;
  const int n = side ? d.d_img : d.d_txt;
  float* dst = side ? xs : xs + d.d_img;
  float ss = 0.f;
  for (int k = 4 * gt; k < n; k += 4 * 256) {
    const Vec4<T> v = Vec4<T>::load(src + k);
#pragma unroll
    for (int e = 0; e < 4; ++e) { dst[k + e] = v.v[e]; ss += v.v[e] * v.v[e]; }
  }
  ss = wave_sum(ss);
  if ((gt & 63) == 0) misc[side * 4 + (gt >> 6)] = ss;
  __syncthreads();
  const float nrm = sqrtf(misc[side * 4] + misc[side * 4 + 1] + misc[side * 4 + 2] + misc[side * 4 + 3]);
  for (int k = gt; k < n; k += 256) dst[k] = dst[k] / nrm;
  if (gt == 0) misc[16 + side] = nrm;
  __syncthreads();
}

// forward of both stages; SAVE keeps the cross-attention internals for the backward.  Ends with a barrier.
template <bool SAVE_CA>
__device__ void head_attention_fwd(const HeadBlocks& hb, const HeadDims& d, const HeadLayout& L, float* hs, int side, int gwave, int lane) {
  float* ws = hs + L.ws + side * WS_FLOATS;
  float* xs = hs + L.x;
  const float* xm = side ? xs : xs + d.d_img;
  const int pm = side ? d.pi : d.pt;
  float* osa_own = hs + L.osa + side * SA_FLAT;
  const float* osa_other = hs + L.osa + (1 - side) * SA_FLAT;
  proj_self(xm, pm, hb.sa, ws, gwave, lane);                                   // :677-680
  __syncthreads();
  if (gwave == 0) attn_core<SA_HID, SA_OUT, false>(ws, hb.sa, false, osa_own, lane);
  __syncthreads();
  // cross_attention_1 (T->I, :683-684): queries from O_sat, keys/values from O_sai; cross_attention_2 the other way
  proj_cross(osa_own, osa_other, hb.ca, ws, gwave, lane);
  __syncthreads();
  if (gwave == 0) attn_core<CA_HID, CA_OUT, SAVE_CA>(ws, hb.ca, d.reverse, hs + L.oc + side * (HP * CA_OUT), lane);
  __syncthreads();
}

__device__ __forceinline__ float cat_at(const float* hs, const HeadLayout& L, int c) { return c < CA_FLAT ? hs[L.oc + c] : hs[L.x + c - CA_FLAT]; }

template <typename T>
__global__ void __launch_bounds__(HEAD_THREADS)
head_fwd_k(const T* __restrict__ img, const T* __restrict__ txt, MmrcaHeadWeights w, float* __restrict__ logits,
           HeadDims d, HeadLayout L, float drop_p, uint64_t seed) {
  extern __shared__ __attribute__((aligned(16))) float hs[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, side = wave >> 2, gwave = wave & 3, gt = t & 255;
  const int64_t b = blockIdx.x;
  float* misc = hs + L.misc;
  load_normalised(img + b * d.d_img, txt + b * d.d_txt, d, hs + L.x, misc, side, gt);
  if (d.mode != 1) {           // features_only never looks at the attention blocks
    const HeadBlocks hb = head_blocks(w, side);
    head_attention_fwd<false>(hb, d, L, hs, side, gwave, lane);
  }
  int c0, c1, woff; active_cols(d, c0, c1, woff);
  const int wact = c1 - c0;
  float acc[MAX_CLASSES];
#pragma unroll
  for (int k = 0; k < MAX_CLASSES; ++k) acc[k] = 0.f;
  for (int c = c0 + t; c < c1; c += HEAD_THREADS) {
    const float v = cat_at(hs, L, c) * drop_scale(drop_p, seed, b, c - woff, wact);
#pragma unroll
    for (int k = 0; k < MAX_CLASSES; ++k) if (k < d.n_classes) acc[k] += v * w.fin_w[(int64_t)k * wact + (c - woff)];
  }
  float* part = misc + 64;     // [MAX_CLASSES][8 waves]
#pragma unroll
  for (int k = 0; k < MAX_CLASSES; ++k) {
    if (k < d.n_classes) {
      const float s = wave_sum(acc[k]);
      if (lane == 0) part[k * 8 + wave] = s;
    }
  }
  __syncthreads();
  if (t < d.n_classes) {
    float s = w.fin_b[t];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += part[t * 8 + i];
    logits[b * d.n_classes + t] = s;
  }
}

#define STOP(n) if (d.stop == (n)) return
template <typename T>
__global__ void __launch_bounds__(HEAD_THREADS)
head_bwd_k(const float* __restrict__ dlogits, const T* __restrict__ img, const T* __restrict__ txt, MmrcaHeadWeights w,
           MmrcaHeadGrads gw, T* __restrict__ dimg, T* __restrict__ dtxt, HeadDims d, HeadLayout L, HeadScratch sc,
           float drop_p, uint64_t seed) {
  extern __shared__ __attribute__((aligned(16))) float hs[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, side = wave >> 2, gwave = wave & 3, gt = t & 255;
  const int64_t b = blockIdx.x;
  float* misc = hs + L.misc;
  float* dl = misc + 32;
  float* xs = hs + L.x; float* dxs = hs + L.dx;
  load_normalised(img + b * d.d_img, txt + b * d.d_txt, d, xs, misc, side, gt);
  const HeadBlocks hb = head_blocks(w, side);
  if (d.mode != 1) {
    head_attention_fwd<true>(hb, d, L, hs, side, gwave, lane);
    // projection inputs of the weight-gradient GEMMs
    for (int k = 4 * t; k < d.d_img; k += 4 * HEAD_THREADS) *reinterpret_cast<f32x4*>(sc.xi + b * d.d_img + k) = ld4(xs + k);
    for (int k = 4 * t; k < d.d_txt; k += 4 * HEAD_THREADS) *reinterpret_cast<f32x4*>(sc.xt + b * d.d_txt + k) = ld4(xs + d.d_img + k);
    for (int k = 4 * t; k < SA_FLAT; k += 4 * HEAD_THREADS) {
      *reinterpret_cast<f32x4*>(sc.osat + b * SA_FLAT + k) = ld4(hs + L.osa + k);
      *reinterpret_cast<f32x4*>(sc.osai + b * SA_FLAT + k) = ld4(hs + L.osa + SA_FLAT + k);
    }
  }
  if (t < d.n_classes) { const float v = dlogits[b * d.n_classes + t]; dl[t] = v; atomicAdd(gw.fin_b + t, v); }
  __syncthreads(); STOP(1);
  // classifier backward (dropout mask regenerated from the counter); v = dropped input goes to the workspace
  int c0, c1, woff; active_cols(d, c0, c1, woff);
  const int wact = c1 - c0;
  for (int c = t; c < d.wfull; c += HEAD_THREADS) {
    float gsum = 0.f;
    if (c >= c0 && c < c1) {
      const float s = drop_scale(drop_p, seed, b, c - woff, wact);
      sc.v[b * d.wfull + (c - woff)] = cat_at(hs, L, c) * s;
      for (int k = 0; k < d.n_classes; ++k) gsum += dl[k] * w.fin_w[(int64_t)k * wact + (c - woff)];
      gsum *= s;
    }
    if (c < CA_FLAT) hs[L.doc + c] = gsum; else dxs[c - CA_FLAT] = gsum;
  }
  __syncthreads(); STOP(2);
  if (d.mode != 1) {           // features_only never sees the attention blocks: their gradients stay zero
    float* ws = hs + L.ws + side * WS_FLOATS;
    float* G = hs + L.g + side * G_FLOATS;
    const float* xm = side ? xs : xs + d.d_img;
    float* dxm = side ? dxs : dxs + d.d_img;
    const int pm = side ? d.pi : d.pt;
    float* osa_own = hs + L.osa + side * SA_FLAT;
    float* dosa = hs + L.oc;                       // [text | image]: aliases O_c | dO_c once those are consumed
    const int64_t row0 = b * HP;
    // ---- cross stage
    const AttnG gca = grad_block(gw, 1, side);
    if (gwave == 0) attn_bwd_core<CA_HID, CA_OUT>(ws, G, hb.ca, gca, d.reverse, hs + L.doc + side * (HP * CA_OUT), lane);
    __syncthreads(); STOP(3);
    for (int k = t; k < 2 * SA_FLAT; k += HEAD_THREADS) dosa[k] = 0.f;
    attn_bwd_qkv<CA_HID, CA_OUT>(ws, G, d.reverse, sc.dq_ca[side] + row0 * CA_HID, CA_HID, sc.dkv_ca[side] + row0 * CA_DKV, CA_DKV,
                                 sc.dkv_ca[side] + row0 * CA_DKV + CA_HID, CA_DKV, gwave, lane);
    __syncthreads(); STOP(4);
    // queries came from the own side's self-attention output, keys/values from the other side's
    attn_bwd_dx<CA_HID, CA_OUT>(G, hb.ca, SA_OUT, dosa + side * SA_FLAT, 1, gwave, lane);
    __syncthreads(); STOP(5);             // the other side's key/value gradient lands in the same array
    attn_bwd_dx<CA_HID, CA_OUT>(G, hb.ca, SA_OUT, dosa + (1 - side) * SA_FLAT, 2, gwave, lane);
    __syncthreads(); STOP(6);
    // ---- self stage: recompute its forward into the (now free) workspace, then differentiate
    proj_self(xm, pm, hb.sa, ws, gwave, lane);
    __syncthreads(); STOP(7);
    if (gwave == 0) {
      attn_core<SA_HID, SA_OUT, true>(ws, hb.sa, false, osa_own, lane);
      attn_bwd_core<SA_HID, SA_OUT>(ws, G, hb.sa, grad_block(gw, 0, side), false, dosa + side * SA_FLAT, lane);
    }
    __syncthreads(); STOP(8);
    float* gy = sc.dy_sa[side] + row0 * SA_DY;
    attn_bwd_qkv<SA_HID, SA_OUT>(ws, G, false, gy, SA_DY, gy + SA_HID, SA_DY, gy + 2 * SA_HID, SA_DY, gwave, lane);
    __syncthreads(); STOP(9);
    attn_bwd_dx<SA_HID, SA_OUT>(G, hb.sa, pm, dxm, 3, gwave, lane);
    __syncthreads(); STOP(10);
  }
  // y = x/||x||  ->  dx = (dy - y (y.dy)) / ||x||
  {
    const int n = side ? d.d_img : d.d_txt;
    const float* y = side ? xs : xs + d.d_img;
    const float* dy = side ? dxs : dxs + d.d_img;
    T* dst = side ? dimg : dtxt;
    float dot = 0.f;
    for (int k = gt; k < n; k += 256) dot += y[k] * dy[k];
    dot = wave_sum(dot);
    if (lane == 0) misc[side * 4 + gwave] = dot;
    __syncthreads(); STOP(11);
    const float tot = misc[side * 4] + misc[side * 4 + 1] + misc[side * 4 + 2] + misc[side * 4 + 3];
    const float inv = 1.f / misc[16 + side];
    if (dst) for (int k = gt; k < n; k += 256) dst[b * n + k] = from_f<T>((dy[k] - y[k] * tot) * inv);
  }
}

// ---- weight gradients: dW[c][k] = sum_m dY[m][c] X[m][k] over the B*16 rows of the workspace ---------------------------
struct WgGemm {
  const float* dY; const float* X;
  int ldy, ncols, ldx, nin;
  int nkb, blk0;                 // 64-wide k blocks; first block id
  float* W[3]; float* bias[3];   // destination segments of the dY columns (Q | K | V)
  int seg_end[3];
};
struct WgPlan {
  WgGemm g[6];
  int n_gemm, n_blocks, ksplit, rows;
  const float* dl; const float* v; float* fin_w;
  int n_classes, wact, wfull, B, fin_colblocks;
};
#define FIN_CHUNK 16               // samples per classifier-gradient workgroup (all their loads in flight at once)

__global__ void __launch_bounds__(256)
head_wgrad_k(WgPlan p) {
  extern __shared__ __attribute__((aligned(16))) float red[];     // [4 waves][64 regs][64 lanes]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ngb = p.n_blocks * p.ksplit;
  if ((int)blockIdx.x >= ngb) {
    // classifier weight: dfin[k][c] = sum_b dl[b][k] v[b][c]
    const int fb = blockIdx.x - ngb;
    const int c = (fb % p.fin_colblocks) * 256 + t;
    const int b0 = (fb / p.fin_colblocks) * FIN_CHUNK;
    const int b1 = min(b0 + FIN_CHUNK, p.B);
    if (c >= p.wact) return;
    float acc[MAX_CLASSES];
#pragma unroll
    for (int k = 0; k < MAX_CLASSES; ++k) acc[k] = 0.f;
    float vv[FIN_CHUNK];
#pragma unroll
    for (int i = 0; i < FIN_CHUNK; ++i) vv[i] = b0 + i < b1 ? p.v[(int64_t)(b0 + i) * p.wfull + c] : 0.f;
#pragma unroll
    for (int i = 0; i < FIN_CHUNK; ++i) {
      const int b = min(b0 + i, b1 - 1);
#pragma unroll
      for (int k = 0; k < MAX_CLASSES; ++k) if (k < p.n_classes) acc[k] += p.dl[b * p.n_classes + k] * vv[i];
    }
#pragma unroll
    for (int k = 0; k < MAX_CLASSES; ++k) if (k < p.n_classes) atomicAdd(p.fin_w + (int64_t)k * p.wact + c, acc[k]);
    return;
  }
  const int kz = blockIdx.x / p.n_blocks, blk = blockIdx.x % p.n_blocks;
  int gi = 0;
#pragma unroll
  for (int i = 1; i < 6; ++i) if (i < p.n_gemm && blk >= p.g[i].blk0) gi = i;
  const WgGemm& G = p.g[gi];
  const int local = blk - G.blk0, cb = local / G.nkb, kblk = local % G.nkb;
  const int x = lane & 15, g = lane >> 4;
  const int cbase = 64 * cb + 4 * x, kbase = 64 * kblk + 4 * x;
  const bool c_ok = cbase < G.ncols, k_ok = kbase < G.nin;
  const int row0 = kz * WG_ROWS + wave * 64;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // the rows were written by the previous kernel, i.e. they come from the Infinity Cache / HBM (~2 us): put all 32
  // row loads of the wave in flight before the first MFMA
  f32x4 av[16], bv[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int m = row0 + 4 * s + g;
    const bool m_ok = m < p.rows;
    av[s] = (m_ok && c_ok) ? ld4(G.dY + (int64_t)m * G.ldy + cbase) : zero;
    bv[s] = (m_ok && k_ok) ? ld4(G.X + (int64_t)m * G.ldx + kbase) : zero;
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    bsum += av[s];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma4(av[s][i], bv[s][j], acc[i][j]);
  }
  // acc[i][j][r] = dW[c = 64 cb + 4 (4 g + r) + i][k = 64 kblk + 4 x + j], partial over this wave's 64 rows
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 64 + (i * 4 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
  __shared__ float bred[4][64];      // per-wave column sums of dY (the bias gradient), reduced before the atomics
  if (kblk == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      if (g == 0) bred[wave][4 * x + i] = v;
    }
  }
  __syncthreads();
  if (kblk == 0 && t < 64) {
    const int c = 64 * cb + t;
    if (c < G.ncols) {
      const int sgi = c < G.seg_end[0] ? 0 : (c < G.seg_end[1] ? 1 : 2);
      const int s0 = sgi == 0 ? 0 : G.seg_end[sgi - 1];
      atomicAdd(G.bias[sgi] + (c - s0), (bred[0][t] + bred[1][t]) + (bred[2][t] + bred[3][t]));
    }
  }
  // reduce the four waves in LDS: one atomic per weight per 256 rows
  float sums[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int e = t + 256 * it;
    const int kk = e & 63, ci = e >> 6;
    const int i = ci & 3, q = ci >> 2, gg = q >> 2, r = q & 3, xx = kk >> 2, j = kk & 3;
    const int idx = ((i * 4 + j) * 4 + r) * 64 + gg * 16 + xx;
    sums[it] = red[idx] + red[4096 + idx] + red[8192 + idx] + red[12288 + idx];
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int e = t + 256 * it;
    const int c = 64 * cb + (e >> 6), k = 64 * kblk + (e & 63);
    if (c < G.ncols && k < G.nin) {
      const int sgi = c < G.seg_end[0] ? 0 : (c < G.seg_end[1] ? 1 : 2);
      const int s0 = sgi == 0 ? 0 : G.seg_end[sgi - 1];
      float* dst = G.W[sgi] + (int64_t)(c - s0) * G.nin + k;
      atomicAdd(dst, sums[it]);
    }
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

extern "C" int mmrca_head_fwd(const void* img, const void* txt, const MmrcaHeadWeights* w, float* logits,
                              int B, int d_img, int d_txt, int n_classes, int reverse, int mode, float drop_p,
                              uint64_t seed, int dtype, void* stream) {
  MMRCA_REQUIRE(img && txt && w && logits, "head_fwd: null pointer");
  if (int rc = head_check(B, d_img, d_txt, n_classes, mode)) return rc;
  MMRCA_REQUIRE(head_weights_ok(w), "head_fwd: null weight pointer");
  MMRCA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "head_fwd: dropout p must be in [0,1)");
  HeadDims d = {d_img, d_txt, d_img / HP, d_txt / HP, CA_FLAT + d_img + d_txt, n_classes, mode, reverse ? 1 : 0, 0};
  const HeadLayout L = head_layout(d, false);
  const size_t lds = (size_t)L.total * 4;
  MMRCA_REQUIRE(lds <= 160 * 1024, "head_fwd: LDS budget exceeded");
  MMRCA_DISPATCH_DTYPE(dtype, "head_fwd",
    MMRCA_MAX_LDS((int)lds, head_fwd_k<T>);
    hipLaunchKernelGGL(head_fwd_k<T>, dim3(B), dim3(HEAD_THREADS), lds, (hipStream_t)stream, (const T*)img, (const T*)txt, *w, logits, d, L,
                       drop_p, seed);)
  MMRCA_CHECK_LAUNCH("head_fwd");
  return 0;
}

extern "C" int64_t mmrca_head_bwd_workspace_bytes(int B, int d_img, int d_txt) {
  if (B <= 0 || d_img <= 0 || d_txt <= 0) return 0;
  return head_scratch_floats(B, d_img, d_txt) * 4;
}

extern "C" int mmrca_head_bwd(const float* dlogits, const void* img, const void* txt, const MmrcaHeadWeights* w,
                              const MmrcaHeadGrads* g, void* dimg, void* dtxt, int B, int d_img, int d_txt, int n_classes,
                              int reverse, int mode, float drop_p, uint64_t seed, int dtype, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  MMRCA_REQUIRE(dlogits && w && g && img && txt, "head_bwd: null pointer");
  if (int rc = head_check(B, d_img, d_txt, n_classes, mode)) return rc;
  MMRCA_REQUIRE(head_weights_ok(w), "head_bwd: null weight pointer");
  {
    float* const* p = (float* const*)g;
    for (size_t i = 0; i < sizeof(MmrcaHeadGrads) / sizeof(float*); ++i) MMRCA_REQUIRE(p[i], "head_bwd: null gradient pointer");
  }
  MMRCA_REQUIRE(workspace && workspace_bytes >= mmrca_head_bwd_workspace_bytes(B, d_img, d_txt) && ((uintptr_t)workspace & 15) == 0,
                "head_bwd: workspace of mmrca_head_bwd_workspace_bytes() bytes (16-byte aligned) required");
  static const int head_stop = getenv("MMRCA_HEAD_STOP") ? atoi(getenv("MMRCA_HEAD_STOP")) : 0;
  HeadDims d = {d_img, d_txt, d_img / HP, d_txt / HP, CA_FLAT + d_img + d_txt, n_classes, mode, reverse ? 1 : 0, head_stop % 100};
  const HeadLayout L = head_layout(d, true);
  const size_t lds = (size_t)L.total * 4;
  MMRCA_REQUIRE(lds <= 160 * 1024, "head_bwd: LDS budget exceeded");
  const HeadScratch sc = carve_scratch((float*)workspace, B, d);
  MMRCA_DISPATCH_DTYPE(dtype, "head_bwd",
    MMRCA_MAX_LDS((int)lds, head_bwd_k<T>);
    hipLaunchKernelGGL(head_bwd_k<T>, dim3(B), dim3(HEAD_THREADS), lds, (hipStream_t)stream, dlogits, (const T*)img, (const T*)txt, *w, *g,
                       (T*)dimg, (T*)dtxt, d, L, sc, drop_p, seed);)
  MMRCA_CHECK_LAUNCH("head_bwd");

  WgPlan p = {};
  p.rows = B * HP; p.B = B; p.ksplit = (p.rows + WG_ROWS - 1) / WG_ROWS;
  p.dl = dlogits; p.v = sc.v; p.fin_w = g->fin_w;  p.n_classes = n_classes; p.wfull = d.wfull;
  p.wact = mode == 1 ? d_img + d_txt : (mode == 2 ? CA_FLAT : d.wfull);
  p.fin_colblocks = (p.wact + 255) / 256;
  if (mode != 1) {
    auto add = [&](const float* dY, int ldy, int ncols, const float* X, int nin, float* W0, float* b0, int e0, float* W1, float* b1,
                   int e1, float* W2, float* b2) {
      WgGemm& q = p.g[p.n_gemm++];
      q.dY = dY; q.X = X; q.ldy = ldy; q.ncols = ncols; q.ldx = nin; q.nin = nin;
      q.nkb = (nin + 63) / 64; q.blk0 = p.n_blocks;
      q.W[0] = W0; q.W[1] = W1; q.W[2] = W2; q.bias[0] = b0; q.bias[1] = b1; q.bias[2] = b2;
      q.seg_end[0] = e0; q.seg_end[1] = e1; q.seg_end[2] = ncols;
      p.n_blocks += ((ncols + 63) / 64) * q.nkb;
    };
    add(sc.dy_sa[0], SA_DY, SA_DY, sc.xt, d.pt, g->sat_wq, g->sat_bq, SA_HID, g->sat_wk, g->sat_bk, 2 * SA_HID, g->sat_wv, g->sat_bv);
    add(sc.dy_sa[1], SA_DY, SA_DY, sc.xi, d.pi, g->sai_wq, g->sai_bq, SA_HID, g->sai_wk, g->sai_bk, 2 * SA_HID, g->sai_wv, g->sai_bv);
    add(sc.dq_ca[0], CA_HID, CA_HID, sc.osat, SA_OUT, g->c1_wq, g->c1_bq, CA_HID, g->c1_wq, g->c1_bq, CA_HID, g->c1_wq, g->c1_bq);
    add(sc.dkv_ca[0], CA_DKV, CA_DKV, sc.osai, SA_OUT, g->c1_wk, g->c1_bk, CA_HID, g->c1_wv, g->c1_bv, CA_DKV, g->c1_wv, g->c1_bv);
    add(sc.dq_ca[1], CA_HID, CA_HID, sc.osai, SA_OUT, g->c2_wq, g->c2_bq, CA_HID, g->c2_wq, g->c2_bq, CA_HID, g->c2_wq, g->c2_bq);
    add(sc.dkv_ca[1], CA_DKV, CA_DKV, sc.osat, SA_OUT, g->c2_wk, g->c2_bk, CA_HID, g->c2_wv, g->c2_bv, CA_DKV, g->c2_wv, g->c2_bv);
  }
  MMRCA_REQUIRE(p.n_blocks == (mode != 1 ? wg_blocks(d_img, d_txt) : 0) && p.n_blocks <= 64 && p.ksplit == wg_ksplit(B), "head_bwd: block plan mismatch");
  const int fin_blocks = p.fin_colblocks * ((B + FIN_CHUNK - 1) / FIN_CHUNK);
  const int grid = p.n_blocks * p.ksplit + fin_blocks;
  if (head_stop >= 100) return 0;
  MMRCA_MAX_LDS(65536, head_wgrad_k);
  hipLaunchKernelGGL(head_wgrad_k, dim3(grid), dim3(256), 65536, (hipStream_t)stream, p);
  MMRCA_CHECK_LAUNCH("head_wgrad");
  return 0;
}

MMRCA_SEED_EPOCH_EXPORT(head)   // this translation unit's copy of the mask epoch (common.h)
