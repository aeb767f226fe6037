// Dense 3x3, stride-1, pad-1 convolutions of NHWC bf16 activations as IMPLICIT GEMMs on the bf16 matrix cores (gfx950).
//
// Replaces im2row3x3_tap + mmrca_gemm (forward) and im2row3x3_tap + weight-gradient GEMM for the FusedMBConv stages of
// EfficientNetV2 (reference: torchvision Conv2dNormActivation 3x3 inside FusedMBConv, built by multimodal_model.py:113-126
// `models.efficientnet_v2_l`): the materialised patch matrix is nine times the activation (0.8 .. 4.2 GB per layer at
// B = 128, 480 px) and its write + reads were the largest single HBM stream of the conv path.
//
//   forward     z[p, co] = sum_{tap, ci} x[p + off(tap), ci] * w[co, tap, ci]          (tap-major weight copy, as before)
//     One block = one PH x PW patch of output pixels (PH * PW = 128, PW a multiple of 16) x NT output channels.  The patch's
//     input HALO ((PH+2) x (PW+2) pixels x CK channels) is copied to LDS once by LDS-DMA, out-of-image positions from a zero
//     page; the nine taps are then nine shifted views of that one image: the A fragment of tap (dy, dx) for 16 consecutive
//     pixels of a patch row is one ds_read_b128 per lane at a tap-dependent scalar offset.  Pixel pitch in LDS = CK/8 + 1
//     16-byte slots (odd), so the 16 lanes of a read group fall on 16 distinct slots of a bank row.  Only the weights stream
//     through a double-buffered 32-deep stage per step.  With more than 96 input channels (the input-gradient form: the
//     "input" is dz with Cout channels) the halo is loaded CK channels at a time.  Channel counts that are multiples of 8 but
//     not of 32 (EfficientNetV2-M: 24 / 48 / 80) take the same path: the weight copy carries Cp = round_up(Cin, 32) channels
//     per tap (zeros), the halo's missing chunks come from the zero page.
//   statistics  BatchNorm moments of the bf16-rounded outputs ride in the epilogue: every wave reduces its 64 pixels to a
//     per-channel (count, mean, M2) triple (two in-register passes, so no shift is needed) and writes it to a per-wave slot;
//     conv_bn_finish_k merges the slots with Chan's formula.  No atomics, bit-reproducible.
//   weight gradient  dw[co, tap, ci] += sum_p dz[p, co] * x[p + off(tap), ci]: 128 x 128 x 32 tiles like gemm_mfma_k32, the
//     B operand (patches) gathered by the staging loads straight from x, fp32 atomics over pixel ranges.
#include "common.h"
#include "lds_asm.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void_c;
typedef const __attribute__((address_space(1))) void gbl_void_c;

__device__ __attribute__((aligned(256))) bf16_t g_conv_zero_page[128];     // zero-initialised: source of every out-of-image load

// sum over the 16 lanes of a DPP row (= one 16-lane group of the MFMA layouts), result in every lane: four v_add_f32_dpp
// (the generic __shfl_xor goes through ds_bpermute, i.e. the LDS pipe -- 128 of them per tile made the statistics cost as much as
// a separate pass over z)
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

struct ConvGeom {
  int B, H, W, Cin, Cout;
  int Cp;                     // Cin rounded up to 32: channels per tap in the weight copy and in the K loop (pad channels are zero)
  int PH, PW;                 // output patch of a block (PH * PW = 128)
  int tiles_y, tiles_x;       // patches per image
  int CK;                     // input channels resident per halo load (Cp % CK == 0, CK % 32 == 0, CK <= 96)
  int tiles_n;                // Cout tiles of NT channels
};

// ---------------------------------------------------------------------------------------------------------------------
// forward / input gradient
// ---------------------------------------------------------------------------------------------------------------------
template <int NJ, bool STATS>
__global__ void __launch_bounds__(256, 3)
conv3x3_igemm_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ z, float* __restrict__ part_mean,
                float* __restrict__ part_m2, float* __restrict__ part_cnt, const ConvGeom gm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = 32 * NJ;                      // output channels per block
  constexpr int BST = NT * 64;                     // bytes of one weight stage (NT rows x 32 k)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int g = lane >> 4, l16 = lane & 15;
  // blocks that share a halo (the tiles_n channel tiles of one patch) and neighbouring patches stay on one XCD (one L2)
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int tn = wgid % gm.tiles_n, tmi = wgid / gm.tiles_n;
  const int tx = tmi % gm.tiles_x, ty = (tmi / gm.tiles_x) % gm.tiles_y, b = tmi / (gm.tiles_x * gm.tiles_y);
  const int y0 = ty * gm.PH, x0 = tx * gm.PW;
  const int n_blk = tn * NT;
  const int HWp = gm.PW + 2, npx = (gm.PH + 2) * HWp;
  const int chunks = gm.CK >> 3, slots = chunks + 1;
  const int halo_slots = npx * slots;
  const int halo_bytes = ((halo_slots + 63) & ~63) * 16;
  char* const bst = smem + halo_bytes;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_c*)smem;

  // A fragment addresses: fragment ig = 4 wr + i is 16 consecutive pixels of patch row ig / segs
  const int segs = gm.PW >> 4;
  unsigned abase[4];
  bool pvalid[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ig = 4 * wr + i, prow = ig / segs, pcol = (ig - prow * segs) * 16;
    abase[i] = lds0 + (unsigned)(((prow * HWp + pcol + l16) * slots + g) * 16);
    pvalid[i] = (y0 + prow < gm.H) && (x0 + pcol + l16 < gm.W);
  }
  unsigned bbase[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int r = (wc * NJ + j) * 16 + l16;
    bbase[j] = lds0 + (unsigned)halo_bytes + (unsigned)(r * 64 + ((g ^ ((4 - (r >> 2)) & 3)) << 4));
  }

  auto load_halo = [&](int c0) {                   // channels c0 .. c0 + CK - 1 of the (PH+2) x (PW+2) input window
    for (int base = wave * 64; base < halo_slots; base += 256) {
      const int idx = base + lane;
      const int px = idx / slots, ch = idx - px * slots;
      const int hy = px / HWp, hx = px - hy * HWp;
      const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
      const bool ok = ch < chunks && c0 + ch * 8 < gm.Cin && px < npx && (unsigned)yy < (unsigned)gm.H && (unsigned)xx < (unsigned)gm.W;
      const bf16_t* src = ok ? x + (((int64_t)b * gm.H + yy) * gm.W + xx) * gm.Cin + c0 + ch * 8 : g_conv_zero_page;
      __builtin_amdgcn_global_load_lds((gbl_void_c*)src, (lds_void_c*)(smem + base * 16), 16, 0, 0);
    }
  };
  auto stage_w = [&](int64_t k0, char* dst) {      // NT rows x 32 k of the [Cout, 9 Cin] weight copy, ROWK image of gemm_mfma_k32
    for (int i = wave; i < 2 * NJ; i += 4) {
      const int r = 16 * i + (lane >> 2);
      const int c = (lane & 3) ^ ((4 - (r >> 2)) & 3);
      int gr = n_blk + r;
      if (gr > gm.Cout - 1) gr = gm.Cout - 1;       // ragged channel tile: a valid row, never stored
      const bf16_t* src = w + (int64_t)gr * (9 * gm.Cp) + k0 + c * 8;
      __builtin_amdgcn_global_load_lds((gbl_void_c*)src, (lds_void_c*)(dst + i * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int kc_n = gm.CK >> 5;                     // 32-deep steps per tap within one halo chunk
  const int steps = 9 * kc_n;
  for (int c0 = 0; c0 < gm.Cp; c0 += gm.CK) {
    // (every wave passed the barrier that closes the previous chunk's last step: halo and both weight stages are free)
    load_halo(c0);
    stage_w(c0, bst);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int tap = 0, kc = 0;
    for (int s = 0; s < steps; ++s) {
      const unsigned cur = (unsigned)((s & 1) * BST);
      int ntap = tap, nkc = kc + 1;
      if (nkc == kc_n) { nkc = 0; ++ntap; }
      if (s + 1 < steps) stage_w((int64_t)ntap * gm.Cp + c0 + nkc * 32, bst + ((s + 1) & 1) * BST);
      const int dy = tap / 3, dx = tap - 3 * dy;
      const unsigned aoff = (unsigned)(((dy * HWp + dx) * slots + kc * 4) * 16);
      Frag<false> af[4], bf[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) DS_READ_B128(af[i].v, abase[i] + aoff, 0);
#pragma unroll
      for (int j = 0; j < NJ; ++j) DS_READ_B128(bf[j].v, bbase[j] + cur, 0);
      lgkm0(af);
      if constexpr (NJ == 4) lgkm0(bf);
      else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(bf[j].v));      // (the wait above covers every read issued before it)
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j].v, af[i].v, acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      tap = ntap; kc = nkc;
    }
  }

  // ---- epilogue: lane (g, l16) holds channels n0 + 4g .. +3 of pixel l16 of fragment i (the transposed MFMA form)
  bf16x4 o[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ig = 4 * wr + i, prow = ig / segs, pcol = (ig - prow * segs) * 16;
    const int64_t p = ((int64_t)b * gm.H + y0 + prow) * gm.W + x0 + pcol + l16;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[i][j][r] = (bf16_t)acc[i][j][r];
      const int n = n_blk + (wc * NJ + j) * 16 + 4 * g;
      if (pvalid[i] && n < gm.Cout) *reinterpret_cast<bf16x4*>(z + p * gm.Cout + n) = o[i][j];
    }
  }
  if constexpr (STATS) {
    float cnt = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) cnt += pvalid[i] ? 1.f : 0.f;
    cnt = row16_sum(cnt);
    const float inv = cnt > 0.f ? 1.0f / cnt : 0.f;
    const int64_t slot = (int64_t)tmi * 2 + wr;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float mu[4], m2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += pvalid[i] ? (float)o[i][j][r] : 0.f;
        mu[r] = row16_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = (float)o[i][j][r] - mu[r]; q += pvalid[i] ? d * d : 0.f; }
        m2[r] = row16_sum(q);
      }
      const int n = n_blk + (wc * NJ + j) * 16 + 4 * g;
      if (l16 == 0 && n < gm.Cout) {
        *reinterpret_cast<f32x4*>(part_mean + slot * gm.Cout + n) = (f32x4){mu[0], mu[1], mu[2], mu[3]};
        *reinterpret_cast<f32x4*>(part_m2 + slot * gm.Cout + n) = (f32x4){m2[0], m2[1], m2[2], m2[3]};
      }
    }
    if (tn == 0 && wc == 0 && lane == 0) part_cnt[slot] = cnt;
  }
}

// Chan merge of (count, mean, M2) slots t0 .. t1-1 for 32 channels by one block of 32 channels x 8 slot lanes: returns the
// merged triple to the threads of slot lane 0.
__device__ __forceinline__ void merge_slots(const float* __restrict__ part_mean, const float* __restrict__ part_m2,
                                            const float* __restrict__ part_cnt, int64_t t0, int64_t t1, int C, int c, bool ok,
                                            double& N, double& mu, double& Q) {
  __shared__ double red[8][33];
  __shared__ double red_n[8];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  double n = 0.0, s = 0.0;
  for (int64_t t = t0 + sl; t < t1; t += 8) {
    const double k = (double)part_cnt[t];
    n += k;
    if (ok) s += k * (double)part_mean[t * C + c];
  }
  red[sl][cl] = s;
  if (cl == 0) red_n[sl] = n;
  __syncthreads();
  N = 0.0;
  double S = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { N += red_n[k]; S += red[k][cl]; }
  mu = N > 0.0 ? S / N : 0.0;
  __syncthreads();
  double q = 0.0;
  if (ok)
    for (int64_t t = t0 + sl; t < t1; t += 8) {
      const double k = (double)part_cnt[t], d = (double)part_mean[t * C + c] - mu;
      q += (double)part_m2[t * C + c] + k * d * d;
    }
  red[sl][cl] = q;
  __syncthreads();
  Q = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) Q += red[k][cl];
}

// level 1: every MERGE_SLOTS first-level slots -> one second-level slot (rows n1 .. of the same arrays)
#define MERGE_SLOTS 256
__global__ void __launch_bounds__(256)
conv_bn_merge_k(float* __restrict__ part_mean, float* __restrict__ part_m2, float* __restrict__ part_cnt, int64_t n1, int C) {
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  const bool ok = c < C;
  const int64_t t0 = (int64_t)blockIdx.y * MERGE_SLOTS;
  const int64_t t1 = t0 + MERGE_SLOTS < n1 ? t0 + MERGE_SLOTS : n1;
  double N, mu, Q;
  merge_slots(part_mean, part_m2, part_cnt, t0, t1, C, c, ok, N, mu, Q);
  if ((threadIdx.x >> 5) == 0) {
    const int64_t o = n1 + blockIdx.y;
    if (ok) { part_mean[o * C + c] = (float)mu; part_m2[o * C + c] = (float)Q; }
    if (blockIdx.x == 0 && threadIdx.x == 0) part_cnt[o] = (float)N;
  }
}

// final level -> mean, rstd (+ running statistics, torch semantics)
__global__ void __launch_bounds__(256)
conv_bn_finish_k(const float* __restrict__ part_mean, const float* __restrict__ part_m2, const float* __restrict__ part_cnt,
                 int64_t t0, int64_t t1, float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                 float* __restrict__ running_var, int C, float eps, float momentum) {
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  const bool ok = c < C;
  double N, mu, Q;
  merge_slots(part_mean, part_m2, part_cnt, t0, t1, C, c, ok, N, mu, Q);
  if ((threadIdx.x >> 5) == 0 && ok) {
    const float var = N > 0.0 ? (float)(Q / N) : 0.f;
    mean[c] = (float)mu;
    rstd[c] = rsqrtf(var + eps);
    if (running_mean && momentum > 0.f) {
      const float nf = (float)N;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (nf > 1.f ? var * nf / (nf - 1.f) : var);
    }
  }
}

// mean / rstd (+ running statistics) from the shifted sums mmrca_gemm_bnstats left: s1 / s2 [nslots, C], shift [C] or NULL.
// block = 32 channels x 8 slot lanes, sums in double.  (shift may alias running_mean: it is read before the update.)
__global__ void __launch_bounds__(256)
bn_finish_sums_k(const float* __restrict__ s1, const float* __restrict__ s2, const float* __restrict__ shift, int64_t nslots, double rows,
                 float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean, float* __restrict__ running_var, int C,
                 float eps, float momentum) {
  __shared__ double red[2][8][33];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int64_t t = sl; t < nslots; t += 8) { a += (double)s1[t * C + c]; b += (double)s2[t * C + c]; }
  red[0][sl][cl] = a; red[1][sl][cl] = b;
  __syncthreads();
  if (sl == 0 && c < C) {
    double S1 = 0.0, S2 = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { S1 += red[0][k][cl]; S2 += red[1][k][cl]; }
    const double d1 = S1 / rows;
    const float mu = (float)((shift ? (double)shift[c] : 0.0) + d1);
    double v = S2 / rows - d1 * d1;
    if (v < 0.0) v = 0.0;
    const float var = (float)v;
    mean[c] = mu;
    rstd[c] = rsqrtf(var + eps);
    if (running_mean && momentum > 0.f) {
      const float nf = (float)rows;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (nf > 1.f ? var * nf / (nf - 1.f) : var);
    }
  }
}
/* mean[C], rstd[C] (+ running statistics when momentum > 0) from mmrca_gemm_bnstats' s1 / s2 [nslots, C] over `rows` rows in all */
extern "C" int mmrca_bn_finish_sums(const float* s1, const float* s2, const float* shift, int64_t nslots, int64_t rows, float* mean, float* rstd,
                                    float* running_mean, float* running_var, int C, float eps, float momentum, void* stream) {
  MMRCA_REQUIRE(s1 && s2 && mean && rstd && nslots > 0 && rows > 0 && C > 0, "bn_finish_sums: bad arguments");
  hipLaunchKernelGGL(bn_finish_sums_k, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)stream, s1, s2, shift, nslots, (double)rows, mean, rstd,
                     running_mean, running_var, C, eps, momentum);
  MMRCA_CHECK_LAUNCH("bn_finish_sums");
  return 0;
}

// patch shape: PW in {16, 32, 64} (PH = 128 / PW), whichever covers the image with the fewest padded pixels (ties: the squarer
// patch, whose halo is smaller)
static void conv_patch(int H, int W, int* PH, int* PW) {
  long best = -1;
  for (int pw = 16; pw <= 64; pw <<= 1) {
    const int ph = 128 / pw;
    const long area = (long)((H + ph - 1) / ph) * ph * (long)((W + pw - 1) / pw) * pw;
    if (best < 0 || area < best) { best = area; *PH = ph; *PW = pw; }
  }
}

static int conv_geom(ConvGeom* gm, int B, int H, int W, int Cin, int Cout, int NT) {
  gm->B = B; gm->H = H; gm->W = W; gm->Cin = Cin; gm->Cout = Cout;
  conv_patch(H, W, &gm->PH, &gm->PW);
  gm->tiles_y = (H + gm->PH - 1) / gm->PH;
  gm->tiles_x = (W + gm->PW - 1) / gm->PW;
  gm->Cp = (Cin + 31) / 32 * 32;
  gm->CK = gm->Cp <= 96 ? gm->Cp : (gm->Cp % 96 == 0 ? 96 : (gm->Cp % 64 == 0 ? 64 : 32));
  gm->tiles_n = (Cout + NT - 1) / NT;
  return 0;
}

static int conv_nt(int Cout) {            // output channels per block: 128, or the whole (smaller) layer
  if (Cout >= 128) return 128;
  return ((Cout + 31) / 32) * 32;
}

static int64_t conv_slots1(int B, int H, int W) {
  int PH, PW;
  conv_patch(H, W, &PH, &PW);
  return (int64_t)B * ((H + PH - 1) / PH) * ((W + PW - 1) / PW) * 2;
}
/* rows of part_mean / part_m2 and length of part_cnt for a [B, H, W] output: one slot per wave row of every patch, plus the
 * second-level slots mmrca_conv_bn_finish merges them into */
extern "C" int64_t mmrca_conv3x3_stat_slots(int B, int H, int W) {
  const int64_t n1 = conv_slots1(B, H, W);
  return n1 + (n1 + MERGE_SLOTS - 1) / MERGE_SLOTS;
}

/* z[B*H*W, Cout] = conv3x3(x[B*H*W, Cin], w_tap[Cout, 9*Cin]) (stride 1, zero padding 1; bf16, NHWC rows, tap-major weights:
 * column tap*Cp + ci, tap = 3*ky + kx, Cp = Cin rounded up to a multiple of 32 with zero pad columns -- Cp == Cin for the
 * FusedMBConv layers of EfficientNetV2-L (32 / 64 / 96 channels), 24 -> 32, 48 -> 64, 80 -> 96 for V2-M).  Cin % 8 == 0,
 * Cout % 8 == 0, 16-byte aligned operands.  With part_* given, the
 * BatchNorm moments of the stored outputs are left in mmrca_conv3x3_stat_slots() slots for mmrca_conv_bn_finish.
 * The input-gradient of the same convolution is this call on dz with the flipped, transposed weights [Cin, 9*Cout]. */
extern "C" int mmrca_conv3x3_fwd(const void* x, const void* w_tap, void* z, float* part_mean, float* part_m2, float* part_cnt,
                                 int B, int H, int W, int Cin, int Cout, int dtype, void* stream) {
  MMRCA_REQUIRE(dtype == MMRCA_BF16, "conv3x3_fwd: bf16 only (the fp32 modes keep im2row + GEMM)");
  MMRCA_REQUIRE(x && w_tap && z && B > 0 && H > 0 && W > 0, "conv3x3_fwd: bad arguments");
  MMRCA_REQUIRE(Cin % 8 == 0 && Cin >= 8, "conv3x3_fwd: Cin must be a multiple of 8 (got %d)", Cin);
  MMRCA_REQUIRE(Cout % 8 == 0 && Cout >= 8, "conv3x3_fwd: Cout must be a multiple of 8 (got %d)", Cout);
  MMRCA_REQUIRE((((uintptr_t)x | (uintptr_t)w_tap | (uintptr_t)z) & 15) == 0, "conv3x3_fwd: operands must be 16-byte aligned");
  const bool stats = part_mean != nullptr;
  MMRCA_REQUIRE(!stats || (part_m2 && part_cnt), "conv3x3_fwd: statistics need part_mean, part_m2 and part_cnt");
  MMRCA_REQUIRE(!stats || Cout % 4 == 0, "conv3x3_fwd: statistics need Cout %% 4 == 0");
  const int NT = conv_nt(Cout);
  ConvGeom gm;
  conv_geom(&gm, B, H, W, Cin, Cout, NT);
  const int64_t nblk = (int64_t)B * gm.tiles_y * gm.tiles_x * gm.tiles_n;
  MMRCA_REQUIRE(nblk < (1ll << 31), "conv3x3_fwd: too many tiles");
  const int halo_slots = (gm.PH + 2) * (gm.PW + 2) * (gm.CK / 8 + 1);
  const int lds = ((halo_slots + 63) & ~63) * 16 + 2 * NT * 64;
  hipStream_t st = (hipStream_t)stream;
#define CONV_LAUNCH(NJ_, ST_)                                                                                                       \
  do {                                                                                                                              \
    MMRCA_MAX_LDS(lds, conv3x3_igemm_k<NJ_, ST_>);             \
    hipLaunchKernelGGL((conv3x3_igemm_k<NJ_, ST_>), dim3((unsigned)nblk), dim3(256), lds, st, (const bf16_t*)x, (const bf16_t*)w_tap, \
                       (bf16_t*)z, part_mean, part_m2, part_cnt, gm);                                                               \
  } while (0)
  switch (NT / 32) {
    case 1: if (stats) CONV_LAUNCH(1, true); else CONV_LAUNCH(1, false); break;
    case 2: if (stats) CONV_LAUNCH(2, true); else CONV_LAUNCH(2, false); break;
    case 3: if (stats) CONV_LAUNCH(3, true); else CONV_LAUNCH(3, false); break;
    default: if (stats) CONV_LAUNCH(4, true); else CONV_LAUNCH(4, false); break;
  }
#undef CONV_LAUNCH
  MMRCA_CHECK_LAUNCH("conv3x3_fwd");
  return 0;
}

/* mean[C], rstd[C] (+ running statistics when momentum > 0) from the slots mmrca_conv3x3_fwd left for a [B, H, W] output
 * (part_* are written: the second-level slots) */
extern "C" int mmrca_conv_bn_finish(float* part_mean, float* part_m2, float* part_cnt, int B, int H, int W, float* mean, float* rstd,
                                    float* running_mean, float* running_var, int C, float eps, float momentum, void* stream) {
  MMRCA_REQUIRE(part_mean && part_m2 && part_cnt && mean && rstd && B > 0 && H > 0 && W > 0 && C > 0, "conv_bn_finish: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n1 = conv_slots1(B, H, W);
  int64_t t0 = 0, t1 = n1;
  if (n1 > 2 * MERGE_SLOTS) {
    const int64_t n2 = (n1 + MERGE_SLOTS - 1) / MERGE_SLOTS;
    MMRCA_REQUIRE(n2 < 65536, "conv_bn_finish: too many slots");
    hipLaunchKernelGGL(conv_bn_merge_k, dim3((C + 31) / 32, (unsigned)n2), dim3(256), 0, st, part_mean, part_m2, part_cnt, n1, C);
    t0 = n1; t1 = n1 + n2;
  }
  hipLaunchKernelGGL(conv_bn_finish_k, dim3((C + 31) / 32), dim3(256), 0, st, (const float*)part_mean, (const float*)part_m2,
                     (const float*)part_cnt, t0, t1, mean, rstd, running_mean, running_var, C, eps, momentum);
  MMRCA_CHECK_LAUNCH("conv_bn_finish");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int krow_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <int MI>
__global__ void __launch_bounds__(256, 4)
conv3x3_wgrad_k(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ x, float* __restrict__ dw, const ConvGeom gm, int tiles_m,
                int tiles_n, int64_t ksplit_len) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][A tile 8 KiB | B tile 8 KiB], KROW images of gemm_mfma_k32
  constexpr int TB = 128 * 32 * 2;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
  const int m_blk = tm * 32 * MI, n_blk = tn * 128;
  const int64_t P = (int64_t)gm.B * gm.H * gm.W;
  const int N = 9 * gm.Cin;
  const int64_t kbeg = (int64_t)blockIdx.y * ksplit_len;
  int64_t kend = kbeg + ksplit_len; if (kend > P) kend = P;
  const int nt = (int)((kend - kbeg + 31) / 32);
  if (nt <= 0) return;

  // this lane's two staging pieces (k rows kr0 and kr0 + 4 of every 32-pixel step) and its fixed column chunk
  const int chp = lane & 15;
  int krow[2], ccol[2];
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    krow[ii] = 4 * (wave * 2 + ii) + (lane >> 4);
    ccol[ii] = ((((chp >> 1) ^ krow_swz(krow[ii]))) << 1) | (chp & 1);
  }
  // A (dz): columns m_blk + 8 c; B (patches of x): column n = n_blk + 8 c -> (tap, ci), fixed for the whole K loop
  bool a_ok[2], b_ok[2];
  int a_col[2], b_off[2], b_dy[2], b_dx[2];
  int py[2], pxx[2];
  int64_t pp[2];
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    a_col[ii] = m_blk + 8 * ccol[ii];
    a_ok[ii] = 8 * ccol[ii] < 32 * MI && a_col[ii] < gm.Cout;
    const int n = n_blk + 8 * ccol[ii];
    b_ok[ii] = n < N;
    const int tap = b_ok[ii] ? n / gm.Cin : 0;
    const int ci = n - tap * gm.Cin;
    b_dy[ii] = tap / 3 - 1; b_dx[ii] = tap - 3 * (tap / 3) - 1;
    b_off[ii] = (b_dy[ii] * gm.W + b_dx[ii]) * gm.Cin + ci;
    pp[ii] = kbeg + krow[ii];
    const int64_t t = pp[ii] / gm.W;
    pxx[ii] = (int)(pp[ii] - t * gm.W);
    py[ii] = (int)(t % gm.H);
  }
  auto stage = [&](char* dst) {                      // stages the step at pp[], then advances the pixel cursors by 32
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int i = wave * 2 + ii;
      const bool in = pp[ii] < kend;
      const bf16_t* sa = (in && a_ok[ii]) ? dz + pp[ii] * gm.Cout + a_col[ii] : g_conv_zero_page;
      const bool bv = in && b_ok[ii] && (unsigned)(py[ii] + b_dy[ii]) < (unsigned)gm.H && (unsigned)(pxx[ii] + b_dx[ii]) < (unsigned)gm.W;
      const bf16_t* sb = bv ? x + pp[ii] * gm.Cin + b_off[ii] : g_conv_zero_page;
      __builtin_amdgcn_global_load_lds((gbl_void_c*)sa, (lds_void_c*)(dst + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void_c*)sb, (lds_void_c*)(dst + TB + i * 1024), 16, 0, 0);
      pp[ii] += 32;
      pxx[ii] += 32;
      while (pxx[ii] >= gm.W) { pxx[ii] -= gm.W; if (++py[ii] == gm.H) py[ii] = 0; }
    }
  };

  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // fragment read addresses (KROW image: two transposed 8-byte reads, k rows +0..3 and +4..7)
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_c*)smem;
  const int g = lane >> 4, l16 = lane & 15, q = l16 >> 2, p4 = l16 & 3, frow = 8 * g + q;
  unsigned aa[MI][2], ba[4][2];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int rb = (wr * MI + i) * 16;
    aa[i][0] = lds0 + frow * 256 + ((((rb >> 4) ^ krow_swz(frow))) << 5) + p4 * 8;
    aa[i][1] = lds0 + (frow + 4) * 256 + ((((rb >> 4) ^ krow_swz(frow + 4))) << 5) + p4 * 8;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rb = wc * 64 + j * 16;
    ba[j][0] = lds0 + TB + frow * 256 + ((((rb >> 4) ^ krow_swz(frow))) << 5) + p4 * 8;
    ba[j][1] = lds0 + TB + (frow + 4) * 256 + ((((rb >> 4) ^ krow_swz(frow + 4))) << 5) + p4 * 8;
  }
  stage(smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const unsigned cur = (unsigned)((t & 1) * 2 * TB);
    if (t + 1 < nt) stage(smem + ((t + 1) & 1) * 2 * TB);
    Frag<true> af[MI], bf[4];
#pragma unroll
    for (int i = 0; i < MI; ++i) read_frag_rt<true>(af[i], aa[i][0] + cur, aa[i][1] + cur);
#pragma unroll
    for (int j = 0; j < 4; ++j) read_frag_rt<true>(bf[j], ba[j][0] + cur, ba[j][1] + cur);
    lgkm0(bf);
#pragma unroll
    for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(af[i].lo), "+v"(af[i].hi));
    bf16x8 av[MI], bv[4];
#pragma unroll
    for (int i = 0; i < MI; ++i) av[i] = frag_val(af[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = frag_val(bf[j]);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n_blk + wc * 64 + j * 16 + l16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m_blk + (wr * MI + i) * 16 + 4 * g + r;
        if (m < gm.Cout && n < N) atomicAdd(dw + (int64_t)m * N + n, acc[i][j][r]);
      }
    }
}

/* dw_tap[Cout, 9*Cin] (fp32) += dz[B*H*W, Cout]^T * patches(x[B*H*W, Cin]) for the 3x3 / stride 1 / pad 1 convolution; tap-major
 * columns as in mmrca_conv3x3_fwd.  Cin % 8 == 0, Cout % 8 == 0, bf16, 16-byte aligned operands.  No padding rows are read. */
extern "C" int mmrca_conv3x3_wgrad(const void* dz, const void* x, float* dw_tap, int B, int H, int W, int Cin, int Cout, int dtype,
                                   void* stream) {
  MMRCA_REQUIRE(dtype == MMRCA_BF16, "conv3x3_wgrad: bf16 only");
  MMRCA_REQUIRE(dz && x && dw_tap && B > 0 && H > 0 && W > 0, "conv3x3_wgrad: bad arguments");
  MMRCA_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && Cin >= 8 && Cout >= 8, "conv3x3_wgrad: Cin and Cout must be multiples of 8");
  MMRCA_REQUIRE((((uintptr_t)x | (uintptr_t)dz) & 15) == 0, "conv3x3_wgrad: operands must be 16-byte aligned");
  ConvGeom gm;
  conv_geom(&gm, B, H, W, Cin, Cout, 128);
  const int MI = Cout >= 128 ? 4 : (Cout > 32 ? 2 : 1);
  const int tiles_m = (Cout + 32 * MI - 1) / (32 * MI), tiles_n = (9 * Cin + 127) / 128;
  const int64_t P = (int64_t)B * H * W;
  // pixel ranges: enough blocks for ~4 rounds of the chip, each at least 64 steps deep
  int64_t want = (4 * 1024 + tiles_m * tiles_n - 1) / (tiles_m * tiles_n);
  int64_t len = (P + want - 1) / want;
  if (len < 2048) len = 2048;
  len = (len + 31) / 32 * 32;
  const int64_t ksplits = (P + len - 1) / len;
  MMRCA_REQUIRE(ksplits < 65536, "conv3x3_wgrad: too many pixel ranges");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)ksplits);
  const int lds = 4 * 128 * 32 * 2;
  if (MI == 4) hipLaunchKernelGGL((conv3x3_wgrad_k<4>), grid, dim3(256), lds, st, (const bf16_t*)dz, (const bf16_t*)x, dw_tap, gm, tiles_m, tiles_n, len);
  else if (MI == 2) hipLaunchKernelGGL((conv3x3_wgrad_k<2>), grid, dim3(256), lds, st, (const bf16_t*)dz, (const bf16_t*)x, dw_tap, gm, tiles_m, tiles_n, len);
  else hipLaunchKernelGGL((conv3x3_wgrad_k<1>), grid, dim3(256), lds, st, (const bf16_t*)dz, (const bf16_t*)x, dw_tap, gm, tiles_m, tiles_n, len);
  MMRCA_CHECK_LAUNCH("conv3x3_wgrad");
  return 0;
}
