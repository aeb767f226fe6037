// Conv-backbone kernels: EfficientNetV2-M/L (the image model the reference's MM_RCA runs,
// CVPR_code/multimodal_model.py:11-36, 113-126) and ShuffleNetV2 x2.0 (models.py:261-278) on NHWC activations.
//
// Layout: an activation is a row-major matrix [B*H*W, C] (channels contiguous), so
//   * every 1x1 convolution IS mmrca_gemm on the rows (weight [Cout, Cin] = torchvision's [Cout, Cin, 1, 1]);
//   * a full 3x3 convolution is mmrca_im2row3x3 + mmrca_gemm: the patch row is channel-major, col[c*9 + ky*3 + kx], which is
//     exactly the flattening of torchvision's [Cout, Cin, 3, 3] weight -- no re-laid weight copy exists; its input gradient is
//     mmrca_gemm (dcol = dz W) + mmrca_col2im3x3 (a gather over the <= 9 output pixels that read an input pixel: no atomics);
//   * depthwise 3x3, BatchNorm (batch statistics in training, running statistics in eval, running-stat update), SiLU / ReLU /
//     sigmoid, squeeze-excitation pooling and scaling, stochastic depth, 3x3/2 max pooling are the kernels below.
// Everything here is HBM-bound elementwise / reduction work: one thread per (pixel, channel) with the channel index
// fastest (coalesced), fp32 arithmetic, fp32 statistics and parameter gradients.
#include "common.h"

#define CONV_ACT_NONE 0
#define CONV_ACT_SILU 1
#define CONV_ACT_RELU 2
#define CONV_ACT_SIGMOID 3

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // (hardware reciprocal, 1 ulp: an IEEE division is ten instructions per element in the BatchNorm passes)
__device__ __forceinline__ float act_f(float u, int act) {
  if (act == CONV_ACT_SILU) return u * sigmoid_f(u);
  if (act == CONV_ACT_RELU) return u > 0.f ? u : 0.f;
  if (act == CONV_ACT_SIGMOID) return sigmoid_f(u);
  return u;
}
// d act(u) / du
__device__ __forceinline__ float act_grad_f(float u, int act) {
  if (act == CONV_ACT_SILU) { const float s = sigmoid_f(u); return s * (1.0f + u * (1.0f - s)); }
  if (act == CONV_ACT_RELU) return u > 0.f ? 1.f : 0.f;
  if (act == CONV_ACT_SIGMOID) { const float s = sigmoid_f(u); return s * (1.0f - s); }
  return 1.f;
}

static inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

// ---------------------------------------------------------------------------------------------------------------------
// NCHW fp32 images -> NHWC rows
// ---------------------------------------------------------------------------------------------------------------------
// Stencil kernels over a 1-D grid: hardware deals consecutive workgroups round robin to the 8 XCDs, so the workgroups that
// read rows y-1, y, y+1 of one image (a few ids apart) sit on different XCDs and each private L2 fetches the same input
// lines from HBM again -- the depthwise 3x3 kernels moved ~3x their algorithmic reads and ran at 2.1-2.4 TB/s.  xcd_block()
// gives XCD k the k-th contiguous eighth of the logical blocks instead: neighbouring rows meet in one L2.
__device__ __forceinline__ int64_t xcd_block() {
  const unsigned nwg = gridDim.x, orig = blockIdx.x;
  const unsigned q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  return (int64_t)((xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3));
}

template <typename T>
__global__ void nchw_to_rows_k(const float* __restrict__ img, T* __restrict__ x, int B, int C, int HW) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // (b, pix, c), c fastest
  if (idx >= (int64_t)B * HW * C) return;
  const int c = (int)(idx % C);
  const int64_t bp = idx / C;
  const int p = (int)(bp % HW), b = (int)(bp / HW);
  x[idx] = from_f<T>(img[((int64_t)b * C + c) * HW + p]);
}

extern "C" int mmrca_nchw_to_rows(const float* images, void* x, int B, int C, int H, int W, int dtype, void* stream) {
  MMRCA_REQUIRE(images && x && B > 0 && C > 0 && H > 0 && W > 0, "nchw_to_rows: bad arguments");
  const int64_t n = (int64_t)B * C * H * W;
  MMRCA_DISPATCH_DTYPE(dtype, "nchw_to_rows",
    hipLaunchKernelGGL(nchw_to_rows_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, images, (T*)x, B, C, H * W);)
  MMRCA_CHECK_LAUNCH("nchw_to_rows");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// im2row / col2im for 3x3, padding 1, stride 1 or 2
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void im2row3x3_k(const T* __restrict__ x, T* __restrict__ col, int B, int H, int W, int C, int Ho, int Wo, int stride,
                            int64_t ldk) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // (out pixel, c), c fastest
  if (idx >= (int64_t)B * Ho * Wo * C) return;
  const int c = (int)(idx % C);
  const int64_t op = idx / C;
  const int ox = (int)(op % Wo), oy = (int)((op / Wo) % Ho), b = (int)(op / ((int64_t)Wo * Ho));
  T* dst = col + op * ldk + (int64_t)c * 9;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
      const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
      dst[ky * 3 + kx] = in ? x[(((int64_t)b * H + iy) * W + ix) * C + c] : from_f<T>(0.f);
    }
}

// ---- 8 channels per thread (bf16, C % 8 == 0): nine 16-byte reads of x, a register transpose to the channel-major patch order
// (element 9 j + tap of the thread's 72-element chunk = channel j, tap), nine 16-byte stores
typedef __attribute__((ext_vector_type(8))) __bf16 im_b8;
__global__ void __launch_bounds__(256)
im2row3x3_v8_k(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, int B, int H, int W, int C, int Ho, int Wo, int stride, int64_t ldk) {
  const int C8 = C >> 3;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * Ho * Wo * C8) return;
  const int c0 = (int)(idx % C8) * 8;
  const int64_t op = idx / C8;
  const int ox = (int)(op % Wo), oy = (int)((op / Wo) % Ho), b = (int)(op / ((int64_t)Wo * Ho));
  im_b8 v[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
      im_b8 t;
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = (bf16_t)0.f;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) t = *reinterpret_cast<const im_b8*>(x + (((int64_t)b * H + iy) * W + ix) * C + c0);
      v[ky * 3 + kx] = t;
    }
  im_b8* dst = reinterpret_cast<im_b8*>(col + op * ldk + (int64_t)c0 * 9);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    im_b8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int e = 8 * i + k; o[k] = v[e % 9][e / 9]; }
    dst[i] = o;
  }
}
// the reverse: an input pixel gathers its (up to nine) contributions; every contribution is one element per channel of a
// 72-element chunk, read as nine 16-byte loads (the same bytes the scalar kernel touches, a ninth of the instructions)
__global__ void __launch_bounds__(256)
col2im3x3_v8_k(const bf16_t* __restrict__ dcol, bf16_t* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo, int stride, int64_t ldk) {
  const int C8 = C >> 3;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * H * W * C8) return;
  const int c0 = (int)(idx % C8) * 8;
  const int64_t ip = idx / C8;
  const int ix = (int)(ip % W), iy = (int)((ip / W) % H), b = (int)(ip / ((int64_t)W * H));
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = iy + 1 - ky;
    if (ty < 0 || ty % stride) continue;
    const int oy = ty / stride;
    if (oy >= Ho) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = ix + 1 - kx;
      if (tx < 0 || tx % stride) continue;
      const int ox = tx / stride;
      if (ox >= Wo) continue;
      const im_b8* src = reinterpret_cast<const im_b8*>(dcol + (((int64_t)b * Ho + oy) * Wo + ox) * ldk + (int64_t)c0 * 9);
      const int tap = ky * 3 + kx;
      im_b8 ch[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) ch[i] = src[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int e = 9 * j + tap; s[j] += (float)ch[e / 8][e % 8]; }
    }
  }
  im_b8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)s[j];
  *reinterpret_cast<im_b8*>(dx + ip * C + c0) = o;
}
static bool im_v8_ok(int C, int64_t ldk, int dtype, const void* a, const void* b) {
  return dtype == MMRCA_BF16 && C % 8 == 0 && ldk % 8 == 0 && ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0;
}

extern "C" int mmrca_im2row3x3(const void* x, void* col, int B, int H, int W, int C, int stride, int64_t ldk, int dtype, void* stream) {
  MMRCA_REQUIRE(x && col && B > 0 && H > 0 && W > 0 && C > 0 && (stride == 1 || stride == 2) && ldk >= 9LL * C, "im2row3x3: bad arguments");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  if (im_v8_ok(C, ldk, dtype, x, col)) {
    const int64_t n8 = (int64_t)B * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(im2row3x3_v8_k, dim3(blocks_for(n8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)col, B, H, W, C,
                       Ho, Wo, stride, ldk);
    MMRCA_CHECK_LAUNCH("im2row3x3(v8)");
    return 0;
  }
  const int64_t n = (int64_t)B * Ho * Wo * C;
  MMRCA_DISPATCH_DTYPE(dtype, "im2row3x3",
    hipLaunchKernelGGL(im2row3x3_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)col, B, H, W, C,
                       Ho, Wo, stride, ldk);)
  MMRCA_CHECK_LAUNCH("im2row3x3");
  return 0;
}

// ---- tap-major patch order, k = tap * C + c (bf16, C % 8 == 0): one 16-byte load and one 16-byte store per thread, lanes
// contiguous on both sides (the channel-major order above writes 16 bytes at a 144-byte lane stride and reads its 72-element
// chunk nine times in col2im: 1.05 TB/s).  The GEMM that follows takes the weight permuted to [C_out, 9, C_in].
__global__ void __launch_bounds__(256)
im2row3x3_tap_v8_k(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, int B, int H, int W, int C, int Ho, int Wo, int stride, int64_t ldk) {
  const int C8 = C >> 3;
  const int64_t idx = xcd_block() * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * Ho * Wo * 9 * C8) return;
  const int c0 = (int)(idx % C8) * 8;
  const int tap = (int)((idx / C8) % 9);
  const int64_t op = idx / (9 * C8);
  const int ox = (int)(op % Wo), oy = (int)((op / Wo) % Ho), b = (int)(op / ((int64_t)Wo * Ho));
  const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
  im_b8 t;
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = (bf16_t)0.f;
  if (iy >= 0 && iy < H && ix >= 0 && ix < W) t = *reinterpret_cast<const im_b8*>(x + (((int64_t)b * H + iy) * W + ix) * C + c0);
  *reinterpret_cast<im_b8*>(col + op * ldk + (int64_t)tap * C + c0) = t;
}
__global__ void __launch_bounds__(256)
col2im3x3_tap_v8_k(const bf16_t* __restrict__ dcol, bf16_t* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo, int stride, int64_t ldk) {
  const int C8 = C >> 3;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * H * W * C8) return;
  const int c0 = (int)(idx % C8) * 8;
  const int64_t ip = idx / C8;
  const int ix = (int)(ip % W), iy = (int)((ip / W) % H), b = (int)(ip / ((int64_t)W * H));
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = iy + 1 - ky;
    if (ty < 0 || ty % stride) continue;
    const int oy = ty / stride;
    if (oy >= Ho) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = ix + 1 - kx;
      if (tx < 0 || tx % stride) continue;
      const int ox = tx / stride;
      if (ox >= Wo) continue;
      const im_b8 v = *reinterpret_cast<const im_b8*>(dcol + (((int64_t)b * Ho + oy) * Wo + ox) * ldk + (int64_t)(ky * 3 + kx) * C + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
    }
  }
  im_b8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)s[j];
  *reinterpret_cast<im_b8*>(dx + ip * C + c0) = o;
}

extern "C" int mmrca_im2row3x3_tap(const void* x, void* col, int B, int H, int W, int C, int stride, int64_t ldk, int dtype, void* stream) {
  MMRCA_REQUIRE(x && col && B > 0 && H > 0 && W > 0 && C > 0 && (stride == 1 || stride == 2) && ldk >= 9LL * C, "im2row3x3_tap: bad arguments");
  MMRCA_REQUIRE(im_v8_ok(C, ldk, dtype, x, col), "im2row3x3_tap: bf16, C %% 8 == 0, ldk %% 8 == 0 and 16-byte aligned operands only (C=%d)", C);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int64_t n = (int64_t)B * Ho * Wo * 9 * (C / 8);
  hipLaunchKernelGGL(im2row3x3_tap_v8_k, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)col, B, H, W, C,
                     Ho, Wo, stride, ldk);
  MMRCA_CHECK_LAUNCH("im2row3x3_tap");
  return 0;
}
extern "C" int mmrca_col2im3x3_tap(const void* dcol, void* dx, int B, int H, int W, int C, int stride, int64_t ldk, int dtype, void* stream) {
  MMRCA_REQUIRE(dcol && dx && B > 0 && H > 0 && W > 0 && C > 0 && (stride == 1 || stride == 2) && ldk >= 9LL * C, "col2im3x3_tap: bad arguments");
  MMRCA_REQUIRE(im_v8_ok(C, ldk, dtype, dcol, dx), "col2im3x3_tap: bf16, C %% 8 == 0, ldk %% 8 == 0 and 16-byte aligned operands only (C=%d)", C);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int64_t n8 = (int64_t)B * H * W * (C / 8);
  hipLaunchKernelGGL(col2im3x3_tap_v8_k, dim3(blocks_for(n8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dcol, (bf16_t*)dx, B, H, W, C,
                     Ho, Wo, stride, ldk);
  MMRCA_CHECK_LAUNCH("col2im3x3_tap");
  return 0;
}

// dx[b, iy, ix, c] = sum over (ky, kx) with (iy + 1 - ky) % stride == 0 etc. of dcol[out pixel][c*9 + ky*3 + kx]
template <typename T>
__global__ void col2im3x3_k(const T* __restrict__ dcol, T* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo, int stride,
                            int64_t ldk) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // (in pixel, c), c fastest
  if (idx >= (int64_t)B * H * W * C) return;
  const int c = (int)(idx % C);
  const int64_t ip = idx / C;
  const int ix = (int)(ip % W), iy = (int)((ip / W) % H), b = (int)(ip / ((int64_t)W * H));
  float s = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = iy + 1 - ky;
    if (ty < 0 || ty % stride) continue;
    const int oy = ty / stride;
    if (oy >= Ho) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = ix + 1 - kx;
      if (tx < 0 || tx % stride) continue;
      const int ox = tx / stride;
      if (ox >= Wo) continue;
      s += to_f(dcol[(((int64_t)b * Ho + oy) * Wo + ox) * ldk + (int64_t)c * 9 + ky * 3 + kx]);
    }
  }
  dx[idx] = from_f<T>(s);
}

extern "C" int mmrca_col2im3x3(const void* dcol, void* dx, int B, int H, int W, int C, int stride, int64_t ldk, int dtype, void* stream) {
  MMRCA_REQUIRE(dcol && dx && B > 0 && H > 0 && W > 0 && C > 0 && (stride == 1 || stride == 2) && ldk >= 9LL * C, "col2im3x3: bad arguments");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  if (im_v8_ok(C, ldk, dtype, dcol, dx)) {
    const int64_t n8 = (int64_t)B * H * W * (C / 8);
    hipLaunchKernelGGL(col2im3x3_v8_k, dim3(blocks_for(n8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dcol, (bf16_t*)dx, B, H, W, C,
                       Ho, Wo, stride, ldk);
    MMRCA_CHECK_LAUNCH("col2im3x3(v8)");
    return 0;
  }
  const int64_t n = (int64_t)B * H * W * C;
  MMRCA_DISPATCH_DTYPE(dtype, "col2im3x3",
    hipLaunchKernelGGL(col2im3x3_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)dcol, (T*)dx, B, H, W, C,
                       Ho, Wo, stride, ldk);)
  MMRCA_CHECK_LAUNCH("col2im3x3");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// depthwise 3x3, padding 1, stride 1 or 2.  w: [C, 9] in `dtype` (torchvision's [C, 1, 3, 3]); dw: fp32 [C, 9], +=
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void dwconv3x3_fwd_k(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y, int B, int H, int W, int C,
                                int Ho, int Wo, int stride) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * Ho * Wo * C) return;
  const int c = (int)(idx % C);
  const int64_t op = idx / C;
  const int ox = (int)(op % Wo), oy = (int)((op / Wo) % Ho), b = (int)(op / ((int64_t)Wo * Ho));
  float s = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) s = fmaf(to_f(x[(((int64_t)b * H + iy) * W + ix) * C + c]), to_f(w[c * 9 + ky * 3 + kx]), s);
    }
  y[idx] = from_f<T>(s);
}

template <typename T>
__global__ void dwconv3x3_bwd_data_k(const T* __restrict__ dy, const T* __restrict__ w, T* __restrict__ dx, int B, int H, int W, int C,
                                     int Ho, int Wo, int stride) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * H * W * C) return;
  const int c = (int)(idx % C);
  const int64_t ip = idx / C;
  const int ix = (int)(ip % W), iy = (int)((ip / W) % H), b = (int)(ip / ((int64_t)W * H));
  float s = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = iy + 1 - ky;
    if (ty < 0 || ty % stride) continue;
    const int oy = ty / stride;
    if (oy >= Ho) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = ix + 1 - kx;
      if (tx < 0 || tx % stride) continue;
      const int ox = tx / stride;
      if (ox >= Wo) continue;
      s = fmaf(to_f(dy[(((int64_t)b * Ho + oy) * Wo + ox) * C + c]), to_f(w[c * 9 + ky * 3 + kx]), s);
    }
  }
  dx[idx] = from_f<T>(s);
}

// dw[c][tap] += sum over output pixels of dy * x(tap).  Block = 256 threads = 64 channels x 4 pixel lanes; each block walks a
// slice of the output pixels, reduces its 4 pixel lanes in LDS and adds 9 values per channel atomically.
template <typename T>
__global__ void dwconv3x3_bwd_weight_k(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ dw, int B, int H, int W,
                                       int C, int Ho, int Wo, int stride, int64_t pix_per_block) {
  __shared__ float red[4][64][9];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  const int64_t npix = (int64_t)B * Ho * Wo;
  const int64_t p0 = (int64_t)blockIdx.y * pix_per_block;
  const int64_t p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
  if (c < C) {
    for (int64_t op = p0 + pl; op < p1; op += 4) {
      const int ox = (int)(op % Wo), oy = (int)((op / Wo) % Ho), b = (int)(op / ((int64_t)Wo * Ho));
      const float g = to_f(dy[op * C + c]);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) acc[ky * 3 + kx] = fmaf(g, to_f(x[(((int64_t)b * H + iy) * W + ix) * C + c]), acc[ky * 3 + kx]);
        }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) red[pl][cl][t] = acc[t];
  __syncthreads();
  if (pl == 0 && c < C) {
#pragma unroll
    for (int t = 0; t < 9; ++t) atomicAdd(dw + (int64_t)c * 9 + t, red[0][cl][t] + red[1][cl][t] + red[2][cl][t] + red[3][cl][t]);
  }
}

// ---- 8 channels per thread (bf16, C % 8 == 0: every EfficientNetV2 depthwise layer): 16-byte loads of x / dy and of the 72
// contiguous weights of the thread's channels; the index arithmetic (three divisions by run-time extents) is paid once per 8
// channels and 9 taps instead of once per element.  Same products and the same fp32 summation order per channel.
typedef __attribute__((ext_vector_type(8))) __bf16 cv_b8;

__device__ __forceinline__ void dw_load_w8(const bf16_t* __restrict__ w, int c0, float (&wf)[8][9]) {
  // w[c0*9 .. c0*9+71]: 9 loads of 8 bf16 (c0 % 8 == 0 -> 16-byte aligned)
  const cv_b8* p = reinterpret_cast<const cv_b8*>(w + (int64_t)c0 * 9);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const cv_b8 v = p[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int e = i * 8 + j; wf[e / 9][e % 9] = (float)v[j]; }
  }
}

__global__ void __launch_bounds__(256)
dwconv3x3_fwd_v8_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y, int B, int H, int W, int C,
                   int Ho, int Wo, int stride) {
  const int C8 = C >> 3;
  const int64_t idx = xcd_block() * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * Ho * Wo * C8) return;
  const int c0 = (int)(idx % C8) * 8;
  const int64_t op = idx / C8;
  const int ox = (int)(op % Wo), oy = (int)((op / Wo) % Ho), b = (int)(op / ((int64_t)Wo * Ho));
  float wf[8][9];
  dw_load_w8(w, c0, wf);
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
        const cv_b8 v = *reinterpret_cast<const cv_b8*>(x + (((int64_t)b * H + iy) * W + ix) * C + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = fmaf((float)v[j], wf[j][ky * 3 + kx], s[j]);
      }
    }
  cv_b8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)s[j];
  *reinterpret_cast<cv_b8*>(y + op * C + c0) = o;
}

// Stride-1 layers (all but three of EfficientNetV2's depthwise convolutions): a thread owns FOUR consecutive output pixels of a
// row for its 8 channels -- the 72 weights are fetched once per four outputs and the 3 x 6 input window serves all of them
// (6.75 sixteen-byte loads per output instead of 18).  FLIP = the input gradient (the same correlation with the kernel turned
// by 180 degrees).  Per output the products are added in the same (ky, kx) order as in the one-pixel kernels: identical bits.
template <bool FLIP>
__global__ void __launch_bounds__(256)
dw3x3_s1_strip_v8_k(const bf16_t* __restrict__ in, const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int B, int H, int W, int C) {
  const int C8 = C >> 3, XG = (W + 3) >> 2;
  const int64_t idx = xcd_block() * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * H * XG * C8) return;
  const int c0 = (int)(idx % C8) * 8;
  const int64_t g = idx / C8;
  const int x0 = (int)(g % XG) * 4, y = (int)((g / XG) % H), b = (int)(g / ((int64_t)XG * H));
  float wf[8][9];
  dw_load_w8(w, c0, wf);
  float s[4][8];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int j = 0; j < 8; ++j) s[o][j] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = y + ky - 1;
    if (iy < 0 || iy >= H) continue;
    const bf16_t* row = in + ((int64_t)b * H + iy) * W * C + c0;
#pragma unroll
    for (int col = 0; col < 6; ++col) {
      const int ix = x0 + col - 1;
      if (ix < 0 || ix >= W) continue;
      const cv_b8 v = *reinterpret_cast<const cv_b8*>(row + (int64_t)ix * C);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int kx = col - o;
        if (kx < 0 || kx > 2) continue;
        const int tap = FLIP ? (2 - ky) * 3 + (2 - kx) : ky * 3 + kx;
#pragma unroll
        for (int j = 0; j < 8; ++j) s[o][j] = fmaf((float)v[j], wf[j][tap], s[o][j]);
      }
    }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    if (x0 + o >= W) continue;
    cv_b8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)s[o][j];
    *reinterpret_cast<cv_b8*>(out + (((int64_t)b * H + y) * W + x0 + o) * C + c0) = r;
  }
}

__global__ void __launch_bounds__(256)
dwconv3x3_bwd_data_v8_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ w, bf16_t* __restrict__ dx, int B, int H, int W, int C,
                        int Ho, int Wo, int stride) {
  const int C8 = C >> 3;
  const int64_t idx = xcd_block() * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * H * W * C8) return;
  const int c0 = (int)(idx % C8) * 8;
  const int64_t ip = idx / C8;
  const int ix = (int)(ip % W), iy = (int)((ip / W) % H), b = (int)(ip / ((int64_t)W * H));
  float wf[8][9];
  dw_load_w8(w, c0, wf);
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = iy + 1 - ky;
    if (ty < 0 || ty % stride) continue;
    const int oy = ty / stride;
    if (oy >= Ho) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = ix + 1 - kx;
      if (tx < 0 || tx % stride) continue;
      const int ox = tx / stride;
      if (ox >= Wo) continue;
      const cv_b8 v = *reinterpret_cast<const cv_b8*>(dy + (((int64_t)b * Ho + oy) * Wo + ox) * C + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] = fmaf((float)v[j], wf[j][ky * 3 + kx], s[j]);
    }
  }
  cv_b8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)s[j];
  *reinterpret_cast<cv_b8*>(dx + ip * C + c0) = o;
}

// weight gradient: block = 256 threads = 8 channel groups (64 channels) x 32 pixel lanes; LDS reduction over the pixel lanes
__global__ void __launch_bounds__(256)
dwconv3x3_bwd_weight_v8_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ dw, int B, int H, int W,
                          int C, int Ho, int Wo, int stride, int64_t pix_per_block) {
  __shared__ float red[4][8][73];             // [wave][channel group][8 x 9 (+1 pad)]: the 8 pixel lanes of a wave are reduced by shuffles first
  const int cg = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cg * 8;
  float acc[8][9];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[j][t] = 0.f;
  const int64_t npix = (int64_t)B * Ho * Wo;
  const int64_t p0 = (int64_t)blockIdx.y * pix_per_block;
  const int64_t p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
  if (c0 < C) {
    for (int64_t op = p0 + pl; op < p1; op += 32) {
      const int ox = (int)(op % Wo), oy = (int)((op / Wo) % Ho), b = (int)(op / ((int64_t)Wo * Ho));
      const cv_b8 gv = *reinterpret_cast<const cv_b8*>(dy + op * C + c0);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const cv_b8 xv = *reinterpret_cast<const cv_b8*>(x + (((int64_t)b * H + iy) * W + ix) * C + c0);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j][ky * 3 + kx] = fmaf((float)gv[j], (float)xv[j], acc[j][ky * 3 + kx]);
          }
        }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v = acc[j][t];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);      // lanes with the same channel group
      if ((threadIdx.x & 63) < 8) red[threadIdx.x >> 6][cg][j * 9 + t] = v;
    }
  __syncthreads();
  for (int e = threadIdx.x; e < 8 * 72; e += 256) {
    const int g = e / 72, r = e % 72;
    const int c = blockIdx.x * 64 + g * 8 + r / 9;
    if (c < C) atomicAdd(dw + (int64_t)c * 9 + r % 9, (red[0][g][r] + red[1][g][r]) + (red[2][g][r] + red[3][g][r]));
  }
}

// stride 1: four consecutive output pixels of a row per thread and iteration -- 4 loads of dy and the 3 x 6 window of x serve 36
// (pixel, tap) pairs (5.5 sixteen-byte loads per pixel instead of 10); same reduction afterwards
__global__ void __launch_bounds__(256)
dwconv3x3_bwd_weight_s1_strip_v8_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ dw, int B, int H, int W,
                                   int C, int64_t grp_per_block) {
  __shared__ float red[4][8][73];
  const int cg = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cg * 8, XG = (W + 3) >> 2;
  float acc[8][9];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[j][t] = 0.f;
  const int64_t ngrp = (int64_t)B * H * XG;
  const int64_t g0 = (int64_t)blockIdx.y * grp_per_block;
  const int64_t g1 = g0 + grp_per_block < ngrp ? g0 + grp_per_block : ngrp;
  if (c0 < C) {
    for (int64_t g = g0 + pl; g < g1; g += 32) {
      const int x0 = (int)(g % XG) * 4, y = (int)((g / XG) % H), b = (int)(g / ((int64_t)XG * H));
      float gv[4][8];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        cv_b8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f;
        if (x0 + o < W) v = *reinterpret_cast<const cv_b8*>(dy + (((int64_t)b * H + y) * W + x0 + o) * C + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[o][j] = (float)v[j];
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = y + ky - 1;
        if (iy < 0 || iy >= H) continue;
        const bf16_t* row = x + ((int64_t)b * H + iy) * W * C + c0;
#pragma unroll
        for (int col = 0; col < 6; ++col) {
          const int ix = x0 + col - 1;
          if (ix < 0 || ix >= W) continue;
          const cv_b8 xv = *reinterpret_cast<const cv_b8*>(row + (int64_t)ix * C);
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            const int kx = col - o;
            if (kx < 0 || kx > 2) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j][ky * 3 + kx] = fmaf(gv[o][j], (float)xv[j], acc[j][ky * 3 + kx]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v = acc[j][t];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      if ((threadIdx.x & 63) < 8) red[threadIdx.x >> 6][cg][j * 9 + t] = v;
    }
  __syncthreads();
  for (int e = threadIdx.x; e < 8 * 72; e += 256) {
    const int g = e / 72, r = e % 72;
    const int c = blockIdx.x * 64 + g * 8 + r / 9;
    if (c < C) atomicAdd(dw + (int64_t)c * 9 + r % 9, (red[0][g][r] + red[1][g][r]) + (red[2][g][r] + red[3][g][r]));
  }
}


// Round 3: the same with TWO output rows per thread and 32-bit index arithmetic.  The kernels above were neither HBM- nor
// L2-bound (2.1-2.4 TB/s, 47 % VALU-busy): a third of their ~970 VALU instructions per thread were 64-bit divisions / address
// multiplies of the index decomposition, another quarter the bf16 -> fp32 conversion of the 72 weights, repeated for every
// four outputs.  Here: unsigned 32-bit indices (tensors below 4 GiB: the launcher checks), eight outputs per thread (the 4 x 6
// input window serves 72 (pixel, tap) pairs: 3 sixteen-byte loads per output instead of 4.5; weights converted once per eight).
template <bool FLIP>
__global__ void __launch_bounds__(256)
dw3x3_s1_strip2_v8_k(const bf16_t* __restrict__ in, const bf16_t* __restrict__ w, bf16_t* __restrict__ out, unsigned B, unsigned H, unsigned W,
                     unsigned C) {
  const unsigned C8 = C >> 3, XG = (W + 3) >> 2, YG = (H + 1) >> 1;
  const unsigned idx = (unsigned)xcd_block() * 256u + threadIdx.x;
  if (idx >= B * YG * XG * C8) return;
  const unsigned c0 = (idx % C8) * 8;
  unsigned g = idx / C8;
  const unsigned x0 = (g % XG) * 4;
  g /= XG;
  const unsigned y0 = (g % YG) * 2, b = g / YG;
  // all 24 loads of the 4 x 6 window are issued before anything waits: out-of-image positions load a clamped (valid) address
  // and are zeroed afterwards -- with the loads inside `if (inside)` regions the compiler waited once per row
  cv_b8 v[4][6];
  bool rv[4], cvd[6];
  unsigned coff[6];
#pragma unroll
  for (int col = 0; col < 6; ++col) {
    const int ix = (int)x0 + col - 1;
    cvd[col] = ix >= 0 && ix < (int)W;
    coff[col] = (unsigned)(ix < 0 ? 0 : (ix >= (int)W ? (int)W - 1 : ix)) * C;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int iy = (int)y0 + r - 1;
    rv[r] = iy >= 0 && iy < (int)H;
    const unsigned rowoff = ((b * H + (unsigned)(iy < 0 ? 0 : (iy >= (int)H ? (int)H - 1 : iy))) * W) * C + c0;
#pragma unroll
    for (int col = 0; col < 6; ++col) v[r][col] = *reinterpret_cast<const cv_b8*>(in + (rowoff + coff[col]));
  }
  float wf[8][9];
  dw_load_w8(w, (int)c0, wf);
  float s[2][4][8];
#pragma unroll
  for (int oy = 0; oy < 2; ++oy)
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int j = 0; j < 8; ++j) s[oy][o][j] = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int col = 0; col < 6; ++col) {
      const bool ok = rv[r] && cvd[col];
      float vf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) vf[j] = ok ? (float)v[r][col][j] : 0.f;
#pragma unroll
      for (int oy = 0; oy < 2; ++oy) {
        const int ky = r - oy;
        if (ky < 0 || ky > 2) continue;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int kx = col - o;
          if (kx < 0 || kx > 2) continue;
          const int tap = FLIP ? (2 - ky) * 3 + (2 - kx) : ky * 3 + kx;
#pragma unroll
          for (int j = 0; j < 8; ++j) s[oy][o][j] = fmaf(vf[j], wf[j][tap], s[oy][o][j]);
        }
      }
    }
  }
#pragma unroll
  for (int oy = 0; oy < 2; ++oy) {
    if (y0 + oy >= H) continue;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (x0 + o >= W) continue;
      cv_b8 r;
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = (bf16_t)s[oy][o][j];
      *reinterpret_cast<cv_b8*>(out + (((b * H + y0 + oy) * W + x0 + o) * C + c0)) = r;
    }
  }
}

// weight gradient, same strip.  A thread owns one 8-channel group (64 consecutive groups per wave: every load instruction is one
// contiguous KiB) and one of the block's four pixel lanes; a group = 2 rows x 4 pixels: 8 loads of dy and the 4 x 6 window of x,
// all issued before the first use (clamped addresses, zeroed afterwards), serve 72 (pixel, tap) pairs.  The four pixel lanes
// meet in LDS (ds_add_f32), then one global atomic per (channel, tap) and block.
template <int R, bool WS>         // output rows per group (R + 2 input rows): 2 needs ~260 registers, 1 fits two waves per SIMD
__global__ void __launch_bounds__(256)
dwconv3x3_bwd_weight_s1_strip2_v8_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ dw, unsigned B, unsigned H,
                                    unsigned W, unsigned C, unsigned grp_per_block, float* __restrict__ ws) {
  __shared__ float red[64 * 73];          // pitch 73: the 64 lanes of a ds_add fall on 64 different banks
  const unsigned cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const unsigned C8 = C >> 3, cgrp = blockIdx.x * 64 + cl, c0 = cgrp * 8;
  const unsigned XG = (W + 3) >> 2, YG = (H + R - 1) / R;
  for (unsigned e = threadIdx.x; e < 64 * 73; e += 256) red[e] = 0.f;
  float acc[8][9];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[j][t] = 0.f;
  const unsigned ngrp = B * YG * XG;
  const unsigned g0 = blockIdx.y * grp_per_block;
  const unsigned g1 = g0 + grp_per_block < ngrp ? g0 + grp_per_block : ngrp;
  if (cgrp < C8) {
    for (unsigned g = g0 + pl; g < g1; g += 4) {
      const unsigned x0 = (g % XG) * 4, t = g / XG, y0 = (t % YG) * R, b = t / YG;
      cv_b8 gvb[R][4], xv[R + 2][6];
      bool rv[R + 2], cvd[6], gok[R][4];
      unsigned coff[6];
#pragma unroll
      for (int col = 0; col < 6; ++col) {
        const int ix = (int)x0 + col - 1;
        cvd[col] = ix >= 0 && ix < (int)W;
        coff[col] = (unsigned)(ix < 0 ? 0 : (ix >= (int)W ? (int)W - 1 : ix)) * C;
      }
#pragma unroll
      for (int oy = 0; oy < R; ++oy) {
        const unsigned yy = y0 + oy < H ? y0 + oy : H - 1;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          gok[oy][o] = y0 + oy < H && x0 + o < W;
          gvb[oy][o] = *reinterpret_cast<const cv_b8*>(dy + (((b * H + yy) * W) * C + c0 + coff[o + 1]));
        }
      }
#pragma unroll
      for (int r = 0; r < R + 2; ++r) {
        const int iy = (int)y0 + r - 1;
        rv[r] = iy >= 0 && iy < (int)H;
        const unsigned rowoff = ((b * H + (unsigned)(iy < 0 ? 0 : (iy >= (int)H ? (int)H - 1 : iy))) * W) * C + c0;
#pragma unroll
        for (int col = 0; col < 6; ++col) xv[r][col] = *reinterpret_cast<const cv_b8*>(x + (rowoff + coff[col]));
      }
      float gv[R][4][8];
#pragma unroll
      for (int oy = 0; oy < R; ++oy)
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
          for (int j = 0; j < 8; ++j) gv[oy][o][j] = gok[oy][o] ? (float)gvb[oy][o][j] : 0.f;
#pragma unroll
      for (int r = 0; r < R + 2; ++r) {
#pragma unroll
        for (int col = 0; col < 6; ++col) {
          const bool ok = rv[r] && cvd[col];
          float xf[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) xf[j] = ok ? (float)xv[r][col][j] : 0.f;
#pragma unroll
          for (int oy = 0; oy < R; ++oy) {
            const int ky = r - oy;
            if (ky < 0 || ky > 2) continue;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
              const int kx = col - o;
              if (kx < 0 || kx > 2) continue;
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[j][ky * 3 + kx] = fmaf(gv[oy][o][j], xf[j], acc[j][ky * 3 + kx]);
            }
          }
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) atomicAdd(&red[cl * 73 + j * 9 + t], acc[j][t]);
  __syncthreads();
  for (unsigned e = threadIdx.x; e < 64 * 72; e += 256) {
    const unsigned c = blockIdx.x * 512 + e / 9;           // e = cl * 72 + j * 9 + t -> channel (64 blk + cl) * 8 + j = 512 blk + e / 9
    const float v = red[(e / 72) * 73 + e % 72];
    // WS: this block's 4,608 sums go to its own row of the workspace (dw_wg_reduce_k adds the rows up).  Without a workspace they
    // are device-scope atomics, which the 8 XCDs' L2s cannot absorb: ~50 per ns in all, a third of the kernel's time at 768 blocks.
    if (WS) ws[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4608 + e] = v;
    else if (c < C) atomicAdd(dw + (int64_t)c * 9 + e % 9, v);
  }
}

// Streaming form of the weight gradient (needs a workspace): thread = (8-channel group c8, pixel lane), flat id = lane * C8 + c8, so
// consecutive threads read consecutive 16-byte chunks and, at any moment, the resident threads cover P_L consecutive pixel
// groups over ALL channels -- the grid walks the two tensors front to back like the forward kernel does.  The slice form above
// gives a block one 1-KiB column slice of every pixel row (stride 2C bytes) and measured 1.4-1.8 TB/s whatever its block count or
// reduction.  Every thread keeps its 72 sums and writes them to ws[lane][c8 * 72 ..]; dw_wg_reduce_k adds the lanes up.
template <int R>
__global__ void __launch_bounds__(256)
dwconv3x3_bwd_weight_s1_flat_v8_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, unsigned B, unsigned H, unsigned W, unsigned C,
                                  unsigned lanes, float* __restrict__ ws) {
  const unsigned C8 = C >> 3, XG = (W + 3) >> 2;
  const unsigned tid = blockIdx.x * 256u + threadIdx.x;
  if (tid >= C8 * lanes) return;
  const unsigned c8 = tid % C8, lane = tid / C8, c0 = c8 * 8;
  const unsigned YG = (H + R - 1) / R, ngrp = B * YG * XG;
  float acc[8][9];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[j][t] = 0.f;
  for (unsigned g = lane; g < ngrp; g += lanes) {
    const unsigned x0 = (g % XG) * 4, t = g / XG, y0 = (t % YG) * R, b = t / YG;
    cv_b8 gvb[R][4], xv[R + 2][6];
    bool rv[R + 2], cvd[6], gok[R][4];
    unsigned coff[6];
#pragma unroll
    for (int col = 0; col < 6; ++col) {
      const int ix = (int)x0 + col - 1;
      cvd[col] = ix >= 0 && ix < (int)W;
      coff[col] = (unsigned)(ix < 0 ? 0 : (ix >= (int)W ? (int)W - 1 : ix)) * C;
    }
#pragma unroll
    for (int oy = 0; oy < R; ++oy) {
      const unsigned yy = y0 + oy < H ? y0 + oy : H - 1;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        gok[oy][o] = y0 + oy < H && x0 + o < W;
        gvb[oy][o] = *reinterpret_cast<const cv_b8*>(dy + (((b * H + yy) * W) * C + c0 + coff[o + 1]));
      }
    }
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
      const int iy = (int)y0 + r - 1;
      rv[r] = iy >= 0 && iy < (int)H;
      const unsigned rowoff = ((b * H + (unsigned)(iy < 0 ? 0 : (iy >= (int)H ? (int)H - 1 : iy))) * W) * C + c0;
#pragma unroll
      for (int col = 0; col < 6; ++col) xv[r][col] = *reinterpret_cast<const cv_b8*>(x + (rowoff + coff[col]));
    }
    float gv[R][4][8];
#pragma unroll
    for (int oy = 0; oy < R; ++oy)
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[oy][o][j] = gok[oy][o] ? (float)gvb[oy][o][j] : 0.f;
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
#pragma unroll
      for (int col = 0; col < 6; ++col) {
        const bool ok = rv[r] && cvd[col];
        float xf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[j] = ok ? (float)xv[r][col][j] : 0.f;
#pragma unroll
        for (int oy = 0; oy < R; ++oy) {
          const int ky = r - oy;
          if (ky < 0 || ky > 2) continue;
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            const int kx = col - o;
            if (kx < 0 || kx > 2) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j][ky * 3 + kx] = fmaf(gv[oy][o][j], xf[j], acc[j][ky * 3 + kx]);
          }
        }
      }
    }
  }
  float* o = ws + ((int64_t)lane * C8 + c8) * 72;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) o[j * 9 + t] = acc[j][t];
}

// dw[i] += sum over the row ranges of ws[range][i]  (i = channel * 9 + tap; a workspace row holds slices * 512 channels).
// block = 64 outputs x 4 range lanes, grid.y = 8 range chunks (one atomic per output and chunk): a thread walks at most
// nranges / 32 rows, several loads in flight -- a single thread per output walked 1,500 rows one load at a time (0.4 ms).
__global__ void __launch_bounds__(256)
dw_wg_reduce_k(const float* __restrict__ ws, float* __restrict__ dw, int nranges, int n, int64_t stride) {
  __shared__ float red[4][64];
  const int il = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + il;
  const int per = (nranges + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = r0 + per < nranges ? r0 + per : nranges;
  float s = 0.f;
  if (i < n) {
    int r = r0 + rl;
    for (; r + 12 < r1; r += 16) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ws[(int64_t)(r + 4 * u) * stride + i];
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (; r < r1; r += 4) s += ws[(int64_t)r * stride + i];
  }
  red[rl][il] = s;
  __syncthreads();
  if (rl == 0 && i < n) atomicAdd(dw + i, (red[0][il] + red[1][il]) + (red[2][il] + red[3][il]));
}

static bool dw_v8_ok(int C, int dtype, const void* a, const void* b, const void* c) {
  return dtype == MMRCA_BF16 && C % 8 == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}

static const bool g_dw_flat = !(getenv("MMRCA_DW_WG_FLAT") && atoi(getenv("MMRCA_DW_WG_FLAT")) == 0);
static const int g_dw_flat_rows = getenv("MMRCA_DW_WG_FLAT_ROWS") ? atoi(getenv("MMRCA_DW_WG_FLAT_ROWS")) : 2;
// threads of the streaming weight gradient = what is resident at once: the two-row form needs > 128 registers (one wave per SIMD:
// 256 CUs x 4 x 64 lanes), the one-row form fits three
static const int g_dw_flat_threads = getenv("MMRCA_DW_WG_FLAT_THREADS") ? atoi(getenv("MMRCA_DW_WG_FLAT_THREADS")) : (g_dw_flat_rows == 2 ? 65536 : 196608);
static const int g_dw_blocks_ws = getenv("MMRCA_DW_WG_BLOCKS_WS") ? atoi(getenv("MMRCA_DW_WG_BLOCKS_WS")) : 768;
static const int g_dw_blocks = getenv("MMRCA_DW_WG_BLOCKS") ? atoi(getenv("MMRCA_DW_WG_BLOCKS")) : 768;
static const bool g_dw_strip2 = !(getenv("MMRCA_DW_STRIP2") && atoi(getenv("MMRCA_DW_STRIP2")) == 0);
static const bool g_dw_strip = !(getenv("MMRCA_DW_STRIP") && atoi(getenv("MMRCA_DW_STRIP")) == 0);
extern "C" int mmrca_dwconv3x3_fwd(const void* x, const void* w, void* y, int B, int H, int W, int C, int stride, int dtype, void* stream) {
  MMRCA_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0 && C > 0 && (stride == 1 || stride == 2), "dwconv3x3_fwd: bad arguments");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const bool small32 = (int64_t)B * H * W * C * 2 < (1ll << 32);          // 32-bit element offsets in the two-row strip kernels
  if (dw_v8_ok(C, dtype, x, w, y) && stride == 1 && g_dw_strip && g_dw_strip2 && small32) {
    const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 3) / 4) * (C / 8);
    hipLaunchKernelGGL(dw3x3_s1_strip2_v8_k<false>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (const bf16_t*)w, (bf16_t*)y, (unsigned)B, (unsigned)H, (unsigned)W, (unsigned)C);
    MMRCA_CHECK_LAUNCH("dwconv3x3_fwd(strip2)");
    return 0;
  }
  if (dw_v8_ok(C, dtype, x, w, y) && stride == 1 && g_dw_strip) {
    const int64_t n = (int64_t)B * H * ((W + 3) / 4) * (C / 8);
    hipLaunchKernelGGL(dw3x3_s1_strip_v8_k<false>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (const bf16_t*)w, (bf16_t*)y, B, H, W, C);
    MMRCA_CHECK_LAUNCH("dwconv3x3_fwd(strip)");
    return 0;
  }
  if (dw_v8_ok(C, dtype, x, w, y)) {
    const int64_t n8 = (int64_t)B * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(dwconv3x3_fwd_v8_k, dim3(blocks_for(n8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w,
                       (bf16_t*)y, B, H, W, C, Ho, Wo, stride);
    MMRCA_CHECK_LAUNCH("dwconv3x3_fwd(v8)");
    return 0;
  }
  const int64_t n = (int64_t)B * Ho * Wo * C;
  MMRCA_DISPATCH_DTYPE(dtype, "dwconv3x3_fwd",
    hipLaunchKernelGGL(dwconv3x3_fwd_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)w, (T*)y,
                       B, H, W, C, Ho, Wo, stride);)
  MMRCA_CHECK_LAUNCH("dwconv3x3_fwd");
  return 0;
}

extern "C" int mmrca_dwconv3x3_bwd_ws(const void* dy, const void* x, const void* w, void* dx, float* dw, int B, int H, int W, int C, int stride,
                                      int dtype, void* ws, int64_t ws_bytes, void* stream);
extern "C" int mmrca_dwconv3x3_bwd(const void* dy, const void* x, const void* w, void* dx, float* dw, int B, int H, int W, int C, int stride,
                                   int dtype, void* stream) {
  return mmrca_dwconv3x3_bwd_ws(dy, x, w, dx, dw, B, H, W, C, stride, dtype, nullptr, 0, stream);
}
/* the same with a scratch buffer for the weight gradient's partial sums (any size; >= 16 MiB lets every shape of the conv backbones
 * use ~3,000 blocks): the stride-1 bf16 kernel then writes per-block sums and a second kernel adds them, instead of fp32 atomics */
extern "C" int mmrca_dwconv3x3_bwd_ws(const void* dy, const void* x, const void* w, void* dx, float* dw, int B, int H, int W, int C, int stride,
                                      int dtype, void* ws, int64_t ws_bytes, void* stream) {
  MMRCA_REQUIRE(dy && x && w && B > 0 && H > 0 && W > 0 && C > 0 && (stride == 1 || stride == 2), "dwconv3x3_bwd: bad arguments");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  const bool small32 = (int64_t)B * H * W * C * 2 < (1ll << 32);
  if (dx && dw_v8_ok(C, dtype, dy, w, dx) && stride == 1 && g_dw_strip && g_dw_strip2 && small32) {
    const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 3) / 4) * (C / 8);
    hipLaunchKernelGGL(dw3x3_s1_strip2_v8_k<true>, dim3(blocks_for(n, 256)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)w,
                       (bf16_t*)dx, (unsigned)B, (unsigned)H, (unsigned)W, (unsigned)C);
    MMRCA_CHECK_LAUNCH("dwconv3x3_bwd(data,strip2)");
  } else if (dx && dw_v8_ok(C, dtype, dy, w, dx) && stride == 1 && g_dw_strip) {
    const int64_t n = (int64_t)B * H * ((W + 3) / 4) * (C / 8);
    hipLaunchKernelGGL(dw3x3_s1_strip_v8_k<true>, dim3(blocks_for(n, 256)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)w,
                       (bf16_t*)dx, B, H, W, C);
    MMRCA_CHECK_LAUNCH("dwconv3x3_bwd(data,strip)");
  } else if (dx && dw_v8_ok(C, dtype, dy, w, dx)) {
    const int64_t n8 = (int64_t)B * H * W * (C / 8);
    hipLaunchKernelGGL(dwconv3x3_bwd_data_v8_k, dim3(blocks_for(n8, 256)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)w, (bf16_t*)dx,
                       B, H, W, C, Ho, Wo, stride);
    MMRCA_CHECK_LAUNCH("dwconv3x3_bwd(data,v8)");
  } else if (dx) {
    const int64_t n = (int64_t)B * H * W * C;
    MMRCA_DISPATCH_DTYPE(dtype, "dwconv3x3_bwd",
      hipLaunchKernelGGL(dwconv3x3_bwd_data_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, st, (const T*)dy, (const T*)w, (T*)dx, B, H, W, C,
                         Ho, Wo, stride);)
    MMRCA_CHECK_LAUNCH("dwconv3x3_bwd(data)");
  }
  if (dw) {
    const int64_t npix = (int64_t)B * Ho * Wo;
    // pixel ranges of the two general kernels below: ~2,048 blocks in all (a small-batch layer -- configs[0]: 4 x 28 x 28 pixels --
    // used to get 6 ranges of 523 pixels, i.e. 131 dependent iterations per thread: 365 us for 0.8 MB), at least 32 pixels each
    const int64_t cslices = (C + 63) / 64;
    int64_t nblk = npix / 32 > 0 ? npix / 32 : 1;
    if (nblk > 2048 / cslices) nblk = 2048 / cslices > 0 ? 2048 / cslices : 1;
    const int64_t per = (npix + nblk - 1) / nblk;
    if (dw_v8_ok(C, dtype, dy, x, dw) && stride == 1 && g_dw_strip && g_dw_strip2 && small32) {
      const int64_t ngrp = (int64_t)B * H * ((W + 3) / 4);
      const int slices = (C / 8 + 63) / 64;                  // 64 channel groups (512 channels) per block
      const int64_t row_bytes = (int64_t)slices * 4608 * 4;  // one workspace row: every block of one pixel range
      const bool use_ws = ws && (((uintptr_t)ws) & 15) == 0 && ws_bytes >= 8 * row_bytes;
      int64_t nb = (use_ws ? g_dw_blocks_ws : g_dw_blocks) / slices;   // blocks in all: ~12 per CU with a workspace, ~3 when every block ends in 4,608 atomics
      if (use_ws && nb > ws_bytes / row_bytes) nb = ws_bytes / row_bytes;
      if (nb > ngrp / 8) nb = ngrp / 8 > 0 ? ngrp / 8 : 1;
      const int64_t gper = (ngrp + nb - 1) / nb;
      const unsigned nranges = (unsigned)((ngrp + gper - 1) / gper);
      // streaming form: lanes = as many pixel lanes as threads fit on the chip at three waves per SIMD, bounded by the workspace
      const int64_t C8 = C / 8;
      int64_t lanes = g_dw_flat ? (int64_t)g_dw_flat_threads / C8 : 0;
      const int64_t ngrp_f = g_dw_flat_rows == 2 ? (int64_t)B * ((H + 1) / 2) * ((W + 3) / 4) : ngrp;
      if (lanes > ngrp_f) lanes = ngrp_f;
      if (ws && lanes * C8 * 72 * 4 > ws_bytes) lanes = ws_bytes / (C8 * 72 * 4);
      if (ws && (((uintptr_t)ws) & 15) == 0 && lanes >= 64) {
        if (g_dw_flat_rows == 2)
          hipLaunchKernelGGL(dwconv3x3_bwd_weight_s1_flat_v8_k<2>, dim3(blocks_for(lanes * C8, 256)), dim3(256), 0, st, (const bf16_t*)dy,
                             (const bf16_t*)x, (unsigned)B, (unsigned)H, (unsigned)W, (unsigned)C, (unsigned)lanes, (float*)ws);
        else
          hipLaunchKernelGGL(dwconv3x3_bwd_weight_s1_flat_v8_k<1>, dim3(blocks_for(lanes * C8, 256)), dim3(256), 0, st, (const bf16_t*)dy,
                             (const bf16_t*)x, (unsigned)B, (unsigned)H, (unsigned)W, (unsigned)C, (unsigned)lanes, (float*)ws);
        hipLaunchKernelGGL(dw_wg_reduce_k, dim3((unsigned)((C * 9 + 63) / 64), 8), dim3(256), 0, st, (const float*)ws, dw, (int)lanes, C * 9,
                           (int64_t)C * 9);
      } else if (use_ws) {
        hipLaunchKernelGGL((dwconv3x3_bwd_weight_s1_strip2_v8_k<1, true>), dim3((unsigned)slices, nranges), dim3(256), 0, st,
                           (const bf16_t*)dy, (const bf16_t*)x, dw, (unsigned)B, (unsigned)H, (unsigned)W, (unsigned)C, (unsigned)gper, (float*)ws);
        hipLaunchKernelGGL(dw_wg_reduce_k, dim3((unsigned)((C * 9 + 63) / 64), 8), dim3(256), 0, st, (const float*)ws, dw, (int)nranges, C * 9,
                           (int64_t)slices * 4608);
      } else {
        hipLaunchKernelGGL((dwconv3x3_bwd_weight_s1_strip2_v8_k<1, false>), dim3((unsigned)slices, nranges), dim3(256), 0, st,
                           (const bf16_t*)dy, (const bf16_t*)x, dw, (unsigned)B, (unsigned)H, (unsigned)W, (unsigned)C, (unsigned)gper, (float*)nullptr);
      }
      MMRCA_CHECK_LAUNCH("dwconv3x3_bwd(weight,strip2)");
      return 0;
    }
    if (dw_v8_ok(C, dtype, dy, x, dw) && stride == 1 && g_dw_strip) {
      const int64_t ngrp = (int64_t)B * H * ((W + 3) / 4);
      int64_t nb = ngrp / 128 > 0 ? ngrp / 128 : 1;
      if (nb > 1024) nb = 1024;
      const int64_t gper = (ngrp + nb - 1) / nb;
      hipLaunchKernelGGL(dwconv3x3_bwd_weight_s1_strip_v8_k, dim3((unsigned)((C + 63) / 64), (unsigned)((ngrp + gper - 1) / gper)), dim3(256), 0, st,
                         (const bf16_t*)dy, (const bf16_t*)x, dw, B, H, W, C, gper);
      MMRCA_CHECK_LAUNCH("dwconv3x3_bwd(weight,strip)");
      return 0;
    }
    if (dw_v8_ok(C, dtype, dy, x, dw)) {
      hipLaunchKernelGGL(dwconv3x3_bwd_weight_v8_k, dim3((unsigned)((C + 63) / 64), (unsigned)nblk), dim3(256), 0, st, (const bf16_t*)dy,
                         (const bf16_t*)x, dw, B, H, W, C, Ho, Wo, stride, per);
      MMRCA_CHECK_LAUNCH("dwconv3x3_bwd(weight,v8)");
      return 0;
    }
    MMRCA_DISPATCH_DTYPE(dtype, "dwconv3x3_bwd",
      hipLaunchKernelGGL(dwconv3x3_bwd_weight_k<T>, dim3((unsigned)((C + 63) / 64), (unsigned)nblk), dim3(256), 0, st, (const T*)dy,
                         (const T*)x, dw, B, H, W, C, Ho, Wo, stride, per);)
    MMRCA_CHECK_LAUNCH("dwconv3x3_bwd(weight)");
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// BatchNorm2d over rows (torch.nn.BatchNorm2d as built by torchvision: eps 1e-3 EfficientNetV2 / 1e-5 ShuffleNetV2, momentum
// 0.1).  Training: batch mean and BIASED variance normalise; running_var is updated with the UNBIASED variance.
// Column reductions: block = 64 channels x 4 row lanes over a slice of rows -> LDS -> one atomic per channel per block.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, bool CENTERED>
__global__ void col_moment_k(const T* __restrict__ x, const float* __restrict__ mean, float* __restrict__ out, int64_t rows, int C,
                             int64_t ld, int64_t rows_per_block, float mscale) {      // CENTERED: mean[] holds the raw column SUMS, mscale = 1 / rows
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float s = 0.f;
  if (c < C) {
    const float m = CENTERED ? mean[c] * mscale : 0.f;
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      const float v = to_f(x[r * ld + c]) - m;
      s += CENTERED ? v * v : v;
    }
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) atomicAdd(out + c, red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

// var_sum -> rstd (in place), running stats
// (mean[] arrives as the raw column sums: the mean is formed here and -- with the same multiplication -- by the centred pass that ran
// before; a separate "sums -> mean" launch between the two passes was one of the nine launches of a two-pass BatchNorm)
__global__ void bn_finish_var_k(float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                                float* __restrict__ running_var, int C, float n, float eps, float momentum, float mscale) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mu = mean[c] * mscale;
  mean[c] = mu;
  const float var = rstd[c] / n;
  rstd[c] = rsqrtf(var + eps);
  if (running_mean && momentum > 0.f) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1.f ? var * n / (n - 1.f) : var);
  }
}
__global__ void bn_eval_stats_k(const float* __restrict__ running_mean, const float* __restrict__ running_var, float* __restrict__ mean,
                                float* __restrict__ rstd, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { mean[c] = running_mean[c]; rstd[c] = rsqrtf(running_var[c] + eps); }
}

// ---- 8 channels per thread (bf16, C % 8 == 0, 16-byte aligned rows): block = 8 channel groups (64 channels) x 32 row lanes;
// the 8 row lanes of a wave are reduced by shuffles, the four waves through LDS, then one atomic per channel.  The partial
// sums are formed in a different order than in the scalar kernels (fp32 column sums over up to 1.8 M rows either way).
typedef __attribute__((ext_vector_type(8))) __bf16 cm_b8;
template <int NV>      // NV accumulators per channel
__device__ __forceinline__ void col_reduce8(float (&acc)[NV][8], float (*red)[8][8 * NV + 1], float* const (&out)[NV], int cblk, int C) {
  const int cg = threadIdx.x & 7, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = acc[k][j];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      if ((threadIdx.x & 63) < 8) red[wv][cg][k * 8 + j] = v;
    }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * NV; e += 256) {
    const int k = e / 64, cl = e % 64, g = cl >> 3, j = cl & 7;
    const int c = cblk * 64 + cl;
    if (c < C) atomicAdd(out[k] + c, (red[0][g][k * 8 + j] + red[1][g][k * 8 + j]) + (red[2][g][k * 8 + j] + red[3][g][k * 8 + j]));
  }
}

template <bool CENTERED>
__global__ void __launch_bounds__(256)
col_moment_v8_k(const bf16_t* __restrict__ x, const float* __restrict__ mean, float* __restrict__ out, int64_t rows, int C, int64_t ld,
                int64_t rows_per_block, float mscale) {
  __shared__ float red[4][8][9];
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cg * 8;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float acc[1][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = 0.f;
  if (c0 < C) {
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = CENTERED ? mean[c0 + j] * mscale : 0.f;
    for (int64_t r = r0 + rl; r < r1; r += 32) {
      const cm_b8 v = *reinterpret_cast<const cm_b8*>(x + r * ld + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = (float)v[j] - m[j]; acc[0][j] += CENTERED ? d * d : d; }
    }
  }
  float* const outs[1] = {out};
  col_reduce8<1>(acc, red, outs, blockIdx.x, C);
}

// Both moments in ONE pass over x (bf16 fast path of mmrca_bn_stats): sums of d and d*d with d = x - shift[c] (fp32 sums).
// Any shift within a few standard deviations of the channel mean keeps var = E[d^2] - E[d]^2 free of cancellation.  Round 2
// used row 0 as the shift -- the top-left pixel of image 0, a zero-padded conv border / letterbox region that can sit hundreds of
// standard deviations from the channel mean, and the subtraction then cancels in fp32 over millions of rows.  The shift is now a
// TRIMMED mean (minimum and maximum dropped) of BN_SHIFT_ROWS rows at hashed positions, one per sixteenth of the tensor (evenly
// spaced rows of a [B*H*W, C] activation are all the same pixel of different images -- e.g. all corners).  Both kernels below
// form it with the same operations in the same order.
#define BN_SHIFT_ROWS 16
__device__ __forceinline__ int64_t bn_shift_row(int k, int64_t rows) {
  const int64_t seg = rows / BN_SHIFT_ROWS;
  if (seg <= 0) return k < rows ? k : rows - 1;
  return (int64_t)k * seg + (int64_t)((2654435761u * (unsigned)(k + 1)) % (unsigned long long)seg);
}
__device__ __forceinline__ float bn_trimmed(float sum, float lo, float hi) { return (sum - lo - hi) * (1.0f / (BN_SHIFT_ROWS - 2)); }
struct BnFinish { int* counters; float* running_mean; float* running_var; float eps, momentum; };
__global__ void __launch_bounds__(256)
col_moment2_v8_k(const bf16_t* __restrict__ x, float* __restrict__ s1, float* __restrict__ s2, int64_t rows, int C, int64_t ld,
                 int64_t rows_per_block, BnFinish fin = BnFinish{nullptr, nullptr, nullptr, 0.f, 0.f}, float* __restrict__ shift_out = nullptr) {
  __shared__ float red[4][8][17];
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cg * 8;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
  if (c0 < C) {
    float m[8], mlo[8], mhi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m[j] = 0.f; mlo[j] = INFINITY; mhi[j] = -INFINITY; }
    for (int k = 0; k < BN_SHIFT_ROWS; ++k) {
      const cm_b8 sv = *reinterpret_cast<const cm_b8*>(x + bn_shift_row(k, rows) * ld + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float v = (float)sv[j]; m[j] += v; mlo[j] = fminf(mlo[j], v); mhi[j] = fmaxf(mhi[j], v); }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = bn_trimmed(m[j], mlo[j], mhi[j]);
    // shift_out (mmrca_bn_moments): the shift this launch subtracted, for a consumer that turns the sums into mean / rstd itself
    // (bn_act_fwd_fin_v8_k: no bn_finish_shifted_k launch); every row lane of every row range holds the same values
    if (shift_out && blockIdx.y == 0 && rl == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) shift_out[c0 + j] = m[j];
    }
    int64_t r = r0 + rl;
    for (; r + 224 < r1; r += 256) {                   // eight rows in flight per thread (one load per iteration ran at 2.6 TB/s)
      cm_b8 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const cm_b8*>(x + (r + 32 * u) * ld + c0);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = (float)v[u][j] - m[j]; acc[0][j] += d; acc[1][j] = fmaf(d, d, acc[1][j]); }
    }
    for (; r < r1; r += 32) {
      const cm_b8 v = *reinterpret_cast<const cm_b8*>(x + r * ld + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = (float)v[j] - m[j]; acc[0][j] += d; acc[1][j] = fmaf(d, d, acc[1][j]); }
    }
  }
  float* const outs[2] = {s1, s2};
  col_reduce8<2>(acc, red, outs, blockIdx.x, C);
  // fin.counters != NULL (round 5): the LAST of the gridDim.y row ranges of this 64-channel block turns the sums into mean / rstd and
  // updates the running statistics itself -- what bn_finish_shifted_k did in a launch of its own (one per BatchNorm layer and step: 146
  // of the ~1,900 launches of an EfficientNetV2-M step at the reference's batch size 16).  The atomics above are device-scope
  // read-modify-writes at L2; the fence orders them before the ticket, and the finishing reads go to L2 as well.
  if (fin.counters) {
    __shared__ int last_blk;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last_blk = atomicAdd(fin.counters + blockIdx.x, 1) == (int)gridDim.y - 1;
    __syncthreads();
    if (last_blk && threadIdx.x < 64) {
      const int c = blockIdx.x * 64 + threadIdx.x;
      if (c < C) {
        float shift = 0.f, slo = INFINITY, shi = -INFINITY;
        for (int k = 0; k < BN_SHIFT_ROWS; ++k) { const float v = (float)x[bn_shift_row(k, rows) * ld + c]; shift += v; slo = fminf(slo, v); shi = fmaxf(shi, v); }
        shift = bn_trimmed(shift, slo, shi);
        const float n = (float)rows;
        const float d1 = __hip_atomic_load(s1 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / n;
        const float q = __hip_atomic_load(s2 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / n;
        const float mu = shift + d1;
        const float var = fmaxf(q - d1 * d1, 0.f);
        s1[c] = mu;
        s2[c] = rsqrtf(var + fin.eps);
        if (fin.running_mean && fin.momentum > 0.f) {
          fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * mu;
          fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (n > 1.f ? var * n / (n - 1.f) : var);
        }
      }
    }
  }
}
// ---- FLAT column reductions (round 4).  The kernels above give a workgroup a 64-channel slice (128 bytes) of many rows; measured
// on the MBConv tensors (tools/bn_bench.sh) they stream at 3.1-3.7 TB/s while the element-wise BatchNorm passes next to them, whose
// waves read 1 KiB of consecutive addresses per instruction, reach 5.8-6.6.  Same access pattern here: the tensor is a flat array of
// 16-byte chunks (8 channels), thread t takes chunks t, t + T, t + 2T, ... with T a multiple of C/8 -- so its channel group never
// changes and its partial sums stay in registers for the whole pass -- eight chunks in flight.  Every thread writes its sums as one
// 64-byte record to a caller workspace ([T][16] floats); bn_flat_reduce_k adds the records of a channel group up (64 groups x 4
// lane subsets per block, one atomic per channel and block).  No atomics in the streaming pass, no shared memory.
__global__ void __launch_bounds__(256)
col_moment2_flat_k(const bf16_t* __restrict__ x, float* __restrict__ ws, int64_t rows, int C8, int64_t T) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int c0 = (int)(t % C8) * 8;
  const int64_t ld = (int64_t)C8 * 8;
  float m[8], mlo[8], mhi[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { m[j] = 0.f; mlo[j] = INFINITY; mhi[j] = -INFINITY; }
  for (int k = 0; k < BN_SHIFT_ROWS; ++k) {            // the shift: same rows, same operations, same order as col_moment2_v8_k / bn_finish_shifted_k
    const cm_b8 sv = *reinterpret_cast<const cm_b8*>(x + bn_shift_row(k, rows) * ld + c0);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float v = (float)sv[j]; m[j] += v; mlo[j] = fminf(mlo[j], v); mhi[j] = fmaxf(mhi[j], v); }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) m[j] = bn_trimmed(m[j], mlo[j], mhi[j]);
  float a0[8], a1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a0[j] = 0.f; a1[j] = 0.f; }
  const int64_t nchunk = rows * C8;
  int64_t q = t;
  for (; q + 7 * T < nchunk; q += 8 * T) {
    cm_b8 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const cm_b8*>(x + (q + u * T) * 8);
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = (float)v[u][j] - m[j]; a0[j] += d; a1[j] = fmaf(d, d, a1[j]); }
  }
  for (; q < nchunk; q += T) {
    const cm_b8 v = *reinterpret_cast<const cm_b8*>(x + q * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = (float)v[j] - m[j]; a0[j] += d; a1[j] = fmaf(d, d, a1[j]); }
  }
  float4* o = reinterpret_cast<float4*>(ws + t * 16);
  o[0] = make_float4(a0[0], a0[1], a0[2], a0[3]); o[1] = make_float4(a0[4], a0[5], a0[6], a0[7]);
  o[2] = make_float4(a1[0], a1[1], a1[2], a1[3]); o[3] = make_float4(a1[4], a1[5], a1[6], a1[7]);
}
__global__ void __launch_bounds__(256)
bn_act_bwd_reduce_flat_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ mean,
                         const float* __restrict__ rstd, const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                         float* __restrict__ ws, int64_t rows, int C8, int act, int64_t T) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int c0 = (int)(t % C8) * 8;
  float m[8], rs[8], g[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { m[j] = mean[c0 + j]; rs[j] = rstd[c0 + j]; g[j] = (float)gamma[c0 + j]; b[j] = (float)beta[c0 + j]; }
  float a0[8], a1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a0[j] = 0.f; a1[j] = 0.f; }
  const int64_t nchunk = rows * C8;
  int64_t q = t;
  for (; q + 7 * T < nchunk; q += 8 * T) {
    cm_b8 xv[8], dv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      xv[u] = *reinterpret_cast<const cm_b8*>(x + (q + u * T) * 8);
      dv[u] = *reinterpret_cast<const cm_b8*>(dy + (q + u * T) * 8);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = ((float)xv[u][j] - m[j]) * rs[j];
        const float du = (float)dv[u][j] * act_grad_f(xh * g[j] + b[j], act);
        a0[j] += du; a1[j] = fmaf(du, xh, a1[j]);
      }
  }
  for (; q < nchunk; q += T) {
    const cm_b8 xv = *reinterpret_cast<const cm_b8*>(x + q * 8), dv = *reinterpret_cast<const cm_b8*>(dy + q * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = ((float)xv[j] - m[j]) * rs[j];
      const float du = (float)dv[j] * act_grad_f(xh * g[j] + b[j], act);
      a0[j] += du; a1[j] = fmaf(du, xh, a1[j]);
    }
  }
  float4* o = reinterpret_cast<float4*>(ws + t * 16);
  o[0] = make_float4(a0[0], a0[1], a0[2], a0[3]); o[1] = make_float4(a0[4], a0[5], a0[6], a0[7]);
  o[2] = make_float4(a1[0], a1[1], a1[2], a1[3]); o[3] = make_float4(a1[4], a1[5], a1[6], a1[7]);
}
// out0[c] += sum over lanes of record[lane * C8 + c / 8][c % 8], out1 likewise from the record's second half.  grid = (C8 / 64, lane ranges)
__global__ void __launch_bounds__(256)
bn_flat_reduce_k(const float* __restrict__ ws, int lanes, int C8, float* __restrict__ out0, float* __restrict__ out1, int lanes_per_block) {
  __shared__ float red[4][64][17];
  const int cl = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int c8 = blockIdx.x * 64 + cl;
  const int l0 = blockIdx.y * lanes_per_block, l1 = l0 + lanes_per_block < lanes ? l0 + lanes_per_block : lanes;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  if (c8 < C8)
    for (int l = l0 + sub; l < l1; l += 4) {
      const float4* p = reinterpret_cast<const float4*>(ws + ((int64_t)l * C8 + c8) * 16);
      const float4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
      acc[0] += v0.x; acc[1] += v0.y; acc[2] += v0.z; acc[3] += v0.w; acc[4] += v1.x; acc[5] += v1.y; acc[6] += v1.z; acc[7] += v1.w;
      acc[8] += v2.x; acc[9] += v2.y; acc[10] += v2.z; acc[11] += v2.w; acc[12] += v3.x; acc[13] += v3.y; acc[14] += v3.z; acc[15] += v3.w;
    }
#pragma unroll
  for (int j = 0; j < 16; ++j) red[sub][cl][j] = acc[j];
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 16; e += 256) {
    const int g = e >> 4, j = e & 15, cc = blockIdx.x * 64 + g;
    if (cc < C8) {
      const float v = (red[0][g][j] + red[1][g][j]) + (red[2][g][j] + red[3][g][j]);
      atomicAdd((j < 8 ? out0 : out1) + cc * 8 + (j & 7), v);
    }
  }
}
// Measured (tools/bn_bench.sh, 354 MB tensor): backward reduce 207 -> 178 + 9 us (3.4 -> 4.0 TB/s), moments 100 -> 94 + 9 us.  Neither gets
// near the 5.8-6.6 TB/s of the element-wise passes, and that is not the access pattern: those passes WRITE a third to a half of their
// bytes.  Every read-only pass here tops out at 3.7-4.3 TB/s, and the reads of the read+write passes run at 3.0-4.2 TB/s too -- the
// read path of this part saturates near 4.2 TB/s, the quoted ~6.3 TB/s "achievable" is a read + write mix.  BatchNorm on batch
// statistics reads 6 tensors and writes 2 per layer: its floor is the 6 reads at ~4.2 TB/s, and the four passes are within ~17 % of it.
// plan of a flat pass: T threads (a multiple of C/8), or 0 when the tensor is too small / not eligible / the workspace is missing
// In situ (configs[2], B = 128) the flat backward reduce measured 607 samples/s against 612 without it: its dy operand was written by the
// launch before and the slice form finds more of it in the caches, and the second launch costs what the faster stream gains on the many
// mid-sized layers.  OFF by default.  The switch is read from the environment ONCE (MMRCA_BN_FLAT=1: both backward-sum and statistics
// reductions eligible; MMRCA_BN_FLAT_MOMENTS=1: the one-pass moments too) -- BatchNorm launches sit in the launch-bound small-batch
// path, a getenv per launch does not belong there -- and mmrca_bn_flat_set() lets the tests flip it at run time.
static int g_bn_flat = ((getenv("MMRCA_BN_FLAT") && atoi(getenv("MMRCA_BN_FLAT")) == 1) ? 1 : 0) |
                       ((getenv("MMRCA_BN_FLAT_MOMENTS") && atoi(getenv("MMRCA_BN_FLAT_MOMENTS")) == 1) ? 2 : 0);
extern "C" int mmrca_bn_flat_set(int mode) { g_bn_flat = mode & 3; return 0; }
static bool bn_flat_on() { return (g_bn_flat & 1) != 0; }
static const int64_t g_bn_flat_threads = getenv("MMRCA_BN_FLAT_THREADS") ? atoll(getenv("MMRCA_BN_FLAT_THREADS")) : 262144;
static int64_t bn_flat_threads(int64_t rows, int C, int64_t ld, int dtype, const void* a, const void* b, const void* ws, int64_t ws_bytes) {
  if (!bn_flat_on() || !ws || dtype != MMRCA_BF16 || C % 8 || ld != C || ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)ws)) & 15)) return 0;
  const int64_t C8 = C / 8, nchunk = rows * C8;
  int64_t T = g_bn_flat_threads;
  if (T > nchunk / 16) T = nchunk / 16;                 // at least 16 chunks per thread
  if (T * 64 > ws_bytes) T = ws_bytes / 64;
  T = T / C8 * C8;
  return T >= 4096 && T / C8 >= 4 ? T : 0;
}
static void bn_flat_reduce(const float* ws, int64_t T, int C8, float* out0, float* out1, hipStream_t st) {
  const int lanes = (int)(T / C8);
  int per = (lanes + 31) / 32;                          // <= 32 lane ranges
  if (per < 16) per = 16;
  hipLaunchKernelGGL(bn_flat_reduce_k, dim3((unsigned)((C8 + 63) / 64), (unsigned)((lanes + per - 1) / per)), dim3(256), 0, st, ws, lanes, C8,
                     out0, out1, per);
}

// (sum d, sum d^2) -> mean, rstd (in place), running stats
__global__ void bn_finish_shifted_k(const bf16_t* __restrict__ x, float* __restrict__ mean, float* __restrict__ rstd,
                                    float* __restrict__ running_mean, float* __restrict__ running_var, int C, float n, float eps,
                                    float momentum, int64_t rows, int64_t ld) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float shift = 0.f, slo = INFINITY, shi = -INFINITY;
  for (int k = 0; k < BN_SHIFT_ROWS; ++k) { const float v = (float)x[bn_shift_row(k, rows) * ld + c]; shift += v; slo = fminf(slo, v); shi = fmaxf(shi, v); }
  shift = bn_trimmed(shift, slo, shi);
  const float d1 = mean[c] / n;
  const float mu = shift + d1;
  const float var = fmaxf(rstd[c] / n - d1 * d1, 0.f);
  mean[c] = mu;
  rstd[c] = rsqrtf(var + eps);
  if (running_mean && momentum > 0.f) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1.f ? var * n / (n - 1.f) : var);
  }
}

__global__ void __launch_bounds__(256)
bn_act_bwd_reduce_v8_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ mean,
                       const float* __restrict__ rstd, const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                       float* __restrict__ sum_du, float* __restrict__ sum_duxh, int64_t rows, int C, int act, int64_t rows_per_block) {
  __shared__ float red[4][8][17];
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cg * 8;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
  if (c0 < C) {
    float m[8], rs[8], g[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m[j] = mean[c0 + j]; rs[j] = rstd[c0 + j]; g[j] = (float)gamma[c0 + j]; b[j] = (float)beta[c0 + j]; }
    int64_t r = r0 + rl;
    for (; r + 224 < r1; r += 256) {                   // eight rows of both operands in flight per thread
      cm_b8 xv[8], dv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        xv[u] = *reinterpret_cast<const cm_b8*>(x + (r + 32 * u) * C + c0);
        dv[u] = *reinterpret_cast<const cm_b8*>(dy + (r + 32 * u) * C + c0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = ((float)xv[u][j] - m[j]) * rs[j];
          const float du = (float)dv[u][j] * act_grad_f(xh * g[j] + b[j], act);
          acc[0][j] += du; acc[1][j] = fmaf(du, xh, acc[1][j]);
        }
    }
    for (; r < r1; r += 32) {
      const cm_b8 xv = *reinterpret_cast<const cm_b8*>(x + r * C + c0), dv = *reinterpret_cast<const cm_b8*>(dy + r * C + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = ((float)xv[j] - m[j]) * rs[j];
        const float du = (float)dv[j] * act_grad_f(xh * g[j] + b[j], act);
        acc[0][j] += du; acc[1][j] = fmaf(du, xh, acc[1][j]);
      }
    }
  }
  float* const outs[2] = {sum_du, sum_duxh};
  col_reduce8<2>(acc, red, outs, blockIdx.x, C);
}

static void col_grid(int64_t rows, int C, dim3* grid, int64_t* per) {
  // ~3,000 blocks in all (12 per CU): every block pays a prologue (the moments' shift = 16 sampled rows) and a reduction tail, so
  // 256-row blocks -- 5,000-10,000 of them on the MBConv tensors -- spent most of their time there (moments at 2.6 TB/s)
  const int64_t slices = (C + 63) / 64;
  int64_t cap = 3072 / slices;
  if (cap < 8) cap = 8;
  if (cap > 2048) cap = 2048;
  int64_t nblk = rows / 256 > 0 ? rows / 256 : 1;
  if (nblk > cap) nblk = cap;
  if (nblk * slices < 512) {      // a small-batch tensor (configs[0]: 3,136 rows x 122 channels = 24 blocks of 64 dependent iterations, 18-23 us
    nblk = rows / 32 > 0 ? rows / 32 : 1;                     // for 0.8 MB): 32-row blocks, still at most ~1,024 of them
    if (nblk * slices > 1024) nblk = 1024 / slices > 0 ? 1024 / slices : 1;
  }
  *per = (rows + nblk - 1) / nblk;
  // x = 64-channel slice (fastest), y = row range: the blocks resident at one time then cover WHOLE rows of a row range.  With the
  // row ranges in x, the ~2,000 resident blocks all read the same 128-byte slice of every row (stride 2C bytes) -- one DRAM burst
  // per page: col_moment2 ran at 2.5 TB/s, the depthwise weight gradient at 1.8.
  *grid = dim3((unsigned)((C + 63) / 64), (unsigned)((rows + *per - 1) / *per));
}

static const bool g_bn_one_pass = !(getenv("MMRCA_BN_ONE_PASS") && atoi(getenv("MMRCA_BN_ONE_PASS")) == 0);
/* mean[C], rstd[C] (fp32) of x[rows, C] (two passes: mean, then centred second moment; the bf16 fast path: one pass of shifted sums); momentum > 0 also updates the running
 * statistics (torch semantics).  train == 0: mean / rstd are derived from the running statistics instead. */
static int bn_stats_impl(const void* x, float* mean, float* rstd, float* running_mean, float* running_var, int64_t rows, int C,
                         int64_t ld, float eps, float momentum, int train, int dtype, void* ws, int64_t ws_bytes, int flags, void* stream,
                         int* counters = nullptr) {
  MMRCA_REQUIRE(mean && rstd && rows > 0 && C > 0 && ld >= C, "bn_stats: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (!train) {
    MMRCA_REQUIRE(running_mean && running_var, "bn_stats: eval mode needs the running statistics");
    hipLaunchKernelGGL(bn_eval_stats_k, dim3((C + 255) / 256), dim3(256), 0, st, running_mean, running_var, mean, rstd, C, eps);
    MMRCA_CHECK_LAUNCH("bn_stats(eval)");
    return 0;
  }
  MMRCA_REQUIRE(x, "bn_stats: null input");
  dim3 grid; int64_t per;
  col_grid(rows, C, &grid, &per);
  if (counters && !(flags & 1)) (void)hipMemsetAsync(counters, 0, sizeof(int) * ((C + 63) / 64), st);
  if (flags & 1) {                      // the caller zeroed mean / rstd (one fill for every layer of the step: conv_engine's arena)
  } else if (rstd == mean + C) {        // adjacent (one [2, C] buffer): one fill instead of two
    (void)hipMemsetAsync(mean, 0, sizeof(float) * 2 * C, st);
  } else {
    (void)hipMemsetAsync(mean, 0, sizeof(float) * C, st);
    (void)hipMemsetAsync(rstd, 0, sizeof(float) * C, st);
  }
  if (dtype == MMRCA_BF16 && C % 8 == 0 && ld % 8 == 0 && (((uintptr_t)x) & 15) == 0) {
    if (g_bn_one_pass) {
      // flat form of the moments: measured 94 + 9 us against 100 us for the slice form on a 354 MB tensor -- both sit at the ~4 TB/s that a
      // READ-ONLY pass reaches on this part (see bn_flat_threads) and the flat one pays a second launch: off unless MMRCA_BN_FLAT_MOMENTS=1
      const bool flat_moments = (g_bn_flat & 2) != 0;
      const int64_t T = flat_moments ? bn_flat_threads(rows, C, ld, dtype, x, x, ws, ws_bytes) : 0;
      if (T) {
        hipLaunchKernelGGL(col_moment2_flat_k, dim3(blocks_for(T, 256)), dim3(256), 0, st, (const bf16_t*)x, (float*)ws, rows, C / 8, T);
        bn_flat_reduce((const float*)ws, T, C / 8, mean, rstd, st);
      } else if (counters) {        // moments + finish in one launch (caller-zeroed tickets, one per 64 channels)
        hipLaunchKernelGGL(col_moment2_v8_k, grid, dim3(256), 0, st, (const bf16_t*)x, mean, rstd, rows, C, ld, per,
                           BnFinish{counters, running_mean, running_var, eps, momentum});
        MMRCA_CHECK_LAUNCH("bn_stats(one pass, fused finish)");
        return 0;
      } else
      hipLaunchKernelGGL(col_moment2_v8_k, grid, dim3(256), 0, st, (const bf16_t*)x, mean, rstd, rows, C, ld, per, BnFinish{nullptr, nullptr, nullptr, 0.f, 0.f});
      hipLaunchKernelGGL(bn_finish_shifted_k, dim3((C + 255) / 256), dim3(256), 0, st, (const bf16_t*)x, mean, rstd, running_mean, running_var, C,
                         (float)rows, eps, momentum, rows, ld);
      MMRCA_CHECK_LAUNCH("bn_stats(one pass)");
      return 0;
    }
    hipLaunchKernelGGL((col_moment_v8_k<false>), grid, dim3(256), 0, st, (const bf16_t*)x, (const float*)nullptr, mean, rows, C, ld, per, 1.0f);
    hipLaunchKernelGGL((col_moment_v8_k<true>), grid, dim3(256), 0, st, (const bf16_t*)x, (const float*)mean, rstd, rows, C, ld, per, 1.0f / (float)rows);
  } else
  MMRCA_DISPATCH_DTYPE(dtype, "bn_stats",
    hipLaunchKernelGGL((col_moment_k<T, false>), grid, dim3(256), 0, st, (const T*)x, (const float*)nullptr, mean, rows, C, ld, per, 1.0f);
    hipLaunchKernelGGL((col_moment_k<T, true>), grid, dim3(256), 0, st, (const T*)x, (const float*)mean, rstd, rows, C, ld, per, 1.0f / (float)rows);)
  hipLaunchKernelGGL(bn_finish_var_k, dim3((C + 255) / 256), dim3(256), 0, st, mean, rstd, running_mean, running_var, C, (float)rows, eps, momentum,
                     1.0f / (float)rows);
  MMRCA_CHECK_LAUNCH("bn_stats");
  return 0;
}
extern "C" int mmrca_bn_stats(const void* x, float* mean, float* rstd, float* running_mean, float* running_var, int64_t rows, int C,
                              int64_t ld, float eps, float momentum, int train, int dtype, void* stream) {
  return bn_stats_impl(x, mean, rstd, running_mean, running_var, rows, C, ld, eps, momentum, train, dtype, nullptr, 0, 0, stream);
}
/* train-mode statistics with the finish step inside the reduction launch: `tickets` = ceil(C / 64) int32 words (zeroed here unless
 * flags bit 0 says the caller cleared them together with mean / rstd, as in mmrca_bn_stats_ws); bf16, C % 8 == 0 -- other inputs run
 * the two-launch form */
extern "C" int mmrca_bn_stats_fused(const void* x, float* mean, float* rstd, float* running_mean, float* running_var, int64_t rows, int C,
                                    int64_t ld, float eps, float momentum, int dtype, int* tickets, int flags, void* stream) {
  MMRCA_REQUIRE(tickets, "bn_stats_fused: null tickets");
  return bn_stats_impl(x, mean, rstd, running_mean, running_var, rows, C, ld, eps, momentum, 1, dtype, nullptr, 0, flags, stream, tickets);
}
/* the same with a workspace (fp32, 16-byte aligned, >= 256 KiB; 16 MiB serves every size): large bf16 tensors take the flat streaming pass */
extern "C" int mmrca_bn_stats_ws(const void* x, float* mean, float* rstd, float* running_mean, float* running_var, int64_t rows, int C,
                                 int64_t ld, float eps, float momentum, int train, int dtype, void* ws, int64_t ws_bytes, int flags,
                                 void* stream) {
  return bn_stats_impl(x, mean, rstd, running_mean, running_var, rows, C, ld, eps, momentum, train, dtype, ws, ws_bytes, flags, stream);
}

template <typename T>
__global__ void bn_act_fwd_k(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                             const T* __restrict__ gamma, const T* __restrict__ beta, T* __restrict__ y, int64_t n, int C, int act) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int c = (int)(idx % C);
  const float u = (to_f(x[idx]) - mean[c]) * rstd[c] * to_f(gamma[c]) + to_f(beta[c]);
  y[idx] = from_f<T>(act_f(u, act));
}

/* y = act(gamma * (x - mean) * rstd + beta) on contiguous [rows, C] */
// ---- 8 channels per thread (bf16, C % 8 == 0): the element-wise passes of BatchNorm with 16-byte accesses; same arithmetic
typedef __attribute__((ext_vector_type(8))) __bf16 bn_b8;
__device__ __forceinline__ void bn_load8(const float* __restrict__ p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
// BatchNorm passes, 8 channels x BN_R rows per thread: thread t -> channel group t % C8, rows BN_R * (t / C8) ...  The per-channel
// constants (mean, rstd, gamma, beta, and the backward's two sums: up to 160 bytes) are loaded once per BN_R row vectors instead of
// once per 16-byte vector -- with one vector per thread they were 3-5x the data loads and held the kernels at 4.2-4.7 TB/s --
// and the BN_R data loads are issued together.
#define BN_R 4
__global__ void __launch_bounds__(256)
bn_act_fwd_v8_k(const bf16_t* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd, const bf16_t* __restrict__ gamma,
                const bf16_t* __restrict__ beta, bf16_t* __restrict__ y, int64_t rows, int C8, int act) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r0 = t / C8 * BN_R;
  if (r0 >= rows) return;
  const int c0 = (int)(t % C8) * 8;
  const int64_t i0 = (r0 * C8 + c0 / 8) * 8, rs8 = (int64_t)C8 * 8;
  bn_b8 xv[BN_R];
#pragma unroll
  for (int k = 0; k < BN_R; ++k) xv[k] = *reinterpret_cast<const bn_b8*>(x + i0 + (r0 + k < rows ? k : 0) * rs8);
  float m[8], r[8];
  bn_load8(mean + c0, m); bn_load8(rstd + c0, r);
  const bn_b8 g = *reinterpret_cast<const bn_b8*>(gamma + c0), b = *reinterpret_cast<const bn_b8*>(beta + c0);
#pragma unroll
  for (int k = 0; k < BN_R; ++k) {
    if (r0 + k >= rows) break;
    bn_b8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)act_f(((float)xv[k][j] - m[j]) * r[j] * (float)g[j] + (float)b[j], act);
    *reinterpret_cast<bn_b8*>(y + i0 + k * rs8) = o;
  }
}
// the same followed by the block's residual connection: out = res + rowscale[sample] * act(bn(x))  (rowscale NULL = 1): the last
// unit of an MBConv / FusedMBConv block writes the block output directly -- no y tensor, no separate residual_add pass
__global__ void __launch_bounds__(256)
bn_act_fwd_res_v8_k(const bf16_t* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd, const bf16_t* __restrict__ gamma,
                    const bf16_t* __restrict__ beta, const bf16_t* __restrict__ res, const float* __restrict__ rowscale, bf16_t* __restrict__ out,
                    int64_t rows, int C8, int act, int64_t rows_per_sample) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r0 = t / C8 * BN_R;
  if (r0 >= rows) return;
  const int c0 = (int)(t % C8) * 8;
  const int64_t i0 = (r0 * C8 + c0 / 8) * 8, rs8 = (int64_t)C8 * 8;
  bn_b8 xv[BN_R], rv[BN_R];
  float sc[BN_R];
#pragma unroll
  for (int k = 0; k < BN_R; ++k) {
    const int64_t kk = r0 + k < rows ? k : 0;
    xv[k] = *reinterpret_cast<const bn_b8*>(x + i0 + kk * rs8);
    rv[k] = *reinterpret_cast<const bn_b8*>(res + i0 + kk * rs8);
    sc[k] = rowscale ? rowscale[(r0 + kk) / rows_per_sample] : 1.f;
  }
  float m[8], r[8];
  bn_load8(mean + c0, m); bn_load8(rstd + c0, r);
  const bn_b8 g = *reinterpret_cast<const bn_b8*>(gamma + c0), b = *reinterpret_cast<const bn_b8*>(beta + c0);
#pragma unroll
  for (int k = 0; k < BN_R; ++k) {
    if (r0 + k >= rows) break;
    bn_b8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      o[j] = (bf16_t)((float)rv[k][j] + sc[k] * act_f(((float)xv[k][j] - m[j]) * r[j] * (float)g[j] + (float)b[j], act));
    *reinterpret_cast<bn_b8*>(out + i0 + k * rs8) = o;
  }
}
// The forward apply with the FINISH step inside (round 6): the moments launch left (sum d, sum d^2) in s1 / s2 and its shift in `shift`
// (mmrca_bn_moments); every thread turns the sums of its eight channels into mean / rstd itself -- the arithmetic of bn_finish_shifted_k,
// operation for operation -- and the threads of the first row group also store them (the backward reads them) and update the running
// statistics.  One launch less per BatchNorm layer and step (146 of the ~2,000 launches of an EfficientNetV2-M step).  RES: the block's
// residual connection as in bn_act_fwd_res_v8_k.
template <bool RES>
__global__ void __launch_bounds__(256)
bn_act_fwd_fin_v8_k(const bf16_t* __restrict__ x, const float* __restrict__ s1, const float* __restrict__ s2, const float* __restrict__ shift,
                    const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta, const bf16_t* __restrict__ res,
                    const float* __restrict__ rowscale, bf16_t* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                    float* __restrict__ running_mean, float* __restrict__ running_var, int64_t rows, int C8, int act, int64_t rows_per_sample,
                    float n, float eps, float momentum) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r0 = t / C8 * BN_R;
  if (r0 >= rows) return;
  const int c0 = (int)(t % C8) * 8;
  const int64_t i0 = (r0 * C8 + c0 / 8) * 8, rs8 = (int64_t)C8 * 8;
  bn_b8 xv[BN_R];
  [[maybe_unused]] bn_b8 rv[BN_R];
  [[maybe_unused]] float sc[BN_R];
#pragma unroll
  for (int k = 0; k < BN_R; ++k) {
    const int64_t kk = r0 + k < rows ? k : 0;
    xv[k] = *reinterpret_cast<const bn_b8*>(x + i0 + kk * rs8);
    if constexpr (RES) {
      rv[k] = *reinterpret_cast<const bn_b8*>(res + i0 + kk * rs8);
      sc[k] = rowscale ? rowscale[(r0 + kk) / rows_per_sample] : 1.f;
    }
  }
  float a1[8], a2[8], sh[8], m[8], r[8];
  bn_load8(s1 + c0, a1); bn_load8(s2 + c0, a2); bn_load8(shift + c0, sh);
  const bn_b8 g = *reinterpret_cast<const bn_b8*>(gamma + c0), b = *reinterpret_cast<const bn_b8*>(beta + c0);
  float var[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float d1 = a1[j] / n;
    m[j] = sh[j] + d1;
    var[j] = fmaxf(a2[j] / n - d1 * d1, 0.f);
    r[j] = rsqrtf(var[j] + eps);
  }
  if (r0 == 0) {                       // (one thread per channel group: t < C8)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mean_out[c0 + j] = m[j];
      rstd_out[c0 + j] = r[j];
      if (running_mean && momentum > 0.f) {
        running_mean[c0 + j] = (1.f - momentum) * running_mean[c0 + j] + momentum * m[j];
        running_var[c0 + j] = (1.f - momentum) * running_var[c0 + j] + momentum * (n > 1.f ? var[j] * n / (n - 1.f) : var[j]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < BN_R; ++k) {
    if (r0 + k >= rows) break;
    bn_b8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = act_f(((float)xv[k][j] - m[j]) * r[j] * (float)g[j] + (float)b[j], act);
      if constexpr (RES) o[j] = (bf16_t)((float)rv[k][j] + sc[k] * v);
      else o[j] = (bf16_t)v;
    }
    *reinterpret_cast<bn_b8*>(y + i0 + k * rs8) = o;
  }
}
__global__ void __launch_bounds__(256)
bn_act_bwd_apply_v8_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                      const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta, const float* __restrict__ sum_du,
                      const float* __restrict__ sum_duxh, bf16_t* __restrict__ dx, int64_t rows, int C8, int act, float inv_rows, int train,
                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
  if (dgamma && blockIdx.x == 0)
    for (int c = threadIdx.x; c < C8 * 8; c += blockDim.x) { dgamma[c] += sum_duxh[c]; dbeta[c] += sum_du[c]; }
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r0 = t / C8 * BN_R;
  if (r0 >= rows) return;
  const int c0 = (int)(t % C8) * 8;
  const int64_t i0 = (r0 * C8 + c0 / 8) * 8, rs8 = (int64_t)C8 * 8;
  bn_b8 xv[BN_R], dv[BN_R];
#pragma unroll
  for (int k = 0; k < BN_R; ++k) {
    const int64_t kk = r0 + k < rows ? k : 0;
    xv[k] = *reinterpret_cast<const bn_b8*>(x + i0 + kk * rs8);
    dv[k] = *reinterpret_cast<const bn_b8*>(dy + i0 + kk * rs8);
  }
  float m[8], r[8], sd[8], sx[8];
  bn_load8(mean + c0, m); bn_load8(rstd + c0, r); bn_load8(sum_du + c0, sd); bn_load8(sum_duxh + c0, sx);
  const bn_b8 g = *reinterpret_cast<const bn_b8*>(gamma + c0), b = *reinterpret_cast<const bn_b8*>(beta + c0);
#pragma unroll
  for (int k = 0; k < BN_R; ++k) {
    if (r0 + k >= rows) break;
    bn_b8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gj = (float)g[j];
      const float xh = ((float)xv[k][j] - m[j]) * r[j];
      float du = (float)dv[k][j] * act_grad_f(xh * gj + (float)b[j], act);
      if (train) du -= (sd[j] + xh * sx[j]) * inv_rows;
      o[j] = (bf16_t)(gj * r[j] * du);
    }
    *reinterpret_cast<bn_b8*>(dx + i0 + k * rs8) = o;
  }
}
static bool bn_v8_ok(int C, int dtype, const void* a, const void* b, const void* c, const void* d) {
  return dtype == MMRCA_BF16 && C % 8 == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}

extern "C" int mmrca_bn_act_fwd(const void* x, const float* mean, const float* rstd, const void* gamma, const void* beta, void* y,
                                int64_t rows, int C, int act, int dtype, void* stream) {
  MMRCA_REQUIRE(x && mean && rstd && gamma && beta && y && rows > 0 && C > 0 && act >= 0 && act <= 3, "bn_act_fwd: bad arguments");
  const int64_t n = rows * C;
  if (bn_v8_ok(C, dtype, x, y, gamma, beta) && ((((uintptr_t)mean) | ((uintptr_t)rstd)) & 15) == 0) {
    hipLaunchKernelGGL(bn_act_fwd_v8_k, dim3(blocks_for((rows + BN_R - 1) / BN_R * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, mean, rstd, (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, rows, C / 8, act);
    MMRCA_CHECK_LAUNCH("bn_act_fwd(v8)");
    return 0;
  }
  MMRCA_DISPATCH_DTYPE(dtype, "bn_act_fwd",
    hipLaunchKernelGGL(bn_act_fwd_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, mean, rstd, (const T*)gamma,
                       (const T*)beta, (T*)y, n, C, act);)
  MMRCA_CHECK_LAUNCH("bn_act_fwd");
  return 0;
}

/* out = res + rowscale[row / rows_per_sample] * act(bn(x)) (rowscale may be NULL): mmrca_bn_act_fwd followed by mmrca_residual_add in
 * one pass (bf16, C % 8 == 0, 16-byte aligned operands only: other cases return -3 and the caller takes the two calls) */
extern "C" int mmrca_bn_act_fwd_res(const void* x, const float* mean, const float* rstd, const void* gamma, const void* beta, const void* res,
                                    const float* rowscale, void* out, int64_t rows, int C, int act, int64_t rows_per_sample, int dtype,
                                    void* stream) {
  MMRCA_REQUIRE(x && mean && rstd && gamma && beta && res && out && rows > 0 && C > 0 && act >= 0 && act <= 3 && rows_per_sample > 0,
                "bn_act_fwd_res: bad arguments");
  if (!(bn_v8_ok(C, dtype, x, out, gamma, beta) && ((((uintptr_t)mean) | ((uintptr_t)rstd) | ((uintptr_t)res)) & 15) == 0))
    return mmrca_fail(-3, "bn_act_fwd_res: only the bf16 / C %% 8 == 0 / 16-byte aligned case is built");
    hipLaunchKernelGGL(bn_act_fwd_res_v8_k, dim3(blocks_for((rows + BN_R - 1) / BN_R * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, mean, rstd, (const bf16_t*)gamma, (const bf16_t*)beta, (const bf16_t*)res, rowscale, (bf16_t*)out, rows,
                     C / 8, act, rows_per_sample);
  MMRCA_CHECK_LAUNCH("bn_act_fwd_res");
  return 0;
}

/* BatchNorm forward in TWO launches instead of three (bf16, C % 8 == 0, train mode).  mmrca_bn_moments: the shifted one-pass sums of
 * x[rows, C] into s1 / s2 (fp32 [C], += : the caller zeroes them) and the shift it used into `shift` (fp32 [C], written); then
 * mmrca_bn_act_fwd_fin: y = act(bn(x)) (res != NULL: y = res + rowscale[row / rows_per_sample] * act(bn(x))) with the finish step inside --
 * mean / rstd from (s1, s2, shift) per thread, stored to mean_out / rstd_out (distinct from s1 / s2) for the backward, running statistics
 * updated (momentum > 0) -- i.e. what mmrca_bn_stats + mmrca_bn_act_fwd(_res) compute, without the bn_finish launch between them.
 * Other dtypes / channel counts return -3 and the caller takes the three launches. */
extern "C" int mmrca_bn_moments(const void* x, float* s1, float* s2, float* shift, int64_t rows, int C, int64_t ld, int dtype, void* stream) {
  MMRCA_REQUIRE(x && s1 && s2 && shift && rows > 0 && C > 0 && ld >= C, "bn_moments: bad arguments");
  if (!(dtype == MMRCA_BF16 && C % 8 == 0 && ld % 8 == 0 && (((uintptr_t)x) & 15) == 0))
    return mmrca_fail(-3, "bn_moments: only the bf16 / C %% 8 == 0 / 16-byte aligned case is built");
  dim3 grid; int64_t per;
  col_grid(rows, C, &grid, &per);
  hipLaunchKernelGGL(col_moment2_v8_k, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, s1, s2, rows, C, ld, per,
                     BnFinish{nullptr, nullptr, nullptr, 0.f, 0.f}, shift);
  MMRCA_CHECK_LAUNCH("bn_moments");
  return 0;
}
extern "C" int mmrca_bn_act_fwd_fin(const void* x, const float* s1, const float* s2, const float* shift, const void* gamma, const void* beta,
                                    const void* res, const float* rowscale, void* y, float* mean_out, float* rstd_out, float* running_mean,
                                    float* running_var, int64_t rows, int C, int act, int64_t rows_per_sample, float eps, float momentum,
                                    int dtype, void* stream) {
  MMRCA_REQUIRE(x && s1 && s2 && shift && gamma && beta && y && mean_out && rstd_out && rows > 0 && C > 0 && act >= 0 && act <= 3 &&
                rows_per_sample > 0 && mean_out != s1 && mean_out != s2 && rstd_out != s1 && rstd_out != s2, "bn_act_fwd_fin: bad arguments");
  if (!(bn_v8_ok(C, dtype, x, y, gamma, beta) &&
        ((((uintptr_t)s1) | ((uintptr_t)s2) | ((uintptr_t)shift) | ((uintptr_t)res)) & 15) == 0))
    return mmrca_fail(-3, "bn_act_fwd_fin: only the bf16 / C %% 8 == 0 / 16-byte aligned case is built");
  const dim3 grid(blocks_for((rows + BN_R - 1) / BN_R * (C / 8), 256));
  if (res)
    hipLaunchKernelGGL(bn_act_fwd_fin_v8_k<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, s1, s2, shift, (const bf16_t*)gamma,
                       (const bf16_t*)beta, (const bf16_t*)res, rowscale, (bf16_t*)y, mean_out, rstd_out, running_mean, running_var, rows, C / 8,
                       act, rows_per_sample, (float)rows, eps, momentum);
  else
    hipLaunchKernelGGL(bn_act_fwd_fin_v8_k<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, s1, s2, shift, (const bf16_t*)gamma,
                       (const bf16_t*)beta, (const bf16_t*)nullptr, (const float*)nullptr, (bf16_t*)y, mean_out, rstd_out, running_mean,
                       running_var, rows, C / 8, act, rows_per_sample, (float)rows, eps, momentum);
  MMRCA_CHECK_LAUNCH("bn_act_fwd_fin");
  return 0;
}

// backward, pass 1: du = dy * act'(u); sums[0][c] += du, sums[1][c] += du * xhat
template <typename T>
__global__ void bn_act_bwd_reduce_k(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const T* __restrict__ gamma, const T* __restrict__ beta,
                                    float* __restrict__ sum_du, float* __restrict__ sum_duxh, int64_t rows, int C, int act,
                                    int64_t rows_per_block) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    const float m = mean[c], rs = rstd[c], g = to_f(gamma[c]), b = to_f(beta[c]);
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      const float xh = (to_f(x[r * C + c]) - m) * rs;
      const float du = to_f(dy[r * C + c]) * act_grad_f(xh * g + b, act);
      s0 += du; s1 = fmaf(du, xh, s1);
    }
  }
  red[0][rl][cl] = s0; red[1][rl][cl] = s1;
  __syncthreads();
  if (rl == 0 && c < C) {
    atomicAdd(sum_du + c, red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl]);
    atomicAdd(sum_duxh + c, red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl]);
  }
}
// pass 2: dx = gamma * rstd * (du - [train] (sum_du + xhat * sum_duxh) / n)
template <typename T>
__global__ void bn_act_bwd_apply_k(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                   const float* __restrict__ rstd, const T* __restrict__ gamma, const T* __restrict__ beta,
                                   const float* __restrict__ sum_du, const float* __restrict__ sum_duxh, T* __restrict__ dx, int64_t n,
                                   int C, int act, float inv_rows, int train, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  // (the parameter gradients ride on block 0: the sums are complete before this launch starts -- it was a launch of its own)
  if (dgamma && blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += blockDim.x) { dgamma[c] += sum_duxh[c]; dbeta[c] += sum_du[c]; }
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int c = (int)(idx % C);
  const float g = to_f(gamma[c]), rs = rstd[c];
  const float xh = (to_f(x[idx]) - mean[c]) * rs;
  float du = to_f(dy[idx]) * act_grad_f(xh * g + to_f(beta[c]), act);
  if (train) du -= (sum_du[c] + xh * sum_duxh[c]) * inv_rows;
  dx[idx] = from_f<T>(g * rs * du);
}
__global__ void bn_param_grads_k(const float* __restrict__ sum_du, const float* __restrict__ sum_duxh, float* __restrict__ dgamma,
                                 float* __restrict__ dbeta, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { dgamma[c] += sum_duxh[c]; dbeta[c] += sum_du[c]; }
}

/* Backward of y = act(BN(x)): dx (may be NULL), dgamma / dbeta (fp32, +=, may be NULL).  scratch: fp32 [2*C] workspace.
 * train != 0: batch statistics took part in the forward (the usual three-term input gradient); 0: statistics were constants. */
static int bn_act_bwd_impl(const void* dy, const void* x, const float* mean, const float* rstd, const void* gamma, const void* beta,
                           void* dx, float* dgamma, float* dbeta, float* scratch, int64_t rows, int C, int act, int train, int dtype,
                           void* stream, bool sums_ready, void* ws = nullptr, int64_t ws_bytes = 0, int flags = 0) {
  MMRCA_REQUIRE(dy && x && mean && rstd && gamma && beta && scratch && rows > 0 && C > 0 && act >= 0 && act <= 3, "bn_act_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid; int64_t per;
  col_grid(rows, C, &grid, &per);
  if (!sums_ready && !(flags & 1)) (void)hipMemsetAsync(scratch, 0, sizeof(float) * 2 * C, st);      // (flags & 1: the caller zeroed the scratch)
  const int64_t n = rows * C;
  MMRCA_DISPATCH_DTYPE(dtype, "bn_act_bwd",
    const int64_t Tf = sums_ready ? 0 : bn_flat_threads(rows, C, C, dtype, dy, x, ws, ws_bytes);
    if (sums_ready) {}
    else if (Tf) {
      hipLaunchKernelGGL(bn_act_bwd_reduce_flat_k, dim3(blocks_for(Tf, 256)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, mean, rstd,
                         (const bf16_t*)gamma, (const bf16_t*)beta, (float*)ws, rows, C / 8, act, Tf);
      bn_flat_reduce((const float*)ws, Tf, C / 8, scratch, scratch + C, st);
    }
    else if (sizeof(T) == 2 && C % 8 == 0 && ((((uintptr_t)dy) | ((uintptr_t)x)) & 15) == 0)
      hipLaunchKernelGGL(bn_act_bwd_reduce_v8_k, grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, mean, rstd, (const bf16_t*)gamma,
                         (const bf16_t*)beta, scratch, scratch + C, rows, C, act, per);
    else
    hipLaunchKernelGGL(bn_act_bwd_reduce_k<T>, grid, dim3(256), 0, st, (const T*)dy, (const T*)x, mean, rstd, (const T*)gamma, (const T*)beta,
                       scratch, scratch + C, rows, C, act, per);
    const bool v8 = dx && bn_v8_ok(C, dtype, dy, x, dx, gamma) && ((((uintptr_t)mean) | ((uintptr_t)rstd) | ((uintptr_t)scratch) | ((uintptr_t)beta)) & 15) == 0;
    if (v8) hipLaunchKernelGGL(bn_act_bwd_apply_v8_k, dim3(blocks_for((rows + BN_R - 1) / BN_R * (C / 8), 256)), dim3(256), 0, st, (const bf16_t*)dy,
                               (const bf16_t*)x, mean, rstd, (const bf16_t*)gamma, (const bf16_t*)beta, (const float*)scratch,
                               (const float*)(scratch + C), (bf16_t*)dx, rows, C / 8, act, 1.0f / (float)rows, train, dgamma && dbeta ? dgamma : nullptr, dbeta);
    else if (dx) hipLaunchKernelGGL(bn_act_bwd_apply_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, st, (const T*)dy, (const T*)x, mean, rstd,
                               (const T*)gamma, (const T*)beta, (const float*)scratch, (const float*)(scratch + C), (T*)dx, n, C, act,
                               1.0f / (float)rows, train, dgamma && dbeta ? dgamma : nullptr, dbeta);)
  if (dgamma && dbeta && !dx) hipLaunchKernelGGL(bn_param_grads_k, dim3((C + 255) / 256), dim3(256), 0, st, scratch, scratch + C, dgamma, dbeta, C);
  MMRCA_CHECK_LAUNCH("bn_act_bwd");
  return 0;
}
extern "C" int mmrca_bn_act_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const void* gamma, const void* beta,
                                void* dx, float* dgamma, float* dbeta, float* scratch, int64_t rows, int C, int act, int train, int dtype,
                                void* stream) {
  return bn_act_bwd_impl(dy, x, mean, rstd, gamma, beta, dx, dgamma, dbeta, scratch, rows, C, act, train, dtype, stream, false);
}
/* the same with a workspace (as mmrca_bn_stats_ws): the reduce pass of large bf16 tensors runs in the flat streaming form */
extern "C" int mmrca_bn_act_bwd_ws(const void* dy, const void* x, const float* mean, const float* rstd, const void* gamma, const void* beta,
                                   void* dx, float* dgamma, float* dbeta, float* scratch, int64_t rows, int C, int act, int train, int dtype,
                                   void* ws, int64_t ws_bytes, int flags, void* stream) {
  return bn_act_bwd_impl(dy, x, mean, rstd, gamma, beta, dx, dgamma, dbeta, scratch, rows, C, act, train, dtype, stream, false, ws, ws_bytes, flags);
}
/* the same when sums[0..C) = sum du and sums[C..2C) = sum du * xhat are already there (mmrca_se_dx left them): no reduce pass */
extern "C" int mmrca_bn_act_bwd_sums(const void* dy, const void* x, const float* mean, const float* rstd, const void* gamma, const void* beta,
                                     void* dx, float* dgamma, float* dbeta, float* sums, int64_t rows, int C, int act, int train, int dtype,
                                     void* stream) {
  return bn_act_bwd_impl(dy, x, mean, rstd, gamma, beta, dx, dgamma, dbeta, sums, rows, C, act, train, dtype, stream, true);
}

// ---------------------------------------------------------------------------------------------------------------------
// per-sample pooling over the H*W rows of a sample, squeeze-excitation scaling, residual with stochastic depth
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void rowpool_mean_k(const T* __restrict__ x, T* __restrict__ out, int HW, int C) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl, b = blockIdx.x;
  float s = 0.f;
  if (c < C)
    for (int p = rl; p < HW; p += 4) s += to_f(x[((int64_t)b * HW + p) * C + c]);
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) out[(int64_t)b * C + c] = from_f<T>((red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]) / (float)HW);
}
// ---- 8 channels per thread (bf16, C % 8 == 0): block = (sample, 64 channels) = 8 channel groups x 32 row lanes, 16-byte accesses;
// the row lanes of a wave meet through shuffles, the four waves through LDS.  Shared by the pooling and by the backward of the
// squeeze-excitation scale (which also writes dx = dy * s on the way).
typedef __attribute__((ext_vector_type(8))) __bf16 pl_b8;
__device__ __forceinline__ void pool_reduce8(float (&acc)[8], float (*red)[8][9], bf16_t* __restrict__ out_row, int cblk, int C, float scale) {
  const int cg = threadIdx.x & 7, wv = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = acc[j];
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if ((threadIdx.x & 63) < 8) red[wv][cg][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int g = threadIdx.x >> 3, j = threadIdx.x & 7, c = cblk * 64 + threadIdx.x;
    if (c < C) out_row[c] = (bf16_t)(((red[0][g][j] + red[1][g][j]) + (red[2][g][j] + red[3][g][j])) * scale);
  }
}
__global__ void __launch_bounds__(256)
rowpool_mean_v8_k(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int HW, int C) {
  __shared__ float red[4][8][9];
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c0 = blockIdx.y * 64 + cg * 8, b = blockIdx.x;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (c0 < C)
    for (int p = rl; p < HW; p += 32) {
      const pl_b8 v = *reinterpret_cast<const pl_b8*>(x + ((int64_t)b * HW + p) * C + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
    }
  pool_reduce8(acc, red, out + (int64_t)b * C, blockIdx.y, C, 1.0f / (float)HW);
}
__global__ void __launch_bounds__(256)
se_scale_bwd_v8_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ s, bf16_t* __restrict__ dx,
                  bf16_t* __restrict__ ds, int HW, int C) {
  __shared__ float red[4][8][9];
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c0 = blockIdx.y * 64 + cg * 8, b = blockIdx.x;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (c0 < C) {
    const pl_b8 s8 = *reinterpret_cast<const pl_b8*>(s + (int64_t)b * C + c0);
    int p = rl;
    for (; p + 96 < HW; p += 128) {                      // four rows of both operands in flight per thread
      const int64_t i0 = ((int64_t)b * HW + p) * C + c0;
      pl_b8 g[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { g[u] = *reinterpret_cast<const pl_b8*>(dy + i0 + (int64_t)32 * u * C); xv[u] = *reinterpret_cast<const pl_b8*>(x + i0 + (int64_t)32 * u * C); }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        pl_b8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[j] = fmaf((float)g[u][j], (float)xv[u][j], acc[j]); o[j] = (bf16_t)((float)g[u][j] * (float)s8[j]); }
        if (dx) *reinterpret_cast<pl_b8*>(dx + i0 + (int64_t)32 * u * C) = o;
      }
    }
    for (; p < HW; p += 32) {
      const int64_t i = ((int64_t)b * HW + p) * C + c0;
      const pl_b8 g = *reinterpret_cast<const pl_b8*>(dy + i), xv = *reinterpret_cast<const pl_b8*>(x + i);
      pl_b8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) { acc[j] = fmaf((float)g[j], (float)xv[j], acc[j]); o[j] = (bf16_t)((float)g[j] * (float)s8[j]); }
      if (dx) *reinterpret_cast<pl_b8*>(dx + i) = o;      // dx == NULL: only ds (mmrca_se_dx writes dx with the pooled gradient added)
    }
  }
  pool_reduce8(acc, red, ds + (int64_t)b * C, blockIdx.y, C, 1.0f);
}
static bool pool_v8_ok(int C, int dtype, const void* a, const void* b, const void* c, const void* d) {
  return dtype == MMRCA_BF16 && C % 8 == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}

/* out[b, c] = mean over the HW rows of sample b (AdaptiveAvgPool2d(1) / x.mean([2, 3])) */
extern "C" int mmrca_rowpool_mean(const void* x, void* out, int B, int HW, int C, int dtype, void* stream) {
  MMRCA_REQUIRE(x && out && B > 0 && HW > 0 && C > 0, "rowpool_mean: bad arguments");
  if (pool_v8_ok(C, dtype, x, x, x, x)) {
    hipLaunchKernelGGL(rowpool_mean_v8_k, dim3(B, (C + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, HW, C);
    MMRCA_CHECK_LAUNCH("rowpool_mean(v8)");
    return 0;
  }
  MMRCA_DISPATCH_DTYPE(dtype, "rowpool_mean",
    hipLaunchKernelGGL(rowpool_mean_k<T>, dim3(B, (C + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)out, HW, C);)
  MMRCA_CHECK_LAUNCH("rowpool_mean");
  return 0;
}

// dx[b, p, c] (+)= dpool[b, c] / HW
template <typename T>
__global__ void rowpool_mean_bwd_k(const T* __restrict__ dpool, T* __restrict__ dx, int64_t n, int HW, int C, int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int c = (int)(idx % C);
  const int64_t b = idx / ((int64_t)HW * C);
  const float g = to_f(dpool[b * C + c]) / (float)HW;
  dx[idx] = from_f<T>(accumulate ? to_f(dx[idx]) + g : g);
}
// ---- 8 channels per thread (bf16, C % 8 == 0) for the element-wise squeeze-excitation / pooling passes
typedef __attribute__((ext_vector_type(8))) __bf16 se_b8;
__global__ void __launch_bounds__(256)
rowpool_mean_bwd_v8_k(const bf16_t* __restrict__ dpool, bf16_t* __restrict__ dx, int64_t n8, int HW, int C8, int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n8) return;
  const int c8 = (int)(idx % C8);
  const int64_t b = idx / ((int64_t)HW * C8);
  const se_b8 g = *reinterpret_cast<const se_b8*>(dpool + (b * C8 + c8) * 8);
  se_b8 o;
  if (accumulate) {
    const se_b8 d = *reinterpret_cast<const se_b8*>(dx + idx * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)d[j] + (float)g[j] / (float)HW);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)g[j] / (float)HW);
  }
  *reinterpret_cast<se_b8*>(dx + idx * 8) = o;
}
__global__ void __launch_bounds__(256)
se_scale_fwd_v8_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ sc, bf16_t* __restrict__ y, int64_t n8, int HW, int C8) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n8) return;
  const int c8 = (int)(idx % C8);
  const int64_t b = idx / ((int64_t)HW * C8);
  const se_b8 xv = *reinterpret_cast<const se_b8*>(x + idx * 8), sv = *reinterpret_cast<const se_b8*>(sc + (b * C8 + c8) * 8);
  se_b8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)xv[j] * (float)sv[j]);
  *reinterpret_cast<se_b8*>(y + idx * 8) = o;
}
static bool se_v8_ok(int C, int dtype, const void* a, const void* b, const void* c) {
  return dtype == MMRCA_BF16 && C % 8 == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}

extern "C" int mmrca_rowpool_mean_bwd(const void* dpool, void* dx, int B, int HW, int C, int accumulate, int dtype, void* stream) {
  MMRCA_REQUIRE(dpool && dx && B > 0 && HW > 0 && C > 0, "rowpool_mean_bwd: bad arguments");
  const int64_t n = (int64_t)B * HW * C;
  if (se_v8_ok(C, dtype, dpool, dx, dx)) {
    hipLaunchKernelGGL(rowpool_mean_bwd_v8_k, dim3(blocks_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dpool, (bf16_t*)dx,
                       n / 8, HW, C / 8, accumulate);
    MMRCA_CHECK_LAUNCH("rowpool_mean_bwd(v8)");
    return 0;
  }
  MMRCA_DISPATCH_DTYPE(dtype, "rowpool_mean_bwd",
    hipLaunchKernelGGL(rowpool_mean_bwd_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)dpool, (T*)dx, n, HW, C, accumulate);)
  MMRCA_CHECK_LAUNCH("rowpool_mean_bwd");
  return 0;
}

// y = x * s[b, c]
template <typename T>
__global__ void se_scale_fwd_k(const T* __restrict__ x, const T* __restrict__ s, T* __restrict__ y, int64_t n, int HW, int C) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int c = (int)(idx % C);
  const int64_t b = idx / ((int64_t)HW * C);
  y[idx] = from_f<T>(to_f(x[idx]) * to_f(s[b * C + c]));
}
extern "C" int mmrca_se_scale_fwd(const void* x, const void* s, void* y, int B, int HW, int C, int dtype, void* stream) {
  MMRCA_REQUIRE(x && s && y && B > 0 && HW > 0 && C > 0, "se_scale_fwd: bad arguments");
  const int64_t n = (int64_t)B * HW * C;
  if (se_v8_ok(C, dtype, x, s, y)) {
    hipLaunchKernelGGL(se_scale_fwd_v8_k, dim3(blocks_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)s,
                       (bf16_t*)y, n / 8, HW, C / 8);
    MMRCA_CHECK_LAUNCH("se_scale_fwd(v8)");
    return 0;
  }
  MMRCA_DISPATCH_DTYPE(dtype, "se_scale_fwd",
    hipLaunchKernelGGL(se_scale_fwd_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)s, (T*)y, n, HW, C);)
  MMRCA_CHECK_LAUNCH("se_scale_fwd");
  return 0;
}
// dx = dy * s[b, c];  ds[b, c] = sum over the sample's rows of dy * x
template <typename T>
__global__ void se_scale_bwd_k(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ s, T* __restrict__ dx,
                               T* __restrict__ ds, int HW, int C) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl, b = blockIdx.x;
  float acc = 0.f;
  if (c < C) {
    const float sv = to_f(s[(int64_t)b * C + c]);
    for (int p = rl; p < HW; p += 4) {
      const int64_t i = ((int64_t)b * HW + p) * C + c;
      const float g = to_f(dy[i]);
      acc = fmaf(g, to_f(x[i]), acc);
      dx[i] = from_f<T>(g * sv);
    }
  }
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < C) ds[(int64_t)b * C + c] = from_f<T>(red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}
extern "C" int mmrca_se_scale_bwd(const void* dy, const void* x, const void* s, void* dx, void* ds, int B, int HW, int C, int dtype, void* stream) {
  MMRCA_REQUIRE(dy && x && s && ds && B > 0 && HW > 0 && C > 0, "se_scale_bwd: bad arguments");
  if (pool_v8_ok(C, dtype, dy, x, s, dx)) {
    hipLaunchKernelGGL(se_scale_bwd_v8_k, dim3(B, (C + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x,
                       (const bf16_t*)s, (bf16_t*)dx, (bf16_t*)ds, HW, C);
    MMRCA_CHECK_LAUNCH("se_scale_bwd(v8)");
    return 0;
  }
  MMRCA_REQUIRE(dx, "se_scale_bwd: dx == NULL (ds only) is built for the bf16 / C %% 8 == 0 / aligned case only");
  MMRCA_DISPATCH_DTYPE(dtype, "se_scale_bwd",
    hipLaunchKernelGGL(se_scale_bwd_k<T>, dim3(B, (C + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)x, (const T*)s,
                       (T*)dx, (T*)ds, HW, C);)
  MMRCA_CHECK_LAUNCH("se_scale_bwd");
  return 0;
}

// Second half of the squeeze-excitation backward, in one pass: dx = dy * s[b, c] + dpool[b, c] / HW (the scale path plus the pooled
// path; the two-kernel form wrote dy * s, then read and re-wrote it to add the pooled term).  dx is the gradient at the output of
// the depthwise BatchNorm + SiLU that feeds the block, so with BN the sums that BatchNorm backward needs -- sum du and
// sum du * xhat, du = dx * act'(xhat gamma + beta) -- are accumulated here as well (z = that BatchNorm's input): its separate
// reduce pass over dx and z disappears.  block = (sample, 64 channels), as se_scale_bwd_v8_k.
template <bool BN>
__global__ void __launch_bounds__(256)
se_dx_v8_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ s, const bf16_t* __restrict__ dpool, bf16_t* __restrict__ dx, int HW, int C,
           float inv_hw, const bf16_t* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ rstd,
           const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta, int act, float* __restrict__ sum_du, float* __restrict__ sum_duxh) {
  __shared__ float red[4][8][17];
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c0 = blockIdx.y * 64 + cg * 8, b = blockIdx.x;
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
  if (c0 < C) {
    const pl_b8 s8 = *reinterpret_cast<const pl_b8*>(s + (int64_t)b * C + c0), p8 = *reinterpret_cast<const pl_b8*>(dpool + (int64_t)b * C + c0);
    float sc[8], pd[8], m[8], rs[8], ga[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = (float)s8[j]; pd[j] = (float)p8[j] * inv_hw; }
    if (BN) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { m[j] = mean[c0 + j]; rs[j] = rstd[c0 + j]; ga[j] = (float)gamma[c0 + j]; be[j] = (float)beta[c0 + j]; }
    }
    int p = rl;
    for (; p + 32 < HW; p += 64) {                       // two rows in flight per thread
      const int64_t i0 = ((int64_t)b * HW + p) * C + c0, i1 = i0 + (int64_t)32 * C;
      const pl_b8 g0 = *reinterpret_cast<const pl_b8*>(dy + i0), g1 = *reinterpret_cast<const pl_b8*>(dy + i1);
      pl_b8 z0, z1;
      if (BN) { z0 = *reinterpret_cast<const pl_b8*>(z + i0); z1 = *reinterpret_cast<const pl_b8*>(z + i1); }
      pl_b8 o0, o1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o0[j] = (bf16_t)fmaf((float)g0[j], sc[j], pd[j]);
        o1[j] = (bf16_t)fmaf((float)g1[j], sc[j], pd[j]);
        if (BN) {
          const float x0 = ((float)z0[j] - m[j]) * rs[j], x1 = ((float)z1[j] - m[j]) * rs[j];
          const float d0 = (float)o0[j] * act_grad_f(x0 * ga[j] + be[j], act), d1 = (float)o1[j] * act_grad_f(x1 * ga[j] + be[j], act);
          acc[0][j] += d0 + d1;
          acc[1][j] = fmaf(d0, x0, fmaf(d1, x1, acc[1][j]));
        }
      }
      *reinterpret_cast<pl_b8*>(dx + i0) = o0;
      *reinterpret_cast<pl_b8*>(dx + i1) = o1;
    }
    for (; p < HW; p += 32) {
      const int64_t i0 = ((int64_t)b * HW + p) * C + c0;
      const pl_b8 g0 = *reinterpret_cast<const pl_b8*>(dy + i0);
      pl_b8 z0;
      if (BN) z0 = *reinterpret_cast<const pl_b8*>(z + i0);
      pl_b8 o0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o0[j] = (bf16_t)fmaf((float)g0[j], sc[j], pd[j]);
        if (BN) {
          const float x0 = ((float)z0[j] - m[j]) * rs[j];
          const float d0 = (float)o0[j] * act_grad_f(x0 * ga[j] + be[j], act);
          acc[0][j] += d0;
          acc[1][j] = fmaf(d0, x0, acc[1][j]);
        }
      }
      *reinterpret_cast<pl_b8*>(dx + i0) = o0;
    }
  }
  if (BN) {
    float* const outs[2] = {sum_du, sum_duxh};
    col_reduce8<2>(acc, red, outs, blockIdx.y, C);
  }
}

/* dx[b*HW + p, c] = dy[..] * s[b, c] + dpool[b, c] / HW  (bf16, C % 8 == 0, 16-byte aligned; other cases return -3: the caller keeps
 * mmrca_se_scale_bwd + mmrca_rowpool_mean_bwd).  With z given, sums[0..C) += sum du and sums[C..2C) += sum du * xhat of the BatchNorm
 * + activation whose output gradient dx is (du = dx * act'(xhat gamma + beta), xhat = (z - mean) rstd): exactly what
 * mmrca_bn_act_bwd's first pass computes, for mmrca_bn_act_bwd_sums. */
extern "C" int mmrca_se_dx(const void* dy, const void* s, const void* dpool, void* dx, int B, int HW, int C, int dtype, const void* z,
                           const float* mean, const float* rstd, const void* gamma, const void* beta, int act, float* sums, void* stream) {
  MMRCA_REQUIRE(dy && s && dpool && dx && B > 0 && HW > 0 && C > 0, "se_dx: bad arguments");
  MMRCA_REQUIRE(!z || (mean && rstd && gamma && beta && sums && act >= 0 && act <= 3), "se_dx: the BatchNorm sums need mean, rstd, gamma, beta, sums");
  if (!(pool_v8_ok(C, dtype, dy, s, dpool, dx) && (!z || ((((uintptr_t)z) & 15) == 0))))
    return mmrca_fail(-3, "se_dx: only the bf16 / C %% 8 == 0 / 16-byte aligned case is built");
  const dim3 grid(B, (C + 63) / 64);
  if (z) hipLaunchKernelGGL(se_dx_v8_k<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)s, (const bf16_t*)dpool,
                            (bf16_t*)dx, HW, C, 1.0f / (float)HW, (const bf16_t*)z, mean, rstd, (const bf16_t*)gamma, (const bf16_t*)beta, act,
                            sums, sums + C);
  else hipLaunchKernelGGL(se_dx_v8_k<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)s, (const bf16_t*)dpool,
                          (bf16_t*)dx, HW, C, 1.0f / (float)HW, (const bf16_t*)nullptr, (const float*)nullptr, (const float*)nullptr,
                          (const bf16_t*)nullptr, (const bf16_t*)nullptr, 0, (float*)nullptr, (float*)nullptr);
  MMRCA_CHECK_LAUNCH("se_dx");
  return 0;
}

// elementwise activation on [n] with an optional per-column bias (the 1x1 "convolutions" of squeeze-excitation carry a bias)
template <typename T>
__global__ void bias_act_fwd_k(const T* __restrict__ x, const T* __restrict__ bias, T* __restrict__ y, int64_t n, int C, int act) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float u = to_f(x[idx]) + (bias ? to_f(bias[idx % C]) : 0.f);
  y[idx] = from_f<T>(act_f(u, act));
}
/* y = act(x + bias[c]) on [rows, C]; x is the pre-activation WITHOUT bias (kept for the backward) */
extern "C" int mmrca_bias_act_fwd(const void* x, const void* bias, void* y, int64_t rows, int C, int act, int dtype, void* stream) {
  MMRCA_REQUIRE(x && y && rows > 0 && C > 0 && act >= 0 && act <= 3, "bias_act_fwd: bad arguments");
  const int64_t n = rows * C;
  MMRCA_DISPATCH_DTYPE(dtype, "bias_act_fwd",
    hipLaunchKernelGGL(bias_act_fwd_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)bias, (T*)y, n, C, act);)
  MMRCA_CHECK_LAUNCH("bias_act_fwd");
  return 0;
}
// dx = dy * act'(x + bias);  dbias[c] += column sums of dx (tiny matrices: one thread per column)
template <typename T>
__global__ void bias_act_bwd_k(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ bias, T* __restrict__ dx,
                               float* __restrict__ dbias, int64_t rows, int C, int act) {
  // block = 64 columns x 4 row lanes (the [B, C] matrices of squeeze-excitation: one thread per column walking all B rows was
  // 60 us of dependent loads per call)
  // (round 3: 16 row lanes and four rows in flight per thread -- with 4 lanes a thread walked B / 4 = 32 rows one dependent pair of
  // 2-byte loads at a time: 18 us per call, 122 calls per EfficientNetV2-L step)
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (c < C) {
    const float b = bias ? to_f(bias[c]) : 0.f;
    int64_t r = rl;
    for (; r + 48 < rows; r += 64) {
      float gy[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { gy[u] = to_f(dy[(r + 16 * u) * C + c]); xv[u] = to_f(x[(r + 16 * u) * C + c]); }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float g = gy[u] * act_grad_f(xv[u] + b, act);
        dx[(r + 16 * u) * C + c] = from_f<T>(g);
        s += g;
      }
    }
    for (; r < rows; r += 16) {
      const float g = to_f(dy[r * C + c]) * act_grad_f(to_f(x[r * C + c]) + b, act);
      dx[r * C + c] = from_f<T>(g);
      s += g;
    }
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C && dbias) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cl];
    dbias[c] += t;
  }
}
extern "C" int mmrca_bias_act_bwd(const void* dy, const void* x, const void* bias, void* dx, float* dbias, int64_t rows, int C, int act,
                                  int dtype, void* stream) {
  MMRCA_REQUIRE(dy && x && dx && rows > 0 && C > 0 && act >= 0 && act <= 3, "bias_act_bwd: bad arguments");
  MMRCA_DISPATCH_DTYPE(dtype, "bias_act_bwd",
    hipLaunchKernelGGL(bias_act_bwd_k<T>, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, (const T*)dy, (const T*)x, (const T*)bias,
                       (T*)dx, dbias, rows, C, act);)
  MMRCA_CHECK_LAUNCH("bias_act_bwd");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Squeeze-excitation MLP, fused (round 5).  torchvision SqueezeExcitation inside efficientnet_v2_*'s MBConv blocks (the reference
// builds them at CVPR_code/multimodal_model.py:113-126): scale = sigmoid(fc2(silu(fc1(avgpool(x))))) on [B, c] -> [B, sq] -> [B, c].
// As GEMMs these are M = B (64 .. 128) products: four launches forward and six backward per block, each one or two
// workgroups walking a long contraction -- 212 launches and 11 ms of an 79 ms EfficientNetV2-M step (14 %) for ~0 FLOPs
// (and sq = 20 / 44 / 76 is not a multiple of 8: the general kernel).  Here ONE workgroup per sample runs the whole chain out of
// LDS / registers, every weight read coalesced; the weight gradients (sums over the batch) are a second small launch.
// The tensors the backward reads (pooled, h_pre, h, s_pre, s: pre-activations WITHOUT bias, rounded to T like the GEMM outputs they
// replace) keep their layout.
// ---------------------------------------------------------------------------------------------------------------------
// (every loop below keeps several independent 8 / 16-byte loads in flight and reduces across lanes at most a few times per wave:
// the first version -- a wave_sum per output and one dependent load per iteration -- ran the chain in ~150 us per block)
// Warm-up of the two weight matrices (round 6).  The fused squeeze-excitation MLP is a chain of dependent phases on ~0.1-1.2 MB of weights
// that nothing has touched since the previous step: in the step it ran at 52 us per launch against 19 us with the weights in L2
// (tools/se_bench.py) -- four or five cold HBM round trips in a row.  Every thread therefore touches its share of the cache lines of W1
// and W2 (one 4-byte load per 128-byte line, up to 1 MiB / 768 KiB per matrix) before the first phase: ONE round trip brings both matrices
// into L2, and the values are only summed at the very end, so nothing waits for them.
template <int NT, int KW>
__device__ __forceinline__ void se_warm(const void* W1, const void* W2, size_t bytes, float (&wv)[2 * KW]) {
  const char* a = reinterpret_cast<const char*>(W1);
  const char* b = reinterpret_cast<const char*>(W2);
#pragma unroll
  for (int k = 0; k < KW; ++k) {
    const size_t off = ((size_t)k * NT + threadIdx.x) * 128;
    wv[k] = off + 4 <= bytes ? *reinterpret_cast<const float*>(a + off) : 0.f;
    wv[KW + k] = off + 4 <= bytes ? *reinterpret_cast<const float*>(b + off) : 0.f;
  }
}
template <int N2>
__device__ __forceinline__ bool se_warm_done(const float (&wv)[N2]) {
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < N2; ++k) t += wv[k];
  return t == 1.2345678e-37f;          // (never: keeps the loads alive without a store)
}

template <typename T, int NW>
__global__ void __launch_bounds__(64 * NW)
se_mlp_fwd_k(const T* __restrict__ pooled, const T* __restrict__ W1, const T* __restrict__ b1, const T* __restrict__ W2,
             const T* __restrict__ b2, T* __restrict__ h_pre, T* __restrict__ h, T* __restrict__ s_pre, T* __restrict__ s, int c, int sq, int warm) {
  extern __shared__ __attribute__((aligned(16))) float se_sm[];              // pooled [c] | h [sq]
  float* pf = se_sm;
  float* hf = se_sm + c;
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NT = 64 * NW;        // sixteen waves per sample: B <= 128 workgroups run side by side, so the launch lasts as long as ONE does
  float wv[16];
  se_warm<NT, 8>(W1, W2, warm ? (size_t)c * sq * sizeof(T) : 0, wv);
  for (int i = threadIdx.x * 4; i < c; i += 4 * NT) {
    const Vec4<T> v = Vec4<T>::load(pooled + (int64_t)b * c + i);
#pragma unroll
    for (int k = 0; k < 4; ++k) pf[i + k] = v.v[k];
  }
  __syncthreads();
  // fc1 (W1 [sq, c]): a wave takes its share of the outputs four at a time, lanes along the contraction; four reductions interleaved
  const int jw = (sq + NW - 1) / NW, jend = min(sq, (wave + 1) * jw);
  for (int j0 = wave * jw; j0 < jend; j0 += 4) {
    const int nj = min(4, jend - j0);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = lane * 4; i < c; i += 256) {
      const float4 p4 = *reinterpret_cast<const float4*>(pf + i);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < nj) {
          const Vec4<T> w = Vec4<T>::load(W1 + (int64_t)(j0 + u) * c + i);
          a[u] += w.v[0] * p4.x + w.v[1] * p4.y + w.v[2] * p4.z + w.v[3] * p4.w;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] += __shfl_xor(a[u], o, 64);
    if (lane == 0)
      for (int u = 0; u < nj; ++u) {
        const int j = j0 + u;
        const T hp = from_f<T>(a[u]);
        const T hv = from_f<T>(act_f(to_f(hp) + to_f(b1[j]), CONV_ACT_SILU));
        h_pre[(int64_t)b * sq + j] = hp;
        h[(int64_t)b * sq + j] = hv;
        hf[j] = to_f(hv);
      }
  }
  __syncthreads();
  // fc2 (W2 [c, sq]): a thread per output walks its own (contiguous) row, two rows at a time
  for (int i0 = threadIdx.x; i0 < c; i0 += 2 * NT) {
    const int i1 = i0 + NT < c ? i0 + NT : i0;
    const T* r0 = W2 + (int64_t)i0 * sq;
    const T* r1 = W2 + (int64_t)i1 * sq;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 4
    for (int j = 0; j < sq; j += 4) {
      const Vec4<T> w0 = Vec4<T>::load(r0 + j), w1 = Vec4<T>::load(r1 + j);
      const float4 h4 = *reinterpret_cast<const float4*>(hf + j);
      a0 += w0.v[0] * h4.x + w0.v[1] * h4.y + w0.v[2] * h4.z + w0.v[3] * h4.w;
      a1 += w1.v[0] * h4.x + w1.v[1] * h4.y + w1.v[2] * h4.z + w1.v[3] * h4.w;
    }
    {
      const T sp = from_f<T>(a0);
      s_pre[(int64_t)b * c + i0] = sp;
      s[(int64_t)b * c + i0] = from_f<T>(act_f(to_f(sp) + to_f(b2[i0]), CONV_ACT_SIGMOID));
    }
    if (i1 != i0) {
      const T sp = from_f<T>(a1);
      s_pre[(int64_t)b * c + i1] = sp;
      s[(int64_t)b * c + i1] = from_f<T>(act_f(to_f(sp) + to_f(b2[i1]), CONV_ACT_SIGMOID));
    }
  }
  if (se_warm_done(wv)) hf[0] = 0.f;
}

// backward chain of one sample: ds -> ds_pre (x sigmoid') -> dh = ds_pre . W2 -> dh_pre (x silu') -> dpool = dh_pre . W1; the bias
// gradients are the batch sums of ds_pre / dh_pre (fp32 atomics, B addends per entry)
template <typename T, int NW>
__global__ void __launch_bounds__(64 * NW)
se_mlp_bwd_k(const T* __restrict__ ds, const T* __restrict__ s_pre, const T* __restrict__ h_pre, const T* __restrict__ W1,
             const T* __restrict__ b1, const T* __restrict__ W2, const T* __restrict__ b2, T* __restrict__ ds_pre, T* __restrict__ dh_pre,
             T* __restrict__ dpool, float* __restrict__ gb1, float* __restrict__ gb2, int c, int sq, int warm) {
  extern __shared__ __attribute__((aligned(16))) float se_sm[];              // ds_pre [c] | partial dh [NW][sq] | dh_pre [sq]
  constexpr int NT = 64 * NW;
  float* dsp = se_sm;
  float* part = se_sm + c;
  float* dhp = part + NW * sq;
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int KW = NT >= 1024 ? 8 : 24;      // 8 x 1,024 or 24 x 256 lines of 128 bytes per matrix
  float wv[2 * KW];
  se_warm<NT, KW>(W2, W1, warm ? (size_t)c * sq * sizeof(T) : 0, wv);
  for (int i = threadIdx.x; i < c; i += NT) {
    const float g = to_f(ds[(int64_t)b * c + i]) * act_grad_f(to_f(s_pre[(int64_t)b * c + i]) + to_f(b2[i]), CONV_ACT_SIGMOID);
    const T gt = from_f<T>(g);
    ds_pre[(int64_t)b * c + i] = gt;
    dsp[i] = to_f(gt);
    atomicAdd(gb2 + i, g);
  }
  __syncthreads();
  // dh[j] = sum_i ds_pre[i] W2[i][j], 32 columns at a time: a thread walks its own rows (contiguous 8 / 16-byte pieces) into 32
  // private sums, the wave adds them up (32 independent butterflies), the four waves meet in LDS
  for (int jc = 0; jc < sq; jc += 32) {
    const int nq = min(8, (sq - jc) >> 2);
    float acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.f;
    for (int i = threadIdx.x; i < c; i += NT) {
      const float d = dsp[i];
      const T* r = W2 + (int64_t)i * sq + jc;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (q < nq) {
          const Vec4<T> w = Vec4<T>::load(r + 4 * q);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[4 * q + k] += d * w.v[k];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int k = 0; k < 32; ++k) acc[k] += __shfl_xor(acc[k], o, 64);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 32; ++k)
        if (k < 4 * nq) part[wave * sq + jc + k] = acc[k];
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < sq; j += NT) {
    float dh = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) dh += part[w * sq + j];
    const float g = to_f(from_f<T>(dh)) * act_grad_f(to_f(h_pre[(int64_t)b * sq + j]) + to_f(b1[j]), CONV_ACT_SILU);
    const T gt = from_f<T>(g);
    dh_pre[(int64_t)b * sq + j] = gt;
    dhp[j] = to_f(gt);
    atomicAdd(gb1 + j, g);
  }
  __syncthreads();
  // dpool[i] = sum_j dh_pre[j] W1[j][i]: a thread per four consecutive outputs (coalesced rows of W1 [sq, c])
  for (int i = threadIdx.x * 4; i < c; i += 4 * NT) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int j = 0; j < sq; ++j) {
      const Vec4<T> w = Vec4<T>::load(W1 + (int64_t)j * c + i);
      const float d = dhp[j];
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] += d * w.v[k];
    }
    Vec4<T> o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o.v[k] = a[k];
    o.store(dpool + (int64_t)b * c + i);
  }
  if (se_warm_done(wv)) dhp[0] = 0.f;
}

// gW2[i][j] += sum_b ds_pre[b][i] h[b][j]   (blocks [0, n2));   gW1[j][i] += sum_b dh_pre[b][j] pooled[b][i]   (the rest)
template <typename T>
__global__ void __launch_bounds__(256)
se_mlp_wgrad_k(const T* __restrict__ ds_pre, const T* __restrict__ h, const T* __restrict__ dh_pre, const T* __restrict__ pooled,
               float* __restrict__ gW1, float* __restrict__ gW2, int B, int c, int sq, int n2_blocks) {
  const int64_t n = (int64_t)c * sq;
  if ((int)blockIdx.x < n2_blocks) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int i = (int)(e / sq), j = (int)(e - (int64_t)i * sq);
    float a = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) a += to_f(ds_pre[(int64_t)b * c + i]) * to_f(h[(int64_t)b * sq + j]);
    gW2[e] += a;
  } else {
    const int64_t e = (int64_t)((int)blockIdx.x - n2_blocks) * 256 + threadIdx.x;
    if (e >= n) return;
    const int j = (int)(e / c), i = (int)(e - (int64_t)j * c);
    float a = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) a += to_f(dh_pre[(int64_t)b * sq + j]) * to_f(pooled[(int64_t)b * c + i]);
    gW1[e] += a;
  }
}

static const int g_se_warm = getenv("MMRCA_SE_WARM") ? atoi(getenv("MMRCA_SE_WARM")) : 1;
extern "C" int mmrca_se_mlp_fwd(const void* pooled, const void* w1, const void* b1, const void* w2, const void* b2, void* h_pre, void* h,
                                void* s_pre, void* s, int B, int c, int sq, int dtype, void* stream) {
  MMRCA_REQUIRE(pooled && w1 && b1 && w2 && b2 && h_pre && h && s_pre && s, "se_mlp_fwd: null pointer");
  MMRCA_REQUIRE(B > 0 && c > 0 && sq > 0 && c % 4 == 0 && sq % 4 == 0 && (size_t)(c + sq) * 4 <= 64 * 1024,
                "se_mlp_fwd: bad shape B=%d c=%d sq=%d (c and sq must be multiples of 4)", B, c, sq);
  // every operand is read (and every output written) in 4-element vectors: 8-byte alignment for bf16, 16 for fp32
  {
    const uintptr_t al = (dtype == MMRCA_F32 ? 16 : 8) - 1;
    MMRCA_REQUIRE((((uintptr_t)pooled | (uintptr_t)w1 | (uintptr_t)b1 | (uintptr_t)w2 | (uintptr_t)b2 | (uintptr_t)h_pre | (uintptr_t)h |
                    (uintptr_t)s_pre | (uintptr_t)s) & al) == 0,
                  "se_mlp_fwd: every pointer must be %d-byte aligned (4-element vector accesses)", (int)al + 1);
  }
  const size_t lds = (size_t)(c + sq) * sizeof(float);
  MMRCA_DISPATCH_DTYPE(dtype, "se_mlp_fwd",
    hipLaunchKernelGGL((se_mlp_fwd_k<T, 16>), dim3(B), dim3(1024), lds, (hipStream_t)stream, (const T*)pooled, (const T*)w1, (const T*)b1, (const T*)w2,
                       (const T*)b2, (T*)h_pre, (T*)h, (T*)s_pre, (T*)s, c, sq, g_se_warm);)
  MMRCA_CHECK_LAUNCH("se_mlp_fwd");
  return 0;
}

extern "C" int mmrca_se_mlp_bwd(const void* ds, const void* pooled, const void* h_pre, const void* h, const void* s_pre, const void* w1,
                                const void* b1, const void* w2, const void* b2, void* ds_pre, void* dh_pre, void* dpool, float* gw1,
                                float* gb1, float* gw2, float* gb2, int B, int c, int sq, int dtype, void* stream) {
  MMRCA_REQUIRE(ds && pooled && h_pre && h && s_pre && w1 && b1 && w2 && b2 && ds_pre && dh_pre && dpool && gw1 && gb1 && gw2 && gb2,
                "se_mlp_bwd: null pointer");
  MMRCA_REQUIRE(B > 0 && c > 0 && sq > 0 && c % 4 == 0 && sq % 4 == 0 && (size_t)(c + 5 * sq) * 4 <= 64 * 1024,
                "se_mlp_bwd: bad shape B=%d c=%d sq=%d (c and sq must be multiples of 4)", B, c, sq);
  {
    const uintptr_t al = (dtype == MMRCA_F32 ? 16 : 8) - 1;
    MMRCA_REQUIRE((((uintptr_t)ds | (uintptr_t)pooled | (uintptr_t)h_pre | (uintptr_t)h | (uintptr_t)s_pre | (uintptr_t)w1 | (uintptr_t)b1 |
                    (uintptr_t)w2 | (uintptr_t)b2 | (uintptr_t)ds_pre | (uintptr_t)dh_pre | (uintptr_t)dpool) & al) == 0,
                  "se_mlp_bwd: every activation / weight pointer must be %d-byte aligned (4-element vector accesses)", (int)al + 1);
  }
  const size_t lds = (size_t)(c + 5 * sq) * sizeof(float);
  const int64_t n = (int64_t)c * sq;
  const int nb = (int)((n + 255) / 256);
  MMRCA_DISPATCH_DTYPE(dtype, "se_mlp_bwd",
    // (four waves per sample here: with sixteen the 32-value butterflies of the dh stage outweigh the shorter row walk -- measured)
    hipLaunchKernelGGL((se_mlp_bwd_k<T, 4>), dim3(B), dim3(256), lds, (hipStream_t)stream, (const T*)ds, (const T*)s_pre, (const T*)h_pre,
                       (const T*)w1, (const T*)b1, (const T*)w2, (const T*)b2, (T*)ds_pre, (T*)dh_pre, (T*)dpool, gb1, gb2, c, sq, g_se_warm);
    hipLaunchKernelGGL(se_mlp_wgrad_k<T>, dim3(2 * nb), dim3(256), 0, (hipStream_t)stream, (const T*)ds_pre, (const T*)h, (const T*)dh_pre,
                       (const T*)pooled, gw1, gw2, B, c, sq, nb);)
  MMRCA_CHECK_LAUNCH("se_mlp_bwd");
  return 0;
}

// out = a + b * rowscale[sample]   (residual connection with torchvision's "row" stochastic depth; rowscale NULL = 1)
__global__ void __launch_bounds__(256)
residual_add_v8_k(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, const float* __restrict__ rowscale, bf16_t* __restrict__ out,
                  int64_t n8, int64_t per_sample8) {
  typedef __attribute__((ext_vector_type(8))) __bf16 b8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n8) return;
  const float sc = rowscale ? rowscale[idx / per_sample8] : 1.f;
  const b8 bv = *reinterpret_cast<const b8*>(b + idx * 8);
  b8 o;
  if (a) {
    const b8 av = *reinterpret_cast<const b8*>(a + idx * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)av[j] + (float)bv[j] * sc);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(0.f + (float)bv[j] * sc);
  }
  *reinterpret_cast<b8*>(out + idx * 8) = o;
}
template <typename T>
__global__ void residual_add_k(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ rowscale, T* __restrict__ out,
                               int64_t n, int64_t per_sample) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float s = rowscale ? rowscale[idx / per_sample] : 1.f;
  out[idx] = from_f<T>((a ? to_f(a[idx]) : 0.f) + to_f(b[idx]) * s);
}
/* out[b, ...] = a[b, ...] + branch[b, ...] * rowscale[b]; a == NULL gives the scaled branch alone (its backward) */
extern "C" int mmrca_residual_add(const void* a, const void* branch, const float* rowscale, void* out, int B, int64_t per_sample, int dtype,
                                  void* stream) {
  MMRCA_REQUIRE(branch && out && B > 0 && per_sample > 0, "residual_add: bad arguments");
  const int64_t n = (int64_t)B * per_sample;
  if (dtype == MMRCA_BF16 && per_sample % 8 == 0 && ((((uintptr_t)a) | ((uintptr_t)branch) | ((uintptr_t)out)) & 15) == 0) {
    hipLaunchKernelGGL(residual_add_v8_k, dim3(blocks_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)branch,
                       rowscale, (bf16_t*)out, n / 8, per_sample / 8);
    MMRCA_CHECK_LAUNCH("residual_add(v8)");
    return 0;
  }
  MMRCA_DISPATCH_DTYPE(dtype, "residual_add",
    hipLaunchKernelGGL(residual_add_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)a, (const T*)branch, rowscale,
                       (T*)out, n, per_sample);)
  MMRCA_CHECK_LAUNCH("residual_add");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 2 / padding 1 max pooling (ShuffleNetV2's stem): the argmax tap is kept for the backward
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool3x3s2_fwd_k(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ arg, int B, int H, int W, int C,
                                   int Ho, int Wo) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * Ho * Wo * C) return;
  const int c = (int)(idx % C);
  const int64_t op = idx / C;
  const int ox = (int)(op % Wo), oy = (int)((op / Wo) % Ho), b = (int)(op / ((int64_t)Wo * Ho));
  float best = -INFINITY;
  int bt = 0;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
        const float v = to_f(x[(((int64_t)b * H + iy) * W + ix) * C + c]);
        if (v > best) { best = v; bt = ky * 3 + kx; }       // first maximum wins, as torch's max_pool2d
      }
    }
  y[idx] = from_f<T>(best);
  arg[idx] = (unsigned char)bt;
}
template <typename T>
__global__ void maxpool3x3s2_bwd_k(const T* __restrict__ dy, const unsigned char* __restrict__ arg, T* __restrict__ dx, int B, int H, int W,
                                   int C, int Ho, int Wo) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // input pixel: gather from the <= 4 windows that contain it
  if (idx >= (int64_t)B * H * W * C) return;
  const int c = (int)(idx % C);
  const int64_t ip = idx / C;
  const int ix = (int)(ip % W), iy = (int)((ip / W) % H), b = (int)(ip / ((int64_t)W * H));
  float s = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = iy + 1 - ky;
    if (ty < 0 || (ty & 1)) continue;
    const int oy = ty >> 1;
    if (oy >= Ho) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = ix + 1 - kx;
      if (tx < 0 || (tx & 1)) continue;
      const int ox = tx >> 1;
      if (ox >= Wo) continue;
      const int64_t o = (((int64_t)b * Ho + oy) * Wo + ox) * C + c;
      if (arg[o] == ky * 3 + kx) s += to_f(dy[o]);
    }
  }
  dx[idx] = from_f<T>(s);
}
extern "C" int mmrca_maxpool3x3s2_fwd(const void* x, void* y, void* argmax, int B, int H, int W, int C, int dtype, void* stream) {
  MMRCA_REQUIRE(x && y && argmax && B > 0 && H > 0 && W > 0 && C > 0, "maxpool3x3s2_fwd: bad arguments");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t n = (int64_t)B * Ho * Wo * C;
  MMRCA_DISPATCH_DTYPE(dtype, "maxpool3x3s2_fwd",
    hipLaunchKernelGGL(maxpool3x3s2_fwd_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y,
                       (unsigned char*)argmax, B, H, W, C, Ho, Wo);)
  MMRCA_CHECK_LAUNCH("maxpool3x3s2_fwd");
  return 0;
}
extern "C" int mmrca_maxpool3x3s2_bwd(const void* dy, const void* argmax, void* dx, int B, int H, int W, int C, int dtype, void* stream) {
  MMRCA_REQUIRE(dy && dx && argmax && B > 0 && H > 0 && W > 0 && C > 0, "maxpool3x3s2_bwd: bad arguments");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t n = (int64_t)B * H * W * C;
  MMRCA_DISPATCH_DTYPE(dtype, "maxpool3x3s2_bwd",
    hipLaunchKernelGGL(maxpool3x3s2_bwd_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)dy,
                       (const unsigned char*)argmax, (T*)dx, B, H, W, C, Ho, Wo);)
  MMRCA_CHECK_LAUNCH("maxpool3x3s2_bwd");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// channel gather: out[r, j] = in[r, map[j]] for j < Cout (split / concat / ShuffleNetV2's channel shuffle and their
// backward are all instances; map is an int32 device array)
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void channel_gather_k(const T* __restrict__ in, const int* __restrict__ map, T* __restrict__ out, int64_t n, int Cin, int Cout,
                                 int64_t ld_out, int col0) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int j = (int)(idx % Cout);
  const int64_t r = idx / Cout;
  out[r * ld_out + col0 + j] = in[r * Cin + map[j]];
}
// torch channel_shuffle(cat(a', b), groups = 2) of ShuffleNetV2 in ONE launch (it was three gathers through a concat buffer):
// out[r, 2j] = a[r, j], out[r, 2j + 1] = b[r, j], j in [0, bf); a has row pitch lda (its first bf columns are used), b and out are contiguous
template <typename T>
__global__ void channel_interleave2_k(const T* __restrict__ a, int64_t lda, const T* __restrict__ b, T* __restrict__ out, int64_t n, int bf) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int c = (int)(idx % (2 * bf));
  const int64_t r = idx / (2 * bf);
  out[idx] = (c & 1) ? b[r * bf + (c >> 1)] : a[r * lda + (c >> 1)];
}
// its backward in one launch (it was two gathers): d_even[r, j] = dout[r, 2j] (row pitch ld_even: the caller may aim it at the first half of
// the concatenated input gradient), d_odd[r, j] = dout[r, 2j + 1] (contiguous)
template <typename T>
__global__ void channel_deinterleave2_k(const T* __restrict__ dout, T* __restrict__ d_even, int64_t ld_even, T* __restrict__ d_odd, int64_t n, int bf) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int c = (int)(idx % (2 * bf));
  const int64_t r = idx / (2 * bf);
  const T v = dout[idx];
  if (c & 1) d_odd[r * bf + (c >> 1)] = v; else d_even[r * ld_even + (c >> 1)] = v;
}
extern "C" int mmrca_channel_interleave2(const void* a, int64_t lda, const void* b, void* out, int64_t rows, int bf, int dtype, void* stream) {
  MMRCA_REQUIRE(a && b && out && rows > 0 && bf > 0 && lda >= bf, "channel_interleave2: bad arguments");
  const int64_t n = rows * 2 * bf;
  MMRCA_DISPATCH_DTYPE(dtype, "channel_interleave2",
    hipLaunchKernelGGL(channel_interleave2_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)a, lda, (const T*)b, (T*)out, n, bf);)
  MMRCA_CHECK_LAUNCH("channel_interleave2");
  return 0;
}
extern "C" int mmrca_channel_deinterleave2(const void* dout, void* d_even, int64_t ld_even, void* d_odd, int64_t rows, int bf, int dtype, void* stream) {
  MMRCA_REQUIRE(dout && d_even && d_odd && rows > 0 && bf > 0 && ld_even >= bf, "channel_deinterleave2: bad arguments");
  const int64_t n = rows * 2 * bf;
  MMRCA_DISPATCH_DTYPE(dtype, "channel_deinterleave2",
    hipLaunchKernelGGL(channel_deinterleave2_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)dout, (T*)d_even, ld_even,
                       (T*)d_odd, n, bf);)
  MMRCA_CHECK_LAUNCH("channel_deinterleave2");
  return 0;
}
/* out[r, col0 + j] = in[r, map[j]], j in [0, Cout): in is [rows, Cin] contiguous, out has leading dimension ld_out */
extern "C" int mmrca_channel_gather(const void* in, const int* map, void* out, int64_t rows, int Cin, int Cout, int64_t ld_out, int col0,
                                    int dtype, void* stream) {
  MMRCA_REQUIRE(in && map && out && rows > 0 && Cin > 0 && Cout > 0 && ld_out >= col0 + Cout, "channel_gather: bad arguments");
  const int64_t n = rows * Cout;
  MMRCA_DISPATCH_DTYPE(dtype, "channel_gather",
    hipLaunchKernelGGL(channel_gather_k<T>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)in, map, (T*)out, n, Cin,
                       Cout, ld_out, col0);)
  MMRCA_CHECK_LAUNCH("channel_gather");
  return 0;
}
