// K7b: the training-time augmentations of the reference's albumentations pipeline
// (main_both.py:407-429) on the GPU, one descriptor per image, the whole batch per launch.
//
//   stage 0  A.Rotate(limit 90, INTER_LINEAR, BORDER_CONSTANT 0, crop_border=True)   mmrca_image_rotate_crop
//            on the decoded image (any size), result = a smaller uint8 image in the same staging buffer
//   stage 1  PadToMaintainAR -> A.Resize(INTER_LINEAR)                               mmrca_image_resize_u8 (preprocess.hip)
//   stage 2  A.GaussianBlur(3..7) -> VerticalFlip -> HorizontalFlip -> A.RandomBrightnessContrast       aug_filter_k<0>
//   stage 3  A.Sharpen                                                                                  aug_filter_k<1>
//   stage 4  A.Perspective(keep_size)                                                                   aug_warp_k<0>
//   stage 5  A.ShiftScaleRotate(scale only) -> A.Normalize -> ToTensorV2                                aug_warp_k<1>
//
// Every stage works on uint8 images and re-quantises like cv2 does (a cv2 call on a uint8 image returns uint8), so the
// stages cannot be folded into one resampling.  The host decides which transforms fire and draws their parameters;
// an image whose transform did not fire passes through that stage unchanged.  All stages are HBM-bound byte work at the
// network's input resolution (150 KB per image and pass at 224x224): one thread per output pixel, 3 channels.
//
// Arithmetic (cv2 / albumentations are not installed in this image: these are restatements of their published
// behaviour, "parity unpinned", mirrored one to one by oracle/transforms.py which the tests compare against):
//  * warps (rotate, perspective, scale): source coordinate rounded to 1/32 pixel (cv2's INTER_BITS = 5), 4-tap bilinear
//    in fp32, taps outside the image read 0 (BORDER_CONSTANT), result floor(v + 0.5);
//  * Gaussian blur: cv2's fixed small-kernel taps for sigma = 0, separable, BORDER_REFLECT_101, floor(v + 0.5);
//  * brightness/contrast: albumentations' uint8 look-up table = trunc(clip(v * alpha + beta * 255, 0, 255));
//  * sharpen: 3x3 correlation (cv2.filter2D), BORDER_REFLECT_101, round-half-even and saturate;
//  * Perspective warps into a max_width x max_height rectangle and resizes back (keep_size): here the resize map is
//    folded into the homography, i.e. one interpolation instead of two (a deliberate deviation, documented in DESIGN.md).
#include "common.h"

static_assert(sizeof(MmrcaRotateDesc) == 72, "MmrcaRotateDesc layout (preprocess.py mirrors it)");
static_assert(sizeof(MmrcaAugDesc) == 160, "MmrcaAugDesc layout (preprocess.py mirrors it)");

// The oracle (numpy float32) rounds after every multiply and add; keep the compiler from contracting these into FMAs so
// that coordinates and filter sums agree bit for bit (HIP's __fmul_rn / __fadd_rn are plain operators and would still
// be contracted under the default -ffp-contract=fast).
#pragma clang fp contract(off)
#define MUL(a, b) ((a) * (b))
#define ADD(a, b) ((a) + (b))
__device__ __forceinline__ float affine_at(float m0, float m1, float m2, float x, float y) { return ADD(ADD(MUL(m0, x), MUL(m1, y)), m2); }

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}

// bilinear sample of an interleaved uint8 image at (sx, sy) given in pixels, coordinates quantised to 1/32, zero outside
__device__ __forceinline__ void sample32(const uint8_t* __restrict__ img, int h, int w, float sx, float sy, float (&v)[3]) {
  const int X = (int)rintf(sx * 32.f), Y = (int)rintf(sy * 32.f);
  const int x0 = X >> 5, y0 = Y >> 5;
  const float a = (float)(X & 31) * (1.f / 32.f), b = (float)(Y & 31) * (1.f / 32.f);
  const float w00 = (1.f - a) * (1.f - b), w01 = a * (1.f - b), w10 = (1.f - a) * b, w11 = a * b;
  const bool xi0 = x0 >= 0 && x0 < w, xi1 = x0 + 1 >= 0 && x0 + 1 < w, yi0 = y0 >= 0 && y0 < h, yi1 = y0 + 1 >= 0 && y0 + 1 < h;
  const int64_t base = ((int64_t)y0 * w + x0) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float p00 = (xi0 && yi0) ? (float)img[base + c] : 0.f;
    const float p01 = (xi1 && yi0) ? (float)img[base + 3 + c] : 0.f;
    const float p10 = (xi0 && yi1) ? (float)img[base + (int64_t)w * 3 + c] : 0.f;
    const float p11 = (xi1 && yi1) ? (float)img[base + (int64_t)w * 3 + 3 + c] : 0.f;
    const float s = p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11;
    v[c] = fminf(fmaxf(floorf(s + 0.5f), 0.f), 255.f);
  }
}

__global__ void __launch_bounds__(256)
aug_rotate_k(uint8_t* __restrict__ staging, const MmrcaRotateDesc* __restrict__ desc) {
  const MmrcaRotateDesc d = desc[blockIdx.y];
  if (!d.enabled) return;
  const uint8_t* src = staging + d.src_offset;
  uint8_t* dst = staging + d.dst_offset;
  const int n = d.dh * d.dw;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
    const int y = idx / d.dw, x = idx % d.dw;
    const float fx = (float)(x + d.x_min), fy = (float)(y + d.y_min);
    float v[3];
    sample32(src, d.h, d.w, affine_at(d.inv[0], d.inv[1], d.inv[2], fx, fy), affine_at(d.inv[3], d.inv[4], d.inv[5], fx, fy), v);
    uint8_t* o = dst + (int64_t)idx * 3;
    o[0] = (uint8_t)v[0]; o[1] = (uint8_t)v[1]; o[2] = (uint8_t)v[2];
  }
}

// MODE 0: blur -> flips -> brightness/contrast.  MODE 1: sharpen.
template <int MODE>
__global__ void __launch_bounds__(256)
aug_filter_k(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const MmrcaAugDesc* __restrict__ desc, int H, int W) {
  const MmrcaAugDesc& d = desc[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= H * W) return;
  const int oy = idx / W, ox = idx % W;
  const uint8_t* img = in + (int64_t)blockIdx.y * H * W * 3;
  float v[3];
  if (MODE == 0) {
    // the blur is symmetric, so flipping its output = reading the blurred image at the mirrored position
    const int y = d.flip_v ? H - 1 - oy : oy, x = d.flip_h ? W - 1 - ox : ox;
    const int k = d.blur_k;
    if (k > 0) {
      const int r = k >> 1;
      float acc[3] = {0.f, 0.f, 0.f};
      for (int i = 0; i < k; ++i) {
        const int yy = reflect101(y + i - r, H);
        float row[3] = {0.f, 0.f, 0.f};
        for (int j = 0; j < k; ++j) {
          const uint8_t* p = img + ((int64_t)yy * W + reflect101(x + j - r, W)) * 3;
          const float wj = d.blur[j];
          row[0] += wj * (float)p[0]; row[1] += wj * (float)p[1]; row[2] += wj * (float)p[2];
        }
        acc[0] += d.blur[i] * row[0]; acc[1] += d.blur[i] * row[1]; acc[2] += d.blur[i] * row[2];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = fminf(fmaxf(floorf(acc[c] + 0.5f), 0.f), 255.f);
    } else {
      const uint8_t* p = img + ((int64_t)y * W + x) * 3;
      v[0] = (float)p[0]; v[1] = (float)p[1]; v[2] = (float)p[2];
    }
    if (d.has_bc) {
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = truncf(fminf(fmaxf(ADD(MUL(v[c], d.bc_alpha), d.bc_beta), 0.f), 255.f));
    }
  } else {
    if (d.has_sharp) {
      float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int yy = reflect101(oy + i - 1, H);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const uint8_t* p = img + ((int64_t)yy * W + reflect101(ox + j - 1, W)) * 3;
          const float wt = d.sharp[i * 3 + j];
          acc[0] = ADD(acc[0], MUL(wt, (float)p[0])); acc[1] = ADD(acc[1], MUL(wt, (float)p[1])); acc[2] = ADD(acc[2], MUL(wt, (float)p[2]));
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = fminf(fmaxf(rintf(acc[c]), 0.f), 255.f);
    } else {
      const uint8_t* p = img + ((int64_t)oy * W + ox) * 3;
      v[0] = (float)p[0]; v[1] = (float)p[1]; v[2] = (float)p[2];
    }
  }
  uint8_t* o = out + ((int64_t)blockIdx.y * H * W + idx) * 3;
  o[0] = (uint8_t)v[0]; o[1] = (uint8_t)v[1]; o[2] = (uint8_t)v[2];
}

// FINAL 0: perspective, uint8 out.  FINAL 1: scale, then Normalize + ToTensorV2 (fp32 CHW out).
template <int FINAL>
__global__ void __launch_bounds__(256)
aug_warp_k(const uint8_t* __restrict__ in, uint8_t* __restrict__ out_u8, float* __restrict__ out_f, const MmrcaAugDesc* __restrict__ desc,
           int H, int W, float m0, float m1, float m2, float is0, float is1, float is2) {
  const MmrcaAugDesc& d = desc[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= H * W) return;
  const int y = idx / W, x = idx % W;
  const uint8_t* img = in + (int64_t)blockIdx.y * H * W * 3;
  float v[3];
  const bool on = FINAL ? d.has_scale : d.has_persp;
  if (on) {
    float sx, sy;
    if (FINAL) {
      sx = affine_at(d.scale[0], d.scale[1], d.scale[2], (float)x, (float)y);
      sy = affine_at(d.scale[3], d.scale[4], d.scale[5], (float)x, (float)y);
    } else {
      const float wq = affine_at(d.persp[6], d.persp[7], d.persp[8], (float)x, (float)y);
      const float iw = wq != 0.f ? __fdiv_rn(1.f, wq) : 0.f;
      sx = MUL(affine_at(d.persp[0], d.persp[1], d.persp[2], (float)x, (float)y), iw);
      sy = MUL(affine_at(d.persp[3], d.persp[4], d.persp[5], (float)x, (float)y), iw);
    }
    // keep the fixed-point conversion in range (cv2 saturates): far outside = all four taps outside
    sx = fminf(fmaxf(sx, -16384.f), 16384.f); sy = fminf(fmaxf(sy, -16384.f), 16384.f);
    sample32(img, H, W, sx, sy, v);
  } else {
    const uint8_t* p = img + (int64_t)idx * 3;
    v[0] = (float)p[0]; v[1] = (float)p[1]; v[2] = (float)p[2];
  }
  if (FINAL) {
    const int64_t plane = (int64_t)H * W;
    float* o = out_f + (int64_t)blockIdx.y * 3 * plane + idx;
    o[0] = (v[0] / 255.0f - m0) * is0;
    o[plane] = (v[1] / 255.0f - m1) * is1;
    o[2 * plane] = (v[2] / 255.0f - m2) * is2;
  } else {
    uint8_t* o = out_u8 + ((int64_t)blockIdx.y * H * W + idx) * 3;
    o[0] = (uint8_t)v[0]; o[1] = (uint8_t)v[1]; o[2] = (uint8_t)v[2];
  }
}

extern "C" int mmrca_image_rotate_crop(void* staging, const void* desc, int B, int max_pixels, void* stream) {
  MMRCA_REQUIRE(staging && desc, "image_rotate_crop: null pointer");
  MMRCA_REQUIRE(B > 0 && B <= 65535 && max_pixels > 0, "image_rotate_crop: bad shape B=%d max_pixels=%d", B, max_pixels);
  const unsigned gx = (unsigned)min((max_pixels + 255) / 256, 1024);
  hipLaunchKernelGGL(aug_rotate_k, dim3(gx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, (uint8_t*)staging, (const MmrcaRotateDesc*)desc);
  MMRCA_CHECK_LAUNCH("image_rotate_crop");
  return 0;
}

extern "C" int mmrca_image_augment(void* in_u8, void* tmp_u8, const void* desc, float* out, int B, int H, int W, const float* mean3,
                                   const float* std3, void* stream) {
  MMRCA_REQUIRE(in_u8 && tmp_u8 && desc && out && mean3 && std3, "image_augment: null pointer");
  MMRCA_REQUIRE(B > 0 && H > 1 && W > 1 && B <= 65535 && H <= 8192 && W <= 8192, "image_augment: bad shape B=%d %dx%d", B, H, W);
  MMRCA_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "image_augment: zero std");
  const dim3 grid((unsigned)((H * W + 255) / 256), (unsigned)B);
  hipStream_t st = (hipStream_t)stream;
  uint8_t* a = (uint8_t*)in_u8; uint8_t* b = (uint8_t*)tmp_u8;
  const MmrcaAugDesc* d = (const MmrcaAugDesc*)desc;
  hipLaunchKernelGGL(aug_filter_k<0>, grid, dim3(256), 0, st, a, b, d, H, W);
  hipLaunchKernelGGL(aug_filter_k<1>, grid, dim3(256), 0, st, b, a, d, H, W);
  hipLaunchKernelGGL(aug_warp_k<0>, grid, dim3(256), 0, st, a, b, (float*)nullptr, d, H, W, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
  hipLaunchKernelGGL(aug_warp_k<1>, grid, dim3(256), 0, st, b, (uint8_t*)nullptr, out, d, H, W, mean3[0], mean3[1], mean3[2], 1.f / std3[0],
                     1.f / std3[1], 1.f / std3[2]);
  MMRCA_CHECK_LAUNCH("image_augment");
  return 0;
}
