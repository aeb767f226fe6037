// K7 (the step BEFORE the hot path): the reference's validation image pipeline on the GPU.
//   PadToMaintainAR (keep_aspect_ratio.py:18-53)  ->  A.Resize(W, H, cv2.INTER_LINEAR)  ->  (V/H flip)  ->
//   A.Normalize(ImageNet mean/std)  ->  ToTensorV2                                         (main_both.py:407-440)
// One launch for a whole batch of decoded uint8 HWC images of DIFFERENT sizes: they sit back to back in one staging buffer
// (one pinned H2D copy), a descriptor per image says where and how it is padded; a thread produces one output pixel (3
// channels) from 4 taps.  The padded image is never materialised (taps that fall into the padding read 0).  HBM-bound:
// 3 B read per source pixel (gathered, mostly L2-resident after first touch) and 12 B written per output pixel.
// Resize arithmetic = float32 INTER_LINEAR with half-pixel centres and a replicated border, rounded to uint8 (what cv2
// returns for a uint8 image; cv2 itself uses 11-bit fixed-point weights, i.e. single pixels may differ by one step).
#include "common.h"

static_assert(sizeof(MmrcaImageDesc) == 40, "MmrcaImageDesc layout (preprocess.py::DESC_DTYPE mirrors it)");

template <bool U8>          // U8: uint8 [B,H,W,3] output for the augmentation stages (augment.hip), no normalisation
__global__ void __launch_bounds__(256)
image_preprocess_k(const uint8_t* __restrict__ staging, const MmrcaImageDesc* __restrict__ desc, float* __restrict__ out, uint8_t* __restrict__ out_u8,
                   int out_h, int out_w, float m0, float m1, float m2, float is0, float is1, float is2) {
  const MmrcaImageDesc d = desc[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= out_h * out_w) return;
  int y = idx / out_w, x = idx % out_w;
  const int oy = y, ox = x;
  if (d.flip_v) y = out_h - 1 - y;
  if (d.flip_h) x = out_w - 1 - x;
  const float sx = ((float)x + 0.5f) * ((float)d.pw / (float)out_w) - 0.5f;
  const float sy = ((float)y + 0.5f) * ((float)d.ph / (float)out_h) - 0.5f;
  const float x0f = floorf(sx), y0f = floorf(sy);
  const float fx = sx - x0f, fy = sy - y0f;
  int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  x0 = min(max(x0, 0), d.pw - 1); x1 = min(max(x1, 0), d.pw - 1);
  y0 = min(max(y0, 0), d.ph - 1); y1 = min(max(y1, 0), d.ph - 1);
  const uint8_t* src = staging + d.offset;
  float v[3];
  auto tap = [&](int py, int px, int c) -> float {
    const int sy_ = py - d.pad_top, sx_ = px - d.pad_left;
    if (sy_ < 0 || sy_ >= d.h || sx_ < 0 || sx_ >= d.w) return 0.f;
    return (float)src[((int64_t)sy_ * d.w + sx_) * 3 + c];
  };
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float top = tap(y0, x0, c) * (1.f - fx) + tap(y0, x1, c) * fx;
    const float bot = tap(y1, x0, c) * (1.f - fx) + tap(y1, x1, c) * fx;
    const float val = top * (1.f - fy) + bot * fy;
    v[c] = fminf(fmaxf(floorf(val + 0.5f), 0.f), 255.f);          // uint8 rounding of the resized image
  }
  if (U8) {
    uint8_t* q = out_u8 + ((int64_t)blockIdx.y * out_h * out_w + (int64_t)oy * out_w + ox) * 3;
    q[0] = (uint8_t)v[0]; q[1] = (uint8_t)v[1]; q[2] = (uint8_t)v[2];
    return;
  }
  const int64_t plane = (int64_t)out_h * out_w;
  float* o = out + (int64_t)blockIdx.y * 3 * plane + (int64_t)oy * out_w + ox;
  o[0] = (v[0] / 255.0f - m0) * is0;
  o[plane] = (v[1] / 255.0f - m1) * is1;
  o[2 * plane] = (v[2] / 255.0f - m2) * is2;
}

extern "C" int mmrca_image_preprocess(const void* staging, const void* desc, float* out, int B, int out_h, int out_w,
                                      const float* mean3, const float* std3, void* stream) {
  MMRCA_REQUIRE(staging && desc && out && mean3 && std3, "image_preprocess: null pointer");
  MMRCA_REQUIRE(B > 0 && out_h > 0 && out_w > 0 && B <= 65535, "image_preprocess: bad shape B=%d %dx%d", B, out_h, out_w);
  MMRCA_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "image_preprocess: zero std");
  dim3 grid((unsigned)((out_h * out_w + 255) / 256), (unsigned)B);
  hipLaunchKernelGGL(image_preprocess_k<false>, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)staging,
                     (const MmrcaImageDesc*)desc, out, (uint8_t*)nullptr, out_h, out_w, mean3[0], mean3[1], mean3[2], 1.f / std3[0], 1.f / std3[1],
                     1.f / std3[2]);
  MMRCA_CHECK_LAUNCH("image_preprocess");
  return 0;
}

extern "C" int mmrca_image_resize_u8(const void* staging, const void* desc, void* out_u8, int B, int out_h, int out_w, void* stream) {
  MMRCA_REQUIRE(staging && desc && out_u8, "image_resize_u8: null pointer");
  MMRCA_REQUIRE(B > 0 && out_h > 0 && out_w > 0 && B <= 65535, "image_resize_u8: bad shape B=%d %dx%d", B, out_h, out_w);
  dim3 grid((unsigned)((out_h * out_w + 255) / 256), (unsigned)B);
  hipLaunchKernelGGL(image_preprocess_k<true>, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)staging,
                     (const MmrcaImageDesc*)desc, (float*)nullptr, (uint8_t*)out_u8, out_h, out_w, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f);
  MMRCA_CHECK_LAUNCH("image_resize_u8");
  return 0;
}
