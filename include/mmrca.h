/*
 * mmrca.h -- C ABI of libmmrca.so: the MI355X (gfx950) kernels of the MM-RCA training hot path.
 *
 * The reference (espiriki/Garbage_Classification_RCA) has no FFI of its own: its seam is the Python
 * nn.Module contract of MM_RCA.  Each entry point below names the reference code whose arithmetic it
 * replaces (paths relative to the reference checkout).  Python binds these with ctypes (see
 * INTEGRATION.md); nothing in a signature is a torch type.
 *
 * Conventions
 *   - every function returns 0 on success, a negative code on error; mmrca_last_error() gives the text
 *     (thread-local).  Arguments are validated on the host before any launch.
 *   - the caller owns all buffers; the library allocates nothing and never synchronises the device.
 *     All work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *   - tensors are row-major and contiguous unless a leading dimension is given (in elements).
 *   - dtype: MMRCA_F32 = fp32 storage and math ("parity mode"); MMRCA_BF16 = bf16 storage, fp32
 *     accumulation and fp32 softmax/LayerNorm statistics (the benchmarked mode).
 *   - parameter gradients are always fp32 and are ACCUMULATED (+=) into the caller's gradient arena,
 *     which is what main_both.py:112-124 relies on for its sum-not-mean gradient accumulation.
 *   - NO communication entry points, by decision (SURVEY section 8(b) sketched mmrca_comm_init / mmrca_allreduce /
 *     mmrca_comm_destroy): the one exchange step of the path -- the data-parallel gradient average that replaces
 *     nn.DataParallel, main_both.py:386-388 -- runs on RCCL through torch.distributed (backend "nccl"), one process per GPU,
 *     directly over contiguous slices of the same flat fp32 gradient arena these kernels accumulate into
 *     (garbage_classification_rca_amd/distributed.py::GradSync: asynchronous all-reduce per finished parameter group, fp32 or
 *     bf16 on the wire).  A second, library-private RCCL communicator would duplicate the rendezvous torch already made and buy
 *     no bytes or launches; tests/test_rccl_gpu.py executes that path on the real library.
 */
#ifndef MMRCA_H
#define MMRCA_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { MMRCA_F32 = 0, MMRCA_BF16 = 1 };
enum { MMRCA_ACT_NONE = 0, MMRCA_ACT_GELU = 1,
       MMRCA_ACT_GELU_BWD = 2,        /* C = (A.B) * gelu'(preact); preact (the saved pre-activation) is an INPUT */
       MMRCA_ACT_GELU_SAVE_GRAD = 3,  /* C = gelu(v), preact <- gelu'(v): the forward emits the derivative (shares the erf) */
       MMRCA_ACT_MUL = 4,             /* C = (A.B) * preact; preact is an INPUT (pairs with GELU_SAVE_GRAD) */
       MMRCA_ACT_GELU_SAVE_GRAD_BF16 = 5 }; /* mmrca_gemm_x3 only: GELU_SAVE_GRAD with gelu' stored as bf16 (not fp32) -- the form the bf16
                                              * backward of the bf16x3f mode reads */
/* operand layouts of mmrca_gemm: ROWK = [rows][contraction] (contraction contiguous),
 * KROW = [contraction][rows] (rows contiguous) */
enum { MMRCA_ROWK = 0, MMRCA_KROW = 1 };
enum { MMRCA_GEMM_AUTO = 0, MMRCA_GEMM_REF = 1, MMRCA_GEMM_MFMA = 2 /* 128x128 tiles */,
       MMRCA_GEMM_MFMA256 = 3 /* 256x256 tiles, persistent workgroups; A row-major, M % 256 == 0 (a ragged M only through
                                  mmrca_gemm_rows, which checks that round_up(M, 256) rows are readable; AUTO only picks the
                                  kernel when M % 256 == 0), N % 256 == 0, K % 64 == 0, K >= 128, operands < 4 GiB */,
       MMRCA_GEMM_MFMA_PERSIST = 4 /* reserved: experimental kernel of round 1, removed (rejected with an error) */,
       MMRCA_GEMM_MFMA_BK32 = 5 /* 128x128x32 tiles, 32 KiB LDS: four blocks per CU */,
       MMRCA_GEMM_MFMA_1STAGE = 6 /* 128x128x64 tiles, single LDS stage (32 KiB): four blocks per CU */,
       MMRCA_GEMM_MFMA_TALL = 7 /* reserved: removed */,
       MMRCA_GEMM_MFMA_256W = 8 /* reserved: removed */,
       MMRCA_GEMM_MFMA_256X4 = 9 /* reserved: removed */ };

const char* mmrca_last_error(void);
int mmrca_version(void);

/* Step epoch of the counter-based dropout masks.  Mask seeds are launch arguments, so the launches of a step captured in a HIP graph
 * (the small-batch path of training.GraphedTrainStep; the reference has no counterpart -- main_both.py:81-134 launches every op from
 * Python each step) would redraw the captured step's masks at every replay.  Every mask hash runs on seed + epoch * 1000003; this
 * call sets the epoch of all kernels from *device_value (read when the launch EXECUTES: captured once, it follows a counter in HBM
 * from replay to replay) or, with device_value == NULL, from `value`.  0 (the initial state) outside graph replays. */
int mmrca_seed_epoch_set(const uint64_t* device_value, uint64_t value, void* stream);
/* Stochastic depth of the conv image encoders (torchvision.ops.StochasticDepth(p, "row") inside efficientnet_v2_*'s blocks, which the
 * reference builds at multimodal_model.py:113-126): out[i*B + b] = (u >= p[i]) / (1 - p[i]) for n residual blocks and B samples, u the
 * counter-based uniform of (seed + epoch, i*B + b).  p, out: fp32 in HBM; p[i] < 1. */
int mmrca_sd_rowscale(const float* p, float* out, int n, int B, uint64_t seed, void* stream);
/* timing-only ablation switches for kernel development (0 = normal operation; results are WRONG otherwise) */
int mmrca_debug_set(int flags);
/* opt-in flat (streaming) forms of the BatchNorm column reductions: bit 0 = statistics / backward sums, bit 1 = one-pass moments.
 * Initial value: MMRCA_BN_FLAT / MMRCA_BN_FLAT_MOMENTS read once at load; 0 (off) in production.  Needs the `ws` workspace of the
 * mmrca_bn_*_ws entry points. */
int mmrca_bn_flat_set(int mode);
/* diagnostic: a device buffer of B*H*4 uint64 that the fused ViT attention backward fills with s_memtime stamps of its phases
 * (entry, staged, dQ done, exit; tools/attn_stamps.py); NULL (the default) = no stamps */
int mmrca_debug_attn_stamps(void* buf);

/* K2. C[M,N] = act(A (.) B + bias[N]) + addend[M,N]      (torch.nn.Linear fwd / dgrad / wgrad:
 * transformers modeling_distilbert.py q_lin/k_lin/v_lin/out_lin/ffn.lin1/lin2, torchvision ViT in_proj/
 * out_proj/mlp, and their autograd; head linears go through mmrca_head_*).
 *   A: M rows, contraction K; layout a_layout, leading dimension lda.   B: N rows, contraction K.
 *   out_f32_accum != 0: C is fp32 and C += result (wgrad); otherwise C has `dtype`.  In that mode `bias`, if given,
 *   is an fp32 OUTPUT [M]: bias[m] += sum_k A[m][k] (the bias gradient = column sums of dY), fused into the same
 *   pass over A (A must be KROW).
 *   preact (optional, `dtype`): receives A(.)B + bias before the activation.
 *   For a_layout==KROW the contraction runs over rows of A and B; rows K..round_up(K,64) of both
 *   buffers must exist and hold zeros (the engine pads its activation buffers so).
 *   impl: AUTO picks the MFMA kernel when dtype==BF16 and the shape qualifies. */
int mmrca_gemm(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact,
               int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
               int a_layout, int b_layout, int act, int out_f32_accum, int dtype, int impl, void* stream);
/* The same product (no accumulate mode) that also adds the column sums of the stored C to colsum[N] (fp32, +=): when C is an
 * input gradient dX = dY W, these are the bias gradient of the layer that produced X (with act = MMRCA_ACT_MUL and
 * preact = gelu'(h): the FFN1 bias gradient, so GELU backward and its bias reduction need no pass of their own). */
int mmrca_gemm_colsum(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact,
                      float* colsum, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                      int a_layout, int b_layout, int act, int dtype, int impl, void* stream);

/* C[M,N] = A[M,K] . B[N,K]^T (bf16, no epilogue) with the BatchNorm moments of the stored C collected in the epilogue (the 1x1
 * convolutions of the conv backbones, multimodal_model.py:113-126): s1[tm, n] / s2[tm, n] ([mmrca_gemm_bnstats_slots(M), N] fp32,
 * written) = sum / sum of squares of (c - shift[n]) over the valid rows of 128-row block tm; shift (the layer's running mean, or
 * NULL = 0) keeps the variance free of cancellation.  mmrca_bn_finish_sums turns them into mean / rstd / running statistics exactly
 * as mmrca_bn_stats(train = 1) would from C.  Returns -3 for shapes the 128x128 bf16 kernels do not take (N % 8, K % 32, M >= 64). */
int64_t mmrca_gemm_bnstats_slots(int64_t M);
int mmrca_gemm_bnstats(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                       int dtype, const float* shift, float* s1, float* s2, void* stream);
int mmrca_bn_finish_sums(const float* s1, const float* s2, const float* shift, int64_t nslots, int64_t rows, float* mean, float* rstd,
                         float* running_mean, float* running_var, int C, float eps, float momentum, void* stream);
/* mmrca_gemm (colsum == NULL) / mmrca_gemm_colsum with the row contract of MMRCA_GEMM_MFMA256 made explicit.  That kernel
 * streams whole 256-row tiles: with M % 256 != 0 it READS (never stores) rows M .. round_up(M, 256) - 1 of A and of the side
 * operand (addend, or preact under MMRCA_ACT_MUL).  mmrca_gemm / mmrca_gemm_colsum therefore reject impl = MMRCA_GEMM_MFMA256
 * with a ragged M (-4); here the caller states how many rows of each exist, and a request they do not cover is rejected
 * before any launch.  (AUTO needs none of this: it only picks the kernel for whole tiles.)  The other precondition of the
 * tiled kernels cannot be checked from pointers: a KROW operand contracts over its ROW index in 64-row steps, so K must be a
 * multiple of 64 and a caller that rounds its row count up owns zero-filled rows up to that K (wrong sums otherwise, no fault). */
int mmrca_gemm_rows(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact,
                    float* colsum, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                    int64_t a_rows_readable, int64_t side_rows_readable, int a_layout, int b_layout, int act, int dtype,
                    int impl, void* stream);

/* Stream-K tail of the persistent 256x256 kernel (round 6).  A launch of T tiles on G CUs is floor(T / G) whole rounds plus
 * L = T mod G leftover tiles; without a workspace the leftover tiles cost a whole extra round on L CUs (or, from AUTO, a second
 * launch on the 128x128 kernel).  With a workspace registered for the stream of the call, the K loop of every leftover tile is
 * cut into min(G / L, K steps / 2, 4) ranges that run as the LAST work item of as many workgroups of the same launch; each leaves
 * its fp32 partial tile in the workspace and the workgroup that arrives last at the tile's counter adds the partials in the fixed
 * order 0, 1, ... (bitwise reproducible) and runs the epilogue.  No workgroup waits for another.
 *   workspace: caller-owned device memory, >= mmrca_gemm_streamk_workspace_bytes() bytes (4 KiB of counters + 256 partial tiles),
 *   on the CURRENT device, 16-byte aligned, ZERO-FILLED by the caller once (every launch leaves the counters at zero again); one
 *   per (device, stream) that issues
 *   GEMMs concurrently with another; it must outlive every launch (and captured graph) made on that stream.  workspace == NULL
 *   removes the stream's entry.  Without a workspace there is no tail (the Python host registers one per stream; MMRCA_SK=0 stops it).
 * mmrca_gemm_streamk_config: at most `max_split` ranges per tile (default 4, MMRCA_SK_MAX); no tail below `min_ksteps` K steps
 *   (of 64 columns; of 32 in the fused bf16x3 form; default 24, MMRCA_SK_MIN_KSTEPS: the partial tiles cost 2 x 256 KiB of traffic
 *   per range whatever K is); bf16_products: 1 = also for plain bf16 products (default 0, MMRCA_SK_BF16: measured neutral at
 *   K >= 1536 and slower below, profiles/r06_streamk_ab.txt -- the fused bf16x3 products, three times the matrix work per tile,
 *   gain: +1.3 % on the bf16x3f step).  An argument < 0 keeps the current value. */
int64_t mmrca_gemm_streamk_workspace_bytes(void);
int mmrca_gemm_streamk_workspace(void* workspace, int64_t bytes, void* stream);
int mmrca_gemm_streamk_config(int max_split, int min_ksteps, int bf16_products);

/* Weight gradient on 256x256 tiles with the contraction split over workgroups: C[M,N] (fp32) += A (.) B over K
 * (torch autograd of nn.Linear: dW = dY^T X; same call sites as mmrca_gemm's accumulate mode).  bf16 operands, either
 * layout; M % 256 == 0, N % 256 == 0, K % 64 == 0, at most 256 output tiles -- or, with both operands K-major (MMRCA_KROW: the
 * weight-gradient layout), any M, N that are multiples of 8 (the conv layers: 176 x 1056, 48 x 192, ...): the edge tiles then
 * re-read column 0 in place of the columns past the operands' edge (nothing outside [K, M] / [K, N] is read) and the second launch
 * stores only what lies inside C.  The per-workgroup fp32 partial tiles go
 * through `workspace` (caller-owned, >= mmrca_gemm_splitk_workspace_bytes(M, N) bytes, one per concurrently used
 * stream) with plain stores and a second launch adds them into C -- no atomics, bitwise reproducible. */
int64_t mmrca_gemm_splitk_workspace_bytes(int64_t M, int64_t N);
int mmrca_gemm_splitk(const void* A, const void* B, float* C, void* workspace, int64_t workspace_bytes, int64_t M,
                      int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_layout, int b_layout, void* stream);

/* K2x. The same products with fp32 ACCURACY on the bf16 matrix cores ("bf16x3", the fast mode that meets the reference's
 * precision: the reference computes every nn.Linear in fp32, CVPR_code/multimodal_model.py:651-726 with no autocast anywhere).
 * An fp32 operand x is carried as two bf16 planes, hi = bf16(x), lo = bf16(x - hi) (mmrca_split_f32; x = hi + lo to 2^-17),
 * and A.B = A_hi.B_hi + A_lo.B_hi + A_hi.B_lo with fp32 accumulation (A_lo.B_lo is below fp32 resolution) -- one pass of the
 * bf16 MFMA kernels over the virtual contraction [A_hi|A_lo|A_hi].[B_hi|B_hi|B_lo] of length 3K, no operand copies.
 *   A_hi/A_lo, B_hi/B_lo: bf16 planes with the layouts / leading dimensions of mmrca_gemm.
 *   C: fp32 [M, ldc]; or, with C_lo != NULL, the result is stored as two bf16 planes (C = hi plane, C_lo = lo plane) for a
 *   consumer that is another bf16x3 product.  bias [N], addend [M, ldc], preact [M, ldc]: fp32, same roles as in mmrca_gemm
 *   (MMRCA_ACT_GELU_SAVE_GRAD writes gelu' to preact, MMRCA_ACT_MUL multiplies by it).  colsum (optional, fp32 [N], +=):
 *   column sums of the stored C.  out_f32_accum != 0: C (fp32) += result, no epilogue (prefer mmrca_gemm_splitk_x3).
 *   Shapes: N % 128 == 0, K % 64 == 0, M % 128 == 0 for a KROW A; 16-byte aligned operands.
 *   impl: MMRCA_GEMM_AUTO | MMRCA_GEMM_MFMA_1STAGE (128x128 tiles, any epilogue) | MMRCA_GEMM_MFMA256 (persistent 256x256
 *   tiles: row-major A with M % 256 == 0 -- REJECTED otherwise, the kernel streams whole 256-row tiles --, N % 256 == 0,
 *   K >= 128, bias-only or GELU_SAVE_GRAD epilogue, operands < 4 GiB). */
int mmrca_gemm_x3(const void* A_hi, const void* A_lo, const void* B_hi, const void* B_lo, void* C, void* C_lo,
                  const void* bias, const void* addend, void* preact, float* colsum, int64_t M, int64_t N, int64_t K,
                  int64_t lda, int64_t ldb, int64_t ldc, int a_layout, int b_layout, int act, int out_f32_accum,
                  int impl, void* stream);
/* weight gradient in that form: C[M,N] (fp32) += A (.) B over K; the contracts of mmrca_gemm_splitk (workspace included) */
int mmrca_gemm_splitk_x3(const void* A_hi, const void* A_lo, const void* B_hi, const void* B_lo, float* C, void* workspace,
                         int64_t workspace_bytes, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                         int a_layout, int b_layout, void* stream);
/* hi[i] = bf16(src[i]) (round to nearest even), lo[i] = bf16(src[i] - hi[i]); n % 4 == 0 */
int mmrca_split_f32(const float* src, void* hi, void* lo, int64_t n, void* stream);

/* db[N] (fp32) += column sums of dY[M,N]  (bias gradients of every nn.Linear on the path). */
int mmrca_colsum_accum(const void* dY, float* db, int64_t M, int64_t N, int64_t ld, int dtype, void* stream);

/* dH = dG * gelu'(H)   (exact erf GELU: transformers activations "gelu", torchvision MLPBlock nn.GELU). */
int mmrca_gelu_bwd(const void* dG, const void* H, void* dH, int64_t n, int dtype, void* stream);
/* the same on a [M,N] matrix plus db[N] (fp32) += column sums of dH: the FFN1 / mlp.0 bias gradient in the same pass */
int mmrca_gelu_bwd_colsum(const void* dG, const void* H, void* dH, float* db, int64_t M, int64_t N, int64_t ld,
                          int dtype, void* stream);

/* ---- conv image backbones: torchvision efficientnet_v2_m / _l as instantiated at
 * CVPR_code/multimodal_model.py:11-36, 113-126 and shufflenet_v2_x2_0 (models.py:261-278), on NHWC activations stored as
 * row-major [B*H*W, C] matrices.  1x1 convolutions are mmrca_gemm on those rows; a full 3x3 convolution is
 * mmrca_im2row3x3 + mmrca_gemm (patch row = col[c*9 + ky*3 + kx], the flattening of torchvision's [Cout, Cin, 3, 3] weight).
 * CONV activations: 0 none, 1 SiLU, 2 ReLU, 3 sigmoid. ------------------------------------------------------------- */
/* images [B, C, H, W] fp32 (the tensor main_both.py:100-103 moves to the device) -> rows [B*H*W, C] in `dtype` */
int mmrca_nchw_to_rows(const float* images, void* x, int B, int C, int H, int W, int dtype, void* stream);
/* 3x3 patches, padding 1, stride 1 | 2: col[B*Ho*Wo, ldk] (ldk >= 9C), and the gather that sums a patch gradient back */
int mmrca_im2row3x3(const void* x, void* col, int B, int H, int W, int C, int stride, int64_t ldk, int dtype, void* stream);
int mmrca_col2im3x3(const void* dcol, void* dx, int B, int H, int W, int C, int stride, int64_t ldk, int dtype, void* stream);
/* The same pair with TAP-MAJOR patches, k = tap*C + c (bf16, C % 8 == 0, 16-byte aligned): every access a contiguous 16-byte
 * vector.  The GEMM on these patches takes the convolution weight permuted from torchvision's [C_out, C_in, 3, 3] to
 * [C_out, 9, C_in] (conv_engine.py keeps that copy); same convolution (multimodal_model.py:11-36 -> torchvision Conv2d 3x3). */
int mmrca_im2row3x3_tap(const void* x, void* col, int B, int H, int W, int C, int stride, int64_t ldk, int dtype, void* stream);
int mmrca_col2im3x3_tap(const void* dcol, void* dx, int B, int H, int W, int C, int stride, int64_t ldk, int dtype, void* stream);
/* Dense 3x3 / stride 1 / padding 1 convolution WITHOUT a patch matrix (conv_igemm.hip; bf16, NHWC rows, tap-major weights
 * w_tap[Cout, 9*Cp], column tap*Cp + ci with tap = 3*ky + kx and Cp = Cin rounded up to a multiple of 32 (zero pad columns);
 * Cin % 8 == 0, Cout % 8 == 0, 16-byte aligned operands).
 * Replaces mmrca_im2row3x3_tap + mmrca_gemm for torchvision's Conv2dNormActivation 3x3 inside FusedMBConv
 * (multimodal_model.py:113-126 -> models.efficientnet_v2_*).
 *   fwd:   z[B*H*W, Cout] = conv(x[B*H*W, Cin], w_tap).  With part_mean / part_m2 / part_cnt given
 *          ([mmrca_conv3x3_stat_slots(B,H,W)] x Cout floats each, and that many counts) the BatchNorm moments of the stored
 *          outputs are left there as per-wave (count, mean, M2) triples; mmrca_conv_bn_finish merges them (Chan) into mean,
 *          rstd and -- momentum > 0 -- the running statistics, exactly as mmrca_bn_stats(train = 1) would from z.
 *          The input gradient of the same convolution is this call on dz with the flipped, transposed weights
 *          w'[Cin, 9*Cout_p], w'[ci, tap', co] = w[co, ci, 8 - tap'].
 *   wgrad: dw_tap[Cout, 9*Cin] (fp32) += dz^T * patches(x); Cin % 8 == 0, Cout % 8 == 0; reads no padding rows. */
int64_t mmrca_conv3x3_stat_slots(int B, int H, int W);
int mmrca_conv3x3_fwd(const void* x, const void* w_tap, void* z, float* part_mean, float* part_m2, float* part_cnt, int B, int H,
                      int W, int Cin, int Cout, int dtype, void* stream);
int mmrca_conv_bn_finish(float* part_mean, float* part_m2, float* part_cnt, int B, int H, int W, float* mean, float* rstd,
                         float* running_mean, float* running_var, int C, float eps, float momentum, void* stream);
int mmrca_conv3x3_wgrad(const void* dz, const void* x, float* dw_tap, int B, int H, int W, int Cin, int Cout, int dtype,
                        void* stream);
/* depthwise 3x3 (torch Conv2d(C, C, 3, stride, 1, groups=C, bias=False)); w [C, 9] in `dtype`; dw fp32 [C, 9] +=; dx / dw may be NULL */
int mmrca_dwconv3x3_fwd(const void* x, const void* w, void* y, int B, int H, int W, int C, int stride, int dtype, void* stream);
int mmrca_dwconv3x3_bwd(const void* dy, const void* x, const void* w, void* dx, float* dw, int B, int H, int W, int C, int stride,
                        int dtype, void* stream);
/* the same with a caller-provided scratch buffer (16-byte aligned, any size; 16 MiB serves every shape of the conv backbones): the
 * stride-1 bf16 weight gradient then writes per-block partial sums there and adds them up in a second kernel instead of issuing
 * device-scope fp32 atomics */
int mmrca_dwconv3x3_bwd_ws(const void* dy, const void* x, const void* w, void* dx, float* dw, int B, int H, int W, int C, int stride,
                           int dtype, void* ws, int64_t ws_bytes, void* stream);
/* torch.nn.BatchNorm2d over the rows of x [rows, C] (ld): train != 0 -> batch mean / biased variance into mean, rstd and, with
 * momentum > 0, the running statistics update (unbiased variance); train == 0 -> mean / rstd from the running statistics */
int mmrca_bn_stats(const void* x, float* mean, float* rstd, float* running_mean, float* running_var, int64_t rows, int C,
                   int64_t ld, float eps, float momentum, int train, int dtype, void* stream);
/* the same with a caller workspace (fp32 words, 16-byte aligned; 16 MiB serves every size, NULL = the call above): with
 * MMRCA_BN_FLAT=1 large bf16 tensors take the FLAT streaming reduction -- every wave instruction reads 1 KiB of consecutive addresses,
 * per-thread partial sums go to the workspace and a second small launch adds them up (3.8-4.3 TB/s against 3.1-3.7 for the
 * slice-per-workgroup form in isolation; no gain inside the conv step, hence opt-in -- csrc/conv.hip) */
/* mmrca_bn_stats(train = 1) with the finish step (sums -> mean / rstd, running statistics) done by the last workgroup of the reduction
 * instead of a second launch (round 5: one launch less per BatchNorm layer and step -- measured slower in the conv step, the engine
 * does not use it by default; kept, tested, for small tensors outside it).  tickets: ceil(C / 64) int32 words of scratch;
 * flags bit 0: the caller zeroed mean, rstd AND tickets.  bf16 with C % 8 == 0; other inputs take the two-launch form. */
int mmrca_bn_stats_fused(const void* x, float* mean, float* rstd, float* running_mean, float* running_var, int64_t rows, int C,
                         int64_t ld, float eps, float momentum, int dtype, int* tickets, int flags, void* stream);
/* flags bit 0: mean / rstd (scratch for the backward) were zeroed by the caller -- the conv engine keeps every layer's statistics and
 * backward sums in ONE arena and clears it with one fill per step instead of two per layer (the small-batch step is bound by its launch count) */
int mmrca_bn_stats_ws(const void* x, float* mean, float* rstd, float* running_mean, float* running_var, int64_t rows, int C,
                      int64_t ld, float eps, float momentum, int train, int dtype, void* ws, int64_t ws_bytes, int flags, void* stream);
int mmrca_bn_act_fwd(const void* x, const float* mean, const float* rstd, const void* gamma, const void* beta, void* y,
                     int64_t rows, int C, int act, int dtype, void* stream);
/* out = res + rowscale[row / rows_per_sample] * act(bn(x)): mmrca_bn_act_fwd and the block's residual connection
 * (mmrca_residual_add, torchvision StochasticDepth "row" scale, NULL = 1) in one pass.  bf16, C % 8 == 0, 16-byte aligned
 * operands; anything else returns -3 and the caller issues the two calls. */
int mmrca_bn_act_fwd_res(const void* x, const float* mean, const float* rstd, const void* gamma, const void* beta, const void* res,
                         const float* rowscale, void* out, int64_t rows, int C, int act, int64_t rows_per_sample, int dtype, void* stream);
/* The train-mode forward in TWO launches instead of three (bf16, C % 8 == 0, 16-byte aligned operands; anything else returns -3 and the
 * caller takes mmrca_bn_stats + mmrca_bn_act_fwd(_res)).  mmrca_bn_moments: the shifted one-pass sums of x[rows, C] into s1 / s2 (fp32
 * [C], +=: the caller zeroes them) and the shift it subtracted into `shift` (fp32 [C], written).  mmrca_bn_act_fwd_fin: y = act(bn(x)) --
 * res != NULL: y = res + rowscale[row / rows_per_sample] * act(bn(x)), rowscale NULL = 1 -- with the finish step inside: mean / rstd from
 * (s1, s2, shift), stored to mean_out / rstd_out (fp32 [C], distinct from s1 / s2; the backward reads them), running statistics
 * updated when momentum > 0 (torch semantics, as mmrca_bn_stats). */
int mmrca_bn_moments(const void* x, float* s1, float* s2, float* shift, int64_t rows, int C, int64_t ld, int dtype, void* stream);
int mmrca_bn_act_fwd_fin(const void* x, const float* s1, const float* s2, const float* shift, const void* gamma, const void* beta,
                         const void* res, const float* rowscale, void* y, float* mean_out, float* rstd_out, float* running_mean,
                         float* running_var, int64_t rows, int C, int act, int64_t rows_per_sample, float eps, float momentum, int dtype,
                         void* stream);
/* backward of y = act(BN(x)); scratch = fp32 [2C]; dx / dgamma+dbeta (fp32, +=) may be NULL; train as in the forward */
int mmrca_bn_act_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const void* gamma, const void* beta,
                     void* dx, float* dgamma, float* dbeta, float* scratch, int64_t rows, int C, int act, int train, int dtype,
                     void* stream);
/* the same with a workspace, as mmrca_bn_stats_ws: the reduce pass over dy and x in the flat streaming form */
int mmrca_bn_act_bwd_ws(const void* dy, const void* x, const float* mean, const float* rstd, const void* gamma, const void* beta,
                        void* dx, float* dgamma, float* dbeta, float* scratch, int64_t rows, int C, int act, int train, int dtype,
                        void* ws, int64_t ws_bytes, int flags, void* stream);
/* the same when sums[0..C) = sum du and sums[C..2C) = sum du * xhat already hold the first pass's result (mmrca_se_dx computes them
 * while it writes dy): no reduce pass over dy and x */
int mmrca_bn_act_bwd_sums(const void* dy, const void* x, const float* mean, const float* rstd, const void* gamma, const void* beta,
                          void* dx, float* dgamma, float* dbeta, float* sums, int64_t rows, int C, int act, int train, int dtype,
                          void* stream);
/* AdaptiveAvgPool2d(1) / x.mean([2,3]) over the HW rows of each sample, and its backward (dx (+)= dpool / HW) */
int mmrca_rowpool_mean(const void* x, void* out, int B, int HW, int C, int dtype, void* stream);
int mmrca_rowpool_mean_bwd(const void* dpool, void* dx, int B, int HW, int C, int accumulate, int dtype, void* stream);
/* squeeze-excitation scaling y = x * s[b, c] and its backward (dx = dy * s, ds[b, c] = sum_rows dy * x) */
/* The squeeze-excitation MLP of one MBConv block in one launch (torchvision SqueezeExcitation: fc1 [sq, c] + bias, SiLU, fc2 [c, sq] +
 * bias, sigmoid; efficientnet_v2_* as built at CVPR_code/multimodal_model.py:113-126): pooled [B, c] -> h_pre, h [B, sq] -> s_pre, s [B, c]
 * (pre-activations stored WITHOUT bias, as mmrca_bias_act_fwd expects).  One workgroup per sample; (c + sq) * 4 bytes of LDS.
 * c and sq multiples of 4; every pointer (both entry points; not the fp32 gradient accumulators) aligned to 4 elements -- 8 bytes in
 * bf16, 16 in fp32 -- or the call is refused (-1). */
int mmrca_se_mlp_fwd(const void* pooled, const void* w1, const void* b1, const void* w2, const void* b2, void* h_pre, void* h,
                     void* s_pre, void* s, int B, int c, int sq, int dtype, void* stream);
/* its backward from ds = d loss / d s [B, c]: writes dpool [B, c] (and the workspaces ds_pre [B, c], dh_pre [B, sq]) and ADDS the
 * parameter gradients to the fp32 gw1 [sq, c], gb1 [sq], gw2 [c, sq], gb2 [c] (two launches: per-sample chain, batch sums). */
int mmrca_se_mlp_bwd(const void* ds, const void* pooled, const void* h_pre, const void* h, const void* s_pre, const void* w1,
                     const void* b1, const void* w2, const void* b2, void* ds_pre, void* dh_pre, void* dpool, float* gw1,
                     float* gb1, float* gw2, float* gb2, int B, int c, int sq, int dtype, void* stream);
int mmrca_se_scale_fwd(const void* x, const void* s, void* y, int B, int HW, int C, int dtype, void* stream);
int mmrca_se_scale_bwd(const void* dy, const void* x, const void* s, void* dx, void* ds, int B, int HW, int C, int dtype, void* stream);
/* Second half of the squeeze-excitation backward in one pass: dx = dy * s[b, c] + dpool[b, c] / HW (mmrca_se_scale_bwd with
 * dx == NULL computes only ds first).  With z / mean / rstd / gamma / beta / sums given, the first-pass sums of the BatchNorm +
 * activation whose output gradient dx is are accumulated on the way (for mmrca_bn_act_bwd_sums).  bf16, C % 8 == 0, aligned: -3 otherwise. */
int mmrca_se_dx(const void* dy, const void* s, const void* dpool, void* dx, int B, int HW, int C, int dtype, const void* z,
                const float* mean, const float* rstd, const void* gamma, const void* beta, int act, float* sums, void* stream);
/* y = act(x + bias[c]) on small [rows, C] matrices (the biased 1x1 convolutions of squeeze-excitation) and its backward */
int mmrca_bias_act_fwd(const void* x, const void* bias, void* y, int64_t rows, int C, int act, int dtype, void* stream);
int mmrca_bias_act_bwd(const void* dy, const void* x, const void* bias, void* dx, float* dbias, int64_t rows, int C, int act,
                       int dtype, void* stream);
/* out = a + branch * rowscale[sample] (residual + torchvision "row" stochastic depth); a NULL -> scaled branch alone */
int mmrca_residual_add(const void* a, const void* branch, const float* rowscale, void* out, int B, int64_t per_sample, int dtype,
                       void* stream);
/* MaxPool2d(3, 2, 1) with the argmax tap (uint8) kept for the backward */
int mmrca_maxpool3x3s2_fwd(const void* x, void* y, void* argmax, int B, int H, int W, int C, int dtype, void* stream);
int mmrca_maxpool3x3s2_bwd(const void* dy, const void* argmax, void* dx, int B, int H, int W, int C, int dtype, void* stream);
/* torch channel_shuffle(cat(a[:, :bf], b), groups = 2) (ShuffleNetV2: torchvision shufflenetv2.py channel_shuffle after the unit's concat) in
 * one launch: out[r, 2j] = a[r, j], out[r, 2j + 1] = b[r, j]; a has row pitch lda.  And its backward: d_even[r, j] = dout[r, 2j] (row pitch
 * ld_even), d_odd[r, j] = dout[r, 2j + 1]. */
int mmrca_channel_interleave2(const void* a, int64_t lda, const void* b, void* out, int64_t rows, int bf, int dtype, void* stream);
int mmrca_channel_deinterleave2(const void* dout, void* d_even, int64_t ld_even, void* d_odd, int64_t rows, int bf, int dtype, void* stream);
/* out[r, col0 + j] = in[r, map[j]] (channel split / concat / shuffle and their backward) */
int mmrca_channel_gather(const void* in, const int* map, void* out, int64_t rows, int Cin, int Cout, int64_t ld_out, int col0,
                         int dtype, void* stream);

/* K3. Multi-head attention over a fused QKV buffer [rows, 3*H*dh] (q | k | v column blocks; head h at
 * columns h*dh).  out[rows, H*dh].  key_mask (optional): int32 per token row, 0 = masked key; a query row whose
 * keys are all masked yields zeros (torch SDPA semantics used by transformers 5.x).  lse: fp32 [B,H,S].
 * Replaces modeling_distilbert.py:122-203 / torchvision MultiheadAttention (QK^T*scale, softmax, PV).
 * drop_p > 0: attention-probability dropout (modeling_distilbert.py:146), mask from (seed, ((b*H+h)*S+q)*S+key).
 * Token layout: cu_seqlens == NULL -> padded, sequence b = rows [b*S, (b+1)*S).  cu_seqlens = int32 [B+1] (device)
 * -> packed: sequence b = rows [cu[b], cu[b+1]) (captions stored back to back WITHOUT their padding, lengths <= S);
 * S then only sizes the kernels and strides the lse / dropout-counter index spaces, which stay those of the padded
 * layout, so a packed run reproduces the padded run on every kept row. */
int mmrca_mha_fwd(const void* qkv, const int32_t* key_mask, void* out, float* lse,
                  int B, int H, int S, int dh, float scale, float drop_p, uint64_t drop_seed,
                  const int32_t* cu_seqlens, int dtype, int impl, void* stream);
int mmrca_mha_bwd(const void* qkv, const int32_t* key_mask, const void* out, const void* dout, const float* lse,
                  void* dqkv, int B, int H, int S, int dh, float scale, float drop_p, uint64_t drop_seed,
                  const int32_t* cu_seqlens, int dtype, int impl, void* stream);
/* mmrca_mha_fwd on fp32 operands that ALSO writes the context as two bf16 planes out_hi + out_lo (the operand form of the
 * bf16x3 out-projection GEMM, mmrca_gemm_x3).  Head dim 64, 1 <= S <= 208 (the fp32-matrix-core kernels of attention_f32.hip);
 * rejected otherwise -- use mmrca_mha_fwd + mmrca_split_f32 there. */
int mmrca_mha_fwd_planes(const void* qkv, const int32_t* key_mask, void* out, void* out_hi, void* out_lo, float* lse,
                         int B, int H, int S, int dh, float scale, float drop_p, uint64_t drop_seed,
                         const int32_t* cu_seqlens, void* stream);
/* The same with q|k|v given as the two bf16 planes a bf16x3 GEMM wrote (qkv_hi + qkv_lo; mmrca_gemm_x3 with C_lo != NULL): the
 * fp32 values are rebuilt on load.  The bf16x3f mode uses it so that its bf16 backward can read qkv_hi as the bf16 q|k|v.
 * out (the fp32 context) may be NULL: only the planes are written. */
int mmrca_mha_fwd_planes_in(const void* qkv_hi, const void* qkv_lo, const int32_t* key_mask, void* out, void* out_hi, void* out_lo,
                            float* lse, int B, int H, int S, int dh, float scale, float drop_p, uint64_t drop_seed,
                            const int32_t* cu_seqlens, void* stream);
/* The bf16x3 attention forward: q|k|v and the context as two bf16 planes each, every product (QK^T, PV) three-pass on the bf16 matrix
 * cores with P split in registers, fp32 softmax statistics -- the fp32 arithmetic of the kernels above to ~1e-6 at the bf16 MFMA
 * rate.  Head dim 64, 1 <= S <= 224; mask / packed layout / dropout / lse as mmrca_mha_fwd.  The forward of the bf16x3f mode
 * (transformers modeling_distilbert.py:122-203 / torchvision MultiheadAttention under CVPR_code/multimodal_model.py:651-659). */
int mmrca_mha_fwd_x3(const void* qkv_hi, const void* qkv_lo, const int32_t* key_mask, void* out_hi, void* out_lo, float* lse,
                     int B, int H, int S, int dh, float scale, float drop_p, uint64_t drop_seed, const int32_t* cu_seqlens, void* stream);
/* The bf16x3 form of mmrca_mha_cross_fwd: q / k / v are fp32 (separate operands, row strides in elements), every product is
 * three-pass on the bf16 matrix cores with fp32 softmax statistics, the context is written as two bf16 planes out_hi + out_lo (the
 * operand form of mmrca_gemm_x3).  Any head dim % 8 == 0 <= 128, any S_kv (keys are walked in chunks with running softmax
 * statistics), S_q <= 384, attention-probability dropout.  The fp32 arithmetic of Blip2Attention (ViT-g: 257 tokens, 16 heads of 88,
 * transformers modeling_blip_2.py:282-354) and Blip2QFormerMultiHeadAttention (:536-606) under q_former_training.py:279-304, for the
 * compliant (<= 1e-3) mode of BASELINE configs[4]. */
int mmrca_mha_cross_fwd_x3(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, void* out_hi,
                           void* out_lo, int64_t ldo, int B, int H, int Sq, int Skv, int dh, float scale, float drop_p,
                           uint64_t drop_seed, void* stream);
/* mmrca_mha_bwd followed by dqkv_colsum[3*H*dh] (fp32) += column sums of the stored dqkv = the bias gradient of the QKV
 * in-projection (one call; the reduction is a separate HBM pass -- fusing it into the MFMA kernels measured slower).
 * total_rows = number of token rows (B*S padded, cu_seqlens[B] packed). */
int mmrca_mha_bwd_colsum(const void* qkv, const int32_t* key_mask, const void* out, const void* dout, const float* lse,
                         void* dqkv, float* dqkv_colsum, int64_t total_rows, int B, int H, int S, int dh, float scale,
                         float drop_p, uint64_t drop_seed, const int32_t* cu_seqlens, int dtype, int impl, void* stream);
/* K3x. Attention FORWARD with separate operands: out[b*Sq + i, h*dh + d] = softmax_j(q_i . k_j * scale) v_j over the S_kv
 * rows of sequence b.  q: [B*Sq, ldq], k / v: [B*Skv, ldk / ldv], out: [B*Sq, ldo]; head h at columns h*dh of each.
 * Covers what the BLIP-2 path (q_former_training.py:289 -> transformers 5.15.0 modeling_blip_2.py) needs beyond K3:
 * the ViT-g tower's head dim 88 at S = 257 (modeling_blip_2.py:282-354: pass the three column blocks of its fused qkv
 * buffer) and the Q-Former cross-attention, 32 queries over 257 image tokens (modeling_blip_2.py:536-606).
 * drop_p > 0: attention-probability dropout, mask from (seed, ((b*H+h)*Sq + i)*Skv + j).  No key mask: the reference
 * passes an all-ones image_attention_mask.  bf16 with dh % 8 == 0, S_kv <= 288 and 16-byte aligned rows runs on MFMA
 * (impl = MMRCA_GEMM_MFMA demands it); anything else, and impl = MMRCA_GEMM_REF, runs the fp32-math kernel. */
int mmrca_mha_cross_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out,
                        int64_t ldo, int B, int H, int Sq, int Skv, int dh, float scale, float drop_p, uint64_t drop_seed,
                        int dtype, int impl, void* stream);
/* K3c. The same attention for the class-token query only (row 0 of every sequence): what the LAST encoder layer needs,
 * because the reference reads hidden_state[:, 0] (multimodal_model.py:352,517) / torchvision reads x[:, 0] and nothing else
 * of that layer's output.  out / dout: [B, H*dh] (compact), lse: fp32 [B,H].  The backward fills the WHOLE fused dqkv
 * buffer: dQ row 0, dK / dV of every key row, zeros in dQ rows >= 1.  Equal to row 0 of mmrca_mha_fwd, and to
 * mmrca_mha_bwd with dout zero outside row 0 (same mask, dropout-counter and cu_seqlens semantics). */
int mmrca_mha_cls_fwd(const void* qkv, const int32_t* key_mask, void* out, float* lse,
                      int B, int H, int S, int dh, float scale, float drop_p, uint64_t drop_seed,
                      const int32_t* cu_seqlens, int dtype, void* stream);
int mmrca_mha_cls_bwd(const void* qkv, const int32_t* key_mask, const void* out, const void* dout, const float* lse,
                      void* dqkv, int B, int H, int S, int dh, float scale, float drop_p, uint64_t drop_seed,
                      const int32_t* cu_seqlens, int dtype, void* stream);

/* K4. s = x (+ res);  y = LayerNorm(s) * gamma + beta.  sum_out (optional) receives s.  mean/rstd fp32 [rows].
 * Row r of x/res/sum_out/y starts at r*ld_* elements (lets the ViT final norm run on class tokens only).
 * nn.LayerNorm at modeling_distilbert.py:98,240,247 (eps 1e-12), torchvision ln_1/ln_2/encoder.ln (1e-6).
 * Train-mode dropout of the HF encoders is fused here with counter-based masks (seed, row*D+col), inverted scaling:
 *   in_drop  : s = dropout(x) + res   (FFN-output / attention-output dropout, modeling_distilbert.py:222)
 *   out_drop : y = dropout(LayerNorm(s))   (embedding dropout, modeling_distilbert.py:117) */
int mmrca_add_layernorm_fwd(const void* x, const void* res, const void* gamma, const void* beta,
                            void* sum_out, void* y, float* mean, float* rstd,
                            int64_t rows, int D, int64_t ld_x, int64_t ld_y, float eps,
                            float in_drop_p, uint64_t in_drop_seed, float out_drop_p, uint64_t out_drop_seed,
                            int dtype, void* stream);
/* The same on fp32 operands with the normalised output ALSO (y != NULL) or ONLY (y == NULL) written as two bf16 planes
 * y_hi + y_lo: the operand form of the bf16x3 GEMMs that consume it (mmrca_gemm_x3), saving their split pass. */
int mmrca_add_layernorm_fwd_x3(const void* x, const void* res, const void* gamma, const void* beta,
                               void* sum_out, void* y, void* y_hi, void* y_lo, float* mean, float* rstd,
                               int64_t rows, int D, int64_t ld_x, int64_t ld_y, float eps,
                               float in_drop_p, uint64_t in_drop_seed, float out_drop_p, uint64_t out_drop_seed, void* stream);
/* ds = LN'(dy) (+ dres);  dgamma/dbeta (fp32) += .   s is the saved LayerNorm input.
 * dy_drop: mask dy like the forward's out_drop.  dbranch (optional) = dropout-mask(LN'(dy)) with (branch_drop_p, seed):
 * the gradient of the branch that the forward's in_drop dropped (ds itself stays the residual-stream gradient).
 * dcol / dcol_branch (optional, fp32 [D], +=): column sums of ds / of dbranch = the bias gradient of the linear layer whose
 * output fed this LayerNorm (saves a separate pass over the gradient). */
int mmrca_layernorm_bwd(const void* dy, const void* s, const void* gamma, const float* mean, const float* rstd,
                        const void* dres, void* ds, float* dgamma, float* dbeta,
                        int64_t rows, int D, int64_t ld_dy, int64_t ld_s, int64_t ld_ds,
                        float dy_drop_p, uint64_t dy_drop_seed, float branch_drop_p, uint64_t branch_drop_seed,
                        void* dbranch, float* dcol, float* dcol_branch, int dtype, void* stream);
/* The same with bf16 gradients / gamma (dy, dres, ds, dbranch, gamma) against an fp32 saved sum `s`: the LayerNorm backward of the
 * bf16x3f mode, whose forward keeps the residual stream in fp32 (as the reference does, CVPR_code/multimodal_model.py:651-659)
 * while its backward runs at bf16 precision.  D % 8 == 0, D <= 1024, 16-byte aligned operands. */
int mmrca_layernorm_bwd_mixed(const void* dy, const float* s, const void* gamma, const float* mean, const float* rstd,
                              const void* dres, void* ds, float* dgamma, float* dbeta,
                              int64_t rows, int D, int64_t ld_dy, int64_t ld_s, int64_t ld_ds,
                              float dy_drop_p, uint64_t dy_drop_seed, float branch_drop_p, uint64_t branch_drop_seed,
                              void* dbranch, float* dcol, float* dcol_branch, void* stream);

/* K5a. text embeddings: out[r] = word[ids[r]] + pos[pos_ids[r]] (+ type_row)   (modeling_distilbert.py:82-118,
 * BertEmbeddings; LayerNorm follows via mmrca_add_layernorm_fwd).  ids/pos_ids int32 [rows]. */
int mmrca_embed_fwd(const int32_t* ids, const int32_t* pos_ids, const void* word, const void* pos,
                    const void* type_row, void* out, int64_t rows, int D, int dtype, void* stream);
/* backward (fp32 tables, +=).  pad_id / pos_pad_id: nn.Embedding(padding_idx=pad_token_id) -- rows with ids[r] == pad_id add
 * nothing to dword (all three HF encoders), rows with pos_ids[r] == pos_pad_id nothing to dpos (RoBERTa only); -1 = none. */
int mmrca_embed_bwd(const void* dout, const int32_t* ids, const int32_t* pos_ids, float* dword, float* dpos,
                    float* dtype_row, int64_t rows, int D, int pad_id, int pos_pad_id, int dtype, void* stream);

/* K5b. ViT patch embedding (torchvision conv_proj 16x16/16 + class token + pos embedding).
 * patchify: images fp32 NCHW [B,3,H,W] -> rows [B*nP, 3*P*P] (k = c*P*P + py*P + px), nP=(H/P)*(W/P).
 * assemble: x[b,0] = cls + pos[0]; x[b,1+i] = proj[b*nP+i] + pos[1+i]. */
int mmrca_patchify_fwd(const float* images, void* patches, int B, int C, int Himg, int Wimg, int P, int dtype, void* stream);
int mmrca_vit_assemble_fwd(const void* proj, const void* cls, const void* pos, void* x, int B, int nP, int D, int dtype, void* stream);
int mmrca_vit_assemble_bwd(const void* dx, void* dproj, float* dcls, float* dpos, int B, int nP, int D, int dtype, void* stream);

/* K1. fused MM-RCA fusion head (CVPR_code/multimodal_model.py:662-726 with SelfAttention :39-68 and
 * ReverseCrossAttention :71-108): L2-normalise, reshape to 16 pseudo-patches, 2x self-attention, 2x (reverse)
 * cross-attention, LayerNorm+ReLU, concat by mode, dropout, final linear.  One 512-thread workgroup per sample, fp32
 * matrix cores (v_mfma_f32_16x16x4_f32: 16 pseudo-patches = one tile).
 * weights: fp32 pointers in the order of MmrcaHeadWeights.  mode: 0 default, 1 features_only, 2 cross_attention_only.
 * drop_p>0 applies inverted dropout to the concatenated features with a counter-based mask (seed, sample, column).
 * The backward takes the same feature pointers and recomputes the (tiny) forward in LDS; it is two launches: the
 * per-sample kernel leaves dQ|dK|dV and the projection inputs in `workspace` (mmrca_head_bwd_workspace_bytes(), ~98 KB
 * per sample, 16-byte aligned), and a second kernel forms all weight gradients from it as GEMMs over the B*16 rows.
 * Gradients are ACCUMULATED into g (fp32 atomics).
 * dimg/dtxt (optional): gradients wrt the un-normalised backbone features, in `dtype`. */
typedef struct {
  const float *sai_wq, *sai_bq, *sai_wk, *sai_bk, *sai_wv, *sai_bv, *sai_g, *sai_b;   /* self_attention_image */
  const float *sat_wq, *sat_bq, *sat_wk, *sat_bk, *sat_wv, *sat_bv, *sat_g, *sat_b;   /* self_attention_text  */
  const float *c1_wq, *c1_bq, *c1_wk, *c1_bk, *c1_wv, *c1_bv, *c1_g, *c1_b;           /* cross_attention_1    */
  const float *c2_wq, *c2_bq, *c2_wk, *c2_bk, *c2_wv, *c2_bv, *c2_g, *c2_b;           /* cross_attention_2    */
  const float *fin_w, *fin_b;                                                         /* active final linear  */
} MmrcaHeadWeights;
typedef struct {
  float *sai_wq, *sai_bq, *sai_wk, *sai_bk, *sai_wv, *sai_bv, *sai_g, *sai_b;
  float *sat_wq, *sat_bq, *sat_wk, *sat_bk, *sat_wv, *sat_bv, *sat_g, *sat_b;
  float *c1_wq, *c1_bq, *c1_wk, *c1_bk, *c1_wv, *c1_bv, *c1_g, *c1_b;
  float *c2_wq, *c2_bq, *c2_wk, *c2_bk, *c2_wv, *c2_bv, *c2_g, *c2_b;
  float *fin_w, *fin_b;
} MmrcaHeadGrads;
int mmrca_head_fwd(const void* img, const void* txt, const MmrcaHeadWeights* w, float* logits,
                   int B, int d_img, int d_txt, int n_classes, int reverse, int mode,
                   float drop_p, uint64_t seed, int dtype, void* stream);
int64_t mmrca_head_bwd_workspace_bytes(int B, int d_img, int d_txt);
int mmrca_head_bwd(const float* dlogits, const void* img, const void* txt, const MmrcaHeadWeights* w,
                   const MmrcaHeadGrads* g, void* dimg, void* dtxt, int B, int d_img, int d_txt, int n_classes,
                   int reverse, int mode, float drop_p, uint64_t seed, int dtype, void* workspace,
                   int64_t workspace_bytes, void* stream);

/* K6. weighted, label-smoothed cross entropy (torch.nn.CrossEntropyLoss as built at main_both.py:87-93), mean
 * reduction with the weighted denominator.  loss: fp32[1]; dlogits (optional): fp32 [B,C] scaled by grad_scale. */
int mmrca_xent_fwd_bwd(const float* logits, const int32_t* labels, const float* class_w, float smoothing,
                       float* loss, float* dlogits, int B, int C, float grad_scale, void* stream);

/* optimizers over the flat fp32 arenas (torch.optim.SGD / AdamW as constructed at main_both.py:544-549:
 * SGD: g += wd*p; p -= lr*g.   AdamW: p *= 1-lr*wd; Adam step with bias correction, eps 1e-8).
 * lp (optional): bf16 working copy of p, refreshed in the same pass. */
int mmrca_sgd_step(float* p, const float* g, void* lp, int64_t n, float lr, float wd, float grad_scale, void* stream);
int mmrca_adamw_step(float* p, const float* g, float* m, float* v, void* lp, int64_t n, float lr, float beta1,
                     float beta2, float eps, float wd, int step, float grad_scale, void* stream);
int mmrca_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* the same steps in bf16x3 mode: lp_hi / lp_lo receive the two bf16 planes of the updated parameters (lp_lo may be NULL) */
int mmrca_sgd_step_x3(float* p, const float* g, void* lp_hi, void* lp_lo, int64_t n, float lr, float wd, float grad_scale, void* stream);
int mmrca_adamw_step_x3(float* p, const float* g, float* m, float* v, void* lp_hi, void* lp_lo, int64_t n, float lr, float beta1,
                        float beta2, float eps, float wd, int step, float grad_scale, void* stream);

/* K7 (the step before the hot path).  The reference's validation image pipeline
 * (main_both.py:433-440: PadToMaintainAR keep_aspect_ratio.py:18-53 -> A.Resize(INTER_LINEAR) -> A.Normalize ->
 * ToTensorV2, plus the train pipeline's V/H flips :416-417) for a batch of decoded uint8 HWC images of different sizes
 * that sit back to back in one device staging buffer.  out: fp32 [B, 3, out_h, out_w].  The padding itself is decided
 * on the host (preprocess.py::plan_padding restates the reference's rule, axis quirk included) and passed per image. */
typedef struct {
  int64_t offset;            /* byte offset of the image in the staging buffer (uint8, HWC, tightly packed) */
  int32_t h, w;              /* decoded size */
  int32_t pad_top, pad_left; /* zeros added before the first row / column */
  int32_t ph, pw;            /* padded size = what the resize sees */
  int32_t flip_v, flip_h;    /* A.VerticalFlip / A.HorizontalFlip applied to the resized image */
} MmrcaImageDesc;
int mmrca_image_preprocess(const void* staging, const void* desc /* MmrcaImageDesc[B], device */, float* out, int B,
                           int out_h, int out_w, const float* mean3 /* host */, const float* std3 /* host */, void* stream);

/* K7b (row f1, training side): the six remaining augmentations of the reference's TRAIN_PIPELINE (main_both.py:407-429)
 * as per-image descriptors executed on the GPU.  The random draws (which transforms fire, with which parameters) are
 * made on the host (preprocess.py::sample_train_params follows albumentations' distributions); the kernels are
 * deterministic functions of the descriptors.  Stage order = the reference's:
 *   A.Rotate(crop_border) -> PadToMaintainAR -> A.Resize -> A.GaussianBlur -> V/H flip -> A.RandomBrightnessContrast ->
 *   A.Sharpen -> A.Perspective -> A.ShiftScaleRotate(scale only) -> A.Normalize -> ToTensorV2
 * uint8 re-quantisation between stages is kept (cv2 returns uint8 images).  See garbage_classification_rca_amd/csrc/augment.hip
 * for the arithmetic of each stage. */
typedef struct {
  int64_t src_offset;        /* decoded image in the staging buffer */
  int64_t dst_offset;        /* where the rotated + cropped image goes (same buffer, past the sources) */
  int32_t h, w;              /* source size */
  int32_t dh, dw;            /* size after crop_border */
  int32_t x_min, y_min;      /* crop origin in the rotated (h x w) frame */
  int32_t enabled, pad_;
  float inv[6];              /* rotated-frame pixel (x, y) -> source pixel: sx = inv0 x + inv1 y + inv2, sy = inv3 x + inv4 y + inv5 */
} MmrcaRotateDesc;             /* 72 bytes */
int mmrca_image_rotate_crop(void* staging, const void* desc /* MmrcaRotateDesc[B], device */, int B, int max_pixels, void* stream);
/* PadToMaintainAR + Resize only (no flips, no normalisation): uint8 [B, out_h, out_w, 3] for the stages below */
int mmrca_image_resize_u8(const void* staging, const void* desc /* MmrcaImageDesc[B] */, void* out_u8, int B, int out_h, int out_w,
                          void* stream);
typedef struct {
  float blur[7];             /* 1-D Gaussian taps; the 2-D weight is blur[i] * blur[j] */
  int32_t blur_k;            /* 0 = off, else 3 / 5 / 7 */
  int32_t flip_v, flip_h;
  int32_t has_bc;            /* brightness/contrast: v = trunc(clip(v * bc_alpha + bc_beta, 0, 255)) */
  float bc_alpha, bc_beta;
  int32_t has_sharp;         /* 3x3 correlation, reflect-101 border, round-to-nearest-even, saturate */
  float sharp[9];
  int32_t has_persp;         /* output pixel -> input pixel homography (keep_size resize folded in), constant border 0 */
  float persp[9];
  int32_t has_scale;         /* output pixel -> input pixel affine map of ShiftScaleRotate */
  float scale[6];
} MmrcaAugDesc;                /* 160 bytes */
/* in_u8 / tmp_u8: uint8 [B, H, W, 3] (in_u8 is overwritten); out: fp32 [B, 3, H, W], normalised */
int mmrca_image_augment(void* in_u8, void* tmp_u8, const void* desc /* MmrcaAugDesc[B], device */, float* out, int B, int H, int W,
                        const float* mean3 /* host */, const float* std3 /* host */, void* stream);

#ifdef __cplusplus
}
#endif
#endif
