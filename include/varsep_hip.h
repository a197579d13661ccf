/*
 * varsep_hip.h -- C ABI of libvarsep_hip.so: the MI355X (gfx950) kernels behind the var_sep training
 * hot path (encoders E_s/E_t -> residual latent integrator -> decoder D, forward and backward).
 *
 * The reference (JeremieDona/spatiotemporal_variable_separation) has no native layer: every numeric
 * op on this path is a stock torch.nn call that lands in ATen.  The entry
 * points below are therefore what a maintainer would bind in place of those ATen calls; each one
 * cites the reference call site(s) it replaces (paths relative to /root/reference/var_sep).
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (torch); the library never allocates,
 *     frees or keeps a pointer after the call returns; workspaces are passed in explicitly;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *     and is legal inside hipGraph stream capture (no allocation, no synchronisation);
 *   - return value: VS_OK (0) or a negative VS_ERR_* code; nothing throws; vs_last_error() returns a
 *     static description of the most recent failure on the calling thread;
 *   - matrices are row-major; tensors are NCHW contiguous unless a leading dimension is given;
 *   - dtype codes: VS_F32 = 0, VS_BF16 = 1, VS_F16 = 2.  "Compute type" selects the MFMA family: VS_F32 runs the
 *     exact-fp32 v_mfma_f32_32x32x2_f32 path (parity mode, <=1e-3 vs the fp32 CPU oracle is met with
 *     orders of magnitude to spare), VS_BF16 / VS_F16 run v_mfma_f32_32x32x16_{bf16,f16} with fp32
 *     accumulation.  Wherever a signature says VS_F32|VS_BF16 for a storage dtype, VS_F16 is accepted too.
 */
#ifndef VARSEP_HIP_H
#define VARSEP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VS_OK 0
#define VS_ERR_ARG (-1)       /* bad argument (null pointer, negative size, unsupported enum) */
#define VS_ERR_WORKSPACE (-2) /* workspace too small: call the matching *_workspace_bytes query */
#define VS_ERR_LAUNCH (-3)    /* hipLaunchKernel failed; see vs_last_error() */
#define VS_ERR_UNSUPPORTED (-4)

#define VS_F32 0
#define VS_BF16 1
#define VS_F16 2 /* IEEE half: the operand type of the reference's --torch_amp (torch.cuda.amp autocast, train.py:96-97) */

/* activation codes (networks/utils.py:50-72 activation_factory) */
#define VS_ACT_NONE 0
#define VS_ACT_RELU 1
#define VS_ACT_LEAKY 2 /* LeakyReLU(0.2) */
#define VS_ACT_SIGMOID 3
#define VS_ACT_TANH 4
#define VS_ACT_ELU 5

/* operand layouts of vs_gemm: element (i, k) of an operand with leading dimension ld lives at
 *   VS_LAYOUT_R : ptr[i * ld + k]   (reduction index k contiguous)
 *   VS_LAYOUT_S : ptr[k * ld + i]   (output index i contiguous)                                    */
#define VS_LAYOUT_R 0
#define VS_LAYOUT_S 1

const char* vs_version(void);
const char* vs_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * vs_gemm: C[m, n] = epilogue( sum_k A(m, k) * B(n, k) ),  m < M, n < N, k < K.
 *
 * Replaces the dense contractions of the path:
 *   nn.Linear forward            networks/mlp.py:40, networks/conv.py:124        A=x (R), B=W[out,in] (R)
 *   nn.Linear input gradient     (autograd of the above)                          A=dy (R), B=W (S)
 *   nn.Linear weight gradient    (autograd of the above)                          A=dy (S), B=x (S)
 *   ConvTranspose2d k4 s1 p0 on a 1x1 map (conv.py:258,295) and Conv2d k4 valid on a 4x4 map
 *   (conv.py:170), which are plain GEMMs over flattened weights.
 *
 * Epilogue, applied in this order to the fp32 accumulator v of element (m, n):
 *   v *= alpha;  v += bias[n] (if bias);  v = act(v);
 *   v *= act'(mask[m*ldmask+n]) (if mask: derivative of `mask_act` evaluated from its OUTPUT value --
 *        1/0 for ReLU, 1/0.2 for LeakyReLU, y(1-y) for sigmoid, 1-y^2 for tanh; fuses the activation
 *        backward of the previous layer into the input-gradient GEMM);
 *   v += C_old[m, n] (if accumulate);  store as c_dtype.
 *
 * compute: VS_F32 (A, B stored fp32) or VS_BF16 (A, B stored bf16).  c_dtype / mask_dtype: VS_F32|VS_BF16.
 * Any M, N, K >= 1 and any ld are accepted (unaligned shapes take a scalar load path).
 * workspace: only used when the library chooses split-K (few output tiles, long K); query
 * vs_gemm_workspace_bytes(M, N, K) for an upper bound.  May be NULL when that bound is 0.
 */
/* `batch` independent GEMMs of one shape in one launch: problem i reads A + i*stride_a, B + i*stride_b (elements of the compute
 * type; multiples of 16 bytes keep the vector loads) and writes C + i*stride_c (elements of c_dtype); no bias / activation / mask.  Used for the weight
 * gradients of the integrator's residual blocks (reference: resnet.py:22-50, three Linear layers per block).              */
size_t vs_gemm_batched_workspace_bytes(int batch, int64_t M, int64_t N, int64_t K);
int vs_gemm_batched(int compute, int batch, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, int64_t stride_a, int layout_a,
                    const void* B, int64_t ldb, int64_t stride_b, int layout_b, void* C, int64_t ldc, int64_t stride_c, int c_dtype,
                    float alpha, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

size_t vs_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);

int vs_gemm(int compute, int64_t M, int64_t N, int64_t K,
            const void* A, int64_t lda, int layout_a,
            const void* B, int64_t ldb, int layout_b,
            void* C, int64_t ldc, int c_dtype,
            float alpha, const float* bias, int act,
            const void* mask, int64_t ldmask, int mask_dtype, int mask_act,
            int accumulate,
            void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise / reduction helpers around the GEMMs.
 */

/* dst[i] = (dst_dtype) src[i], i < n.  fp32 master weights -> bf16 shadow copies, input frames -> bf16. */
int vs_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);

/* Strided 2-D copy with conversion: dst[r*ldd + c] = src[r*lds + c], r < rows, c < cols.
 * Used to cut the encoder's temporal window out of the [B, T, C*H*W] frame tensor without a
 * host-side slice (train.py:76 full_data[:, t_random - nt_cond : t_random]); when `col_offset_dev`
 * is non-NULL the source column offset is *col_offset_dev * col_offset_scale, read on the device, so
 * a captured hipGraph stays valid when the random window moves.                                      */
int vs_copy2d(const void* src, int src_dtype, int64_t lds, void* dst, int dst_dtype, int64_t ldd,
              int64_t rows, int64_t cols, const int32_t* col_offset_dev, int64_t col_offset_scale, void* stream);
/* Two windows of one [rows, lds] source stacked into a [2 rows, ldd] operand in one launch (the E_t input [random window; conditioning
 * window] of train.py:45-88): rows [0, rows) from src + elem_offset_a + *col_offset_dev * col_offset_scale, rows [rows, 2 rows) from
 * src + elem_offset_b (offsets in elements; the device offset may be NULL).  Converts between the three element types.             */
int vs_copy2d_pair(const void* src, int src_dtype, int64_t lds, void* dst, int dst_dtype, int64_t ldd, int64_t rows, int64_t cols,
                   const int32_t* col_offset_dev, int64_t col_offset_scale, int64_t elem_offset_a, int64_t elem_offset_b, void* stream);

/* out[n] (+)= sum_m X[m*ldx + n]  (fp32 out).  Bias gradient of nn.Linear / conv (autograd of
 * mlp.py:40).  `accumulate` = 0 overwrites.  Internally zeroes/accumulates with float atomics over row
 * blocks, so results can differ in the last bits between runs (documented in DESIGN.md).            */
int vs_colsum(const void* X, int x_dtype, int64_t ldx, int64_t M, int64_t N, float* out, int accumulate,
              void* stream);

/* Up to 12 column-sum jobs in ONE launch (all bias gradients of a Linear chain or of the integrator): job j adds the
 * column sums of X[j] ([M[j], N[j]], leading dimension ldx[j]) into out[j][0..N[j]).  All arrays are HOST arrays of
 * n_jobs entries.  When zero_base is non-NULL, zero_count floats starting there are cleared first (lay the outputs out in
 * one flat buffer and clear it with this single memset).                                                               */
int vs_colsum_multi(int n_jobs, const void* const* X, const int* x_dtype, const int64_t* ldx, const int64_t* M, const int64_t* N,
                    float* const* out, float* zero_base, int64_t zero_count, void* stream);

/* Batch assembly from a simulation set resident in HBM (reference: data/wave_eq.py:67-72 `WaveEq.__getitem__` and :86-90
 * `WaveEqPartial.__getitem__`, for a whole batch of sampler indices): data [n_seq, nt, frame_elems] fp32; item b =
 * item_idx[b] selects sequence item / windows_per_seq and first frame item % windows_per_seq; out [batch, seq_len, frame_elems]
 * (or [batch, seq_len, n_pixels] when pixel_idx != NULL picks n_pixels fixed positions of every frame) in out_dtype.
 * Item indices outside [0, n_seq * windows_per_seq) are the caller's error (the sampler draws from that range).          */
int vs_gather_windows(const float* data, int64_t n_seq, int64_t nt, int64_t frame_elems, const int32_t* item_idx, int batch,
                      int windows_per_seq, int seq_len, const int32_t* pixel_idx, int n_pixels, void* out, int out_dtype,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * ConvTranspose2d k4 s2 p1 forward without any column matrix (reference: the three middle layers of DCGAN64Decoder,
 * networks/conv.py:259-262 `make_conv_block(nn.ConvTranspose2d(.., 4, 2, 1, bias=False), ..)` and their BatchNorm2d).
 * "Tap GEMM + col2im epilogue" (csrc/vs_conv_tap.hip): per block of 16 output channels and tile of 256 input pixels (whole
 * images of 4x4, 8x8 or 16x16) one 256x256 MFMA tile over K = Cin whose rows are (tap, channel); the input tile is staged in LDS
 * by LDS-DMA straight from NCHW, the 16 tap maps are combined in LDS, rounded once, written as 16-byte output runs.
 *   vs_convt_tap_supported : 1 when the geometry is served (16-bit compute, square 4x4 / 8x8 / 16x16 inputs, Cin % 8 == 0,
 *                            Cin >= 32, Cout >= 8, (B / groups) % (256 / (H*W)) == 0), else 0 -> use vs_conv_transpose2d_fwd.
 *   vs_convt_tap_pack_weight: fp32 master weight [Cin][Cout][4][4] -> dst [ceil(Cout/16)][16 taps x 16 channels][Cin] (compute type;
 *                            vs_convt_tap_packed_elems elements), once per optimizer step.
 *   vs_convt_k4s2_tap_fwd  : y [B, Cout, 2H, 2W] (compute type) = convT(x [B, Cin, H, W]) + bias.  bn_sums (fp64 [groups][Cout][2],
 *                            may be NULL): sum and sum of squares of the STORED outputs per (BatchNorm call group, channel),
 *                            zeroed and accumulated by this call -- feeds vs_bn_stats_from_sums, replacing the vs_bn_stats pass.
 *   vs_bn_stats_from_sums  : vs_bn_stats's outputs (mean, invstd [groups, C]; running statistics folded group by group) from
 *                            those sums; n_per_group = elements per (group, channel).                                        */
int vs_convt_tap_supported(int compute, int B, int Cin, int H, int W, int Cout, int groups);
size_t vs_convt_tap_packed_elems(int Cin, int Cout);
int vs_convt_tap_pack_weight(int compute, const float* w, int Cin, int Cout, void* dst, void* stream);
int vs_convt_k4s2_tap_fwd(int compute, const void* x, const void* w_tap, const float* bias, void* y, double* bn_sums, int B, int Cin, int H,
                          int W, int Cout, int groups, void* stream);
/* vs_convt_k4s2_tap_fwd with an fp32 output tensor (no BatchNorm sums): the fp32 parity mode (VARSEP_FP32_SPLIT) assembles the fp32 transposed
 * convolution -- ConvTranspose2d k4 s2 p1 forward, conv.py:260-263, and the input gradient of Conv2d k4 s2 p1, conv.py:119-122 -- from six launches
 * on bf16 pieces of its operands.                                                                                                     */
int vs_convt_k4s2_tap_fwd_f32(int compute, const void* x, const void* w_tap, const float* bias, float* y, int B, int Cin, int H, int W, int Cout,
                              void* stream);
/* Conv2d k3 s1 p1 the same way (reference: every 3x3 block of EncoderSST / DecoderSST(_Skip) conv.py:323-426, ConvResBlock
 * resnet.py:53-88, VGG64Encoder / VGG64Decoder conv.py:127-171, 267-320) on 4x4 / 8x8 / 16x16 maps: tile rows = 9 taps x 28
 * channels, out[m][y][x] = sum_t G_t[m][y+ky-1][x+kx-1] in the epilogue.  The input gradient is the same kernel on dz with the
 * weight read transposed and flipped: pack with flip = 1, Cin := the conv's Cout, Cout := the conv's Cin.
 * y_dtype may be VS_F32 (a module's final block keeps fp32 outputs).                                                        */
int vs_conv_k3_tap_supported(int compute, int B, int Cin, int H, int W, int Cout, int groups);
size_t vs_conv_k3_tap_packed_elems(int Cin, int Cout);
int vs_conv_k3_tap_pack_weight(int compute, const float* w, int Cin, int Cout, int flip, void* dst, void* stream);
int vs_conv_k3s1_tap_fwd(int compute, const void* x, const void* w_tap, const float* bias, void* y, int y_dtype, double* bn_sums, int B,
                         int Cin, int H, int W, int Cout, int groups, void* stream);
/* Conv2d k3 s1 p1 on a FEW 16x16 maps with many channels -- the ConvResnet integrator of the SST recipe (ConvResBlock resnet.py:53-70:
 * 64 -> 512 -> 512 -> 64 on 8 maps, applied once per predicted frame) -- as ONE launch that fills the chip: workgroup = (image, 32 output
 * channels, split of the input channels), the split's channels of the image in LDS, weights streamed in MFMA fragment order from the
 * pre-pack, the x shift of the taps taken on the result.  The split partial sums go to fp32 slabs
 * [vs_conv3_img16_splits(B, Cin, Cout)][B][Cout][256] WITHOUT bias; vs_bn_train_fwd_small_slabs (conv bias + BatchNorm + activation)
 * or vs_slab_sum (bias, output type) consume them.  Input gradient: the same call on dz with the weight packed with flip = 1,
 * Cin := the conv's Cout, Cout := the conv's Cin.  Cin must be a multiple of 64.                                                  */
int vs_conv3_img16_supported(int compute, int B, int Cin, int H, int W, int Cout);
int vs_conv3_img16_splits(int B, int Cin, int Cout);
size_t vs_conv3_img16_packed_elems(int Cin, int Cout);
int vs_conv3_img16_pack_weight(int compute, const float* w, int Cin, int Cout, int flip, void* dst, void* stream);
/* Up to 96 such pre-packs in ONE launch (every weight changes once per optimizer step, so all packs of a network are stale together:
 * 36-67 launches of ~5 us per TaxiBJ / SST step otherwise): job j = vs_conv3_img16_pack_weight(compute, w[j], K[j], M[j], flip[j], dst[j]).  */
int vs_conv3_img16_pack_weights(int compute, int n_jobs, const float* const* w, const int* K, const int* M, const int* flip, void* const* dst,
                                void* stream);
int vs_conv3_img16(int compute, const void* x, const void* w_packed, float* slabs, int B, int Cin, int Cout, void* stream);
/* The same contraction for MANY maps of width 16 / 32 / 64 (every 3x3 block of EncoderSST / DecoderSST(_Skip) conv.py:323-426 and of the VGG
 * encoders / decoders conv.py:127-171, 267-320 on whole batches): workgroup = 256 consecutive pixels of one map (256 / W rows) x 32 output
 * channels, the band's rows + halo of 64 channels at a time in LDS by double-buffered LDS-DMA, no column matrix.  H a multiple of 256 / W
 * (W = 8: whole 8 x 8 maps, four per workgroup), Cin a multiple of 64; weights from vs_conv3_img16_pack_weight (flip = 1 for the input gradient); y in any type, bias added.   */
int vs_conv3_band_supported(int compute, int B, int Cin, int H, int W, int Cout);
int vs_conv3_band(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int W,
                  int Cout, void* stream);
/* Weight gradient of the same convolution without a column matrix (the backward of every Conv2d(k=3, s=1, p=1) above): workgroup = (32 output
 * channels, 32 input channels, a share of the batch's row bands); dz and three column-shifted copies of the x band in LDS, nine 32x32
 * accumulators (one per tap) per wave.  Leaves vs_conv3_wgrad_band_slabs(...) fp32 partial gradients laid out [tap][Cout][Cin] (coalesced
 * stores); vs_conv3_wgrad_band_finish adds them into [Cout][Cin][3][3] (addend = the pending gradient for an accumulating call).  W in {16, 32, 64} with H a multiple of 256 / W, or 8 x 8 maps.    */
/* ... over a batch lying in npieces (<= 64) separate tensors of maps_per_piece maps each (the remembered (dz, x) pairs of a convolution applied
 * once per predicted frame): the gradient over their concatenation without building it; slabs as for B = npieces * maps_per_piece.   */
int vs_conv3_wgrad_band_pieces(int compute, int npieces, const void* const* x, const void* const* dz, int maps_per_piece, float* slabs, int Cin,
                               int H, int W, int Cout, void* stream);
int vs_conv3_wgrad_band_finish(const float* slabs, int nslabs, const float* addend, float* out, int Cout, int Cin, void* stream);
int vs_slab_sum_grouped(const float* slabs, int nslabs, int groups, float* partial, int64_t total, void* stream);   /* first pass over many slabs */
int vs_conv3_wgrad_band_supported(int compute, int B, int Cin, int H, int W, int Cout);

/* ---- 4x4 stride-2 pad-1 convolutions without a column matrix (csrc/vs_conv_k4s2.hip).  Replaces, for the DCGAN stride-2 layers, the
 * nn.Conv2d(c, 2c, 4, 2, 1) forward / weight gradient of reference networks/conv.py:119-122 and the input / weight gradient of
 * nn.ConvTranspose2d(2c, c, 4, 2, 1) of conv.py:260-263.  A stride-2 tap never mixes the parities of the input grid, so on the four
 * parity planes of the input ("space to depth": planes [B][4 C][H/2][W/2], channel = (row parity * 2 + column parity) * C + c) the 4x4
 * window is a 3x3 stride-1 pad-1 window over 4 C channels with 2x2 non-zero taps per plane, and the row-band kernels above
 * (vs_conv3_band, vs_conv3_wgrad_band) carry it.
 *   vs_space_to_depth2          x [B][C][H][W] (16-bit, H even, W a multiple of 16) -> planes
 *   vs_conv_k4s2_pack_weight    fp32 w [M][K][4][4] -> the vs_conv3_band pre-pack over 4 K plane channels (vs_conv_k4s2_packed_elems elements):
 *                               Conv2d weight [Cout][Cin][4][4] for its forward; ConvTranspose2d weight [Cin][Cout][4][4] (M = Cin, K = Cout)
 *                               for its input gradient -- the same index order, no flip
 *   vs_conv_k4s2_wgrad_finish   slabs [nslabs][9][M][4 K] of vs_conv3_wgrad_band(x = planes of the LARGE map, dz = the SMALL map with M
 *                               channels) -> dW [M][K][4][4] (+ addend): Conv2d: M = Cout, K = Cin, large = input, small = dz;
 *                               ConvTranspose2d: M = Cin, K = Cout, large = the output gradient, small = the input                      */
int vs_space_to_depth2_supported(int compute, int B, int C, int H, int W);
int vs_space_to_depth2(int compute, const void* x, void* planes, int B, int C, int H, int W, void* stream);
size_t vs_conv_k4s2_packed_elems(int K, int M);
int vs_conv_k4s2_pack_weight(int compute, const float* w, int K, int M, void* dst, void* stream);
int vs_conv_k4s2_wgrad_finish(const float* slabs, int nslabs, const float* addend, float* out, int M, int K, void* stream);
/*   vs_conv_k4s2_band           the gather on the planes [B][4 K][H][W] -> y [B][M][H][W] (+ bias), vs_conv3_band's arguments with K in place of
 *                               Cin; K a multiple of 64 (vs_conv_k4s2_skip_form): the kernel form that streams and multiplies only the 2 x 2
 *                               taps a plane sees (the pack has the matching form), otherwise the plain 3 x 3 kernel on a zero-padded pack
 *   vs_conv_k4s2_wgrad_band     slabs [vs_conv_k4s2_wgrad_band_slabs(B, K, H, W, M)][9][M][4 K] from the planes and the small map [B][M][H][W]  */
int vs_conv_k4s2_skip_form(int K);
int vs_conv_k4s2_band(int compute, const void* planes, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int K, int H, int W, int M,
                      void* stream);
int vs_conv_k4s2_wgrad_band(int compute, const void* planes, const void* small_map, float* slabs, int B, int K, int H, int W, int M, void* stream);
int vs_conv_k4s2_wgrad_band_slabs(int B, int K, int H, int W, int M);    /* slabs the call above writes (round 4: not vs_conv3_wgrad_band_slabs' count) */
/* vs_conv3_band / vs_conv_k4s2_band with the BatchNorm statistics of the output taken in the epilogue (replaces the vs_bn_stats pass behind
 * conv -> BatchNorm, reference conv.py:41-60): (sum, sum of squares) of the STORED values are added to bn_sums [groups][Cout][2] (fp64),
 * group = map / (B / groups).  vs_bn_stats_from_sums_fold turns them into mean / invstd [groups][C], folds the running estimates in call order
 * and (reset != 0) leaves the sums at zero for the next step -- one launch instead of vs_bn_stats_from_sums' two. */
int vs_conv3_band_bn_supported(int compute, int B, int Cin, int H, int W, int Cout, int groups);
int vs_conv3_band_bn(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int W, int Cout,
                     double* bn_sums, int groups, void* stream);
int vs_conv_k4s2_band_bn(int compute, const void* planes, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int K, int H, int W, int M,
                         double* bn_sums, int groups, void* stream);
int vs_bn_stats_from_sums_fold(double* sums, int groups, int C, int64_t n_per_group, float* mean, float* invstd, float* running_mean, float* running_var,
                               float momentum, float eps, int reset, void* stream);
/* The same statistics without atomics (round 4; replaces the vs_bn_stats pass behind conv -> BatchNorm, reference conv.py:41-60, 119-122): every
 * workgroup writes the (sum, sum of squares) of the stored values of its 32 channels to its own row of parts [vs_conv3_band_bn_parts_rows(B, H,
 * W)][Cout][2] (fp32; row = map * bands + band, the rows of a call group are consecutive); k4 != 0: x are the parity planes [B][4 K][H][W] of a
 * k4 s2 p1 convolution (Cin = 4 K, pack of vs_conv_k4s2_pack_weight).  vs_bn_stats_from_parts_fold adds the rows of every call group in a fixed
 * order (fp64) -> mean / invstd [groups][C] and (running_mean != NULL; var_scratch [groups][C] then required) folds the running estimates in call
 * order with a second small launch.  No zero fill, reproducible launch to launch. */
int vs_conv3_band_bn_parts_rows(int B, int H, int W);
int vs_conv3_band_bn_parts(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int W, int Cout,
                           float* parts, int k4, void* stream);
int vs_bn_stats_from_parts_fold(const float* parts, int rows_per_group, int groups, int C, int64_t n_per_group, float* mean, float* invstd,
                                float* var_scratch, float* running_mean, float* running_var, float momentum, float eps, void* stream);
int vs_conv3_wgrad_band_slabs(int B, int Cin, int H, int W, int Cout);
int vs_conv3_wgrad_band(int compute, const void* x, const void* dz, float* slabs, int B, int Cin, int H, int W, int Cout, void* stream);
/* Round 4: a layer of the ConvResnet integrator (reference resnet.py:53-70: Conv2d k3 s1 p1 -> BatchNorm2d (training mode) [-> LeakyReLU]; 78 block
 * calls each way per SST step) in ONE launch instead of vs_conv3_img16 + vs_bn_train_fwd_small_slabs: the other input-channel splits' partial sums
 * and the other maps' BatchNorm statistics are exchanged inside the launch through epoch-tagged 8-byte granules (the MLP integrator's mechanism),
 * so every workgroup of the launch must be resident (vs_conv3_img16_bn_supported checks: <= 256 workgroups, splits 1 / 2 / 8).
 *   ws          vs_conv3_img16_bn_workspace_bytes() bytes, zero-filled ONCE by the caller and kept (per stream); word 0 = epoch base, word 1 =
 *               sticky error flag (a partner did not answer: results invalid), the rest exchange areas
 *   call_idx    1 .. 65535, distinct for every launch on this workspace since the last vs_exchange_epoch_advance (base += 65536; recordable)
 *   _fwd        z [B][Cout][256] (16-bit pre-BatchNorm output = round16(conv + bias), bit-identical to the two-launch path), y = act(BN(z)) in
 *               y_dtype, mean / invstd [Cout], running estimates updated; skip != NULL: xnew = skip + y (fp32) and its 16-bit copy xnew16
 *   _bwd        dz = BatchNorm + activation backward of this layer applied to dy = input gradient of the FOLLOWING layer's convolution of dz_next
 *               (w_packed: that layer's weight, flip = 1; Cin = its output channels); d gamma / d beta written or ADDED (accumulate) */
size_t vs_conv3_img16_bn_workspace_bytes(void);
int vs_conv3_img16_bn_supported(int compute, int B, int Cin, int Cout);
int vs_conv3_img16_bn_form_supported(int backward, int act, int y_dtype, int compute);   /* built: fwd LeakyReLU -> 16-bit y, fwd none -> fp32 y; bwd LeakyReLU / none */
int vs_exchange_epoch_advance(void* ws, void* stream);
int vs_conv3_img16_bn_fwd(int compute, const void* x, const void* w_packed, void* ws, unsigned call_idx, const float* bias, const float* gamma,
                          const float* beta, int act, float* running_mean, float* running_var, float momentum, float eps, void* z, void* y, int y_dtype,
                          float* mean, float* invstd, const float* skip, float* xnew, void* xnew16, int B, int Cin, int Cout, void* stream);
int vs_conv3_img16_bn_bwd(int compute, const void* dz_next, const void* w_packed, void* ws, unsigned call_idx, const void* z, const float* mean,
                          const float* invstd, const float* gamma, const float* beta, int act, void* dz, float* dgamma, float* dbeta, int accumulate,
                          int B, int Cin, int Cout, void* stream);
int vs_slab_sum(const float* slabs, int nslabs, const float* bias, const float* addend, void* out, int out_dtype, int B, int C, int64_t HW,
                void* stream);
/* Round 4: vs_slab_sum with a SECOND fp32 addend (the gradient of a ConvResBlock output that reaches the block along two paths -- the next
 * block-step and the stacked codes the decoder reads, resnet.py:66-70 / model.py:76-86 -- joins the skip gradient without an add launch). */
int vs_slab_sum2(const float* slabs, int nslabs, const float* bias, const float* addend, const float* addend2, void* out, int out_dtype, int B, int C,
                 int64_t HW, void* stream);
int vs_bn_stats_from_sums(const double* sums, int groups, int C, int64_t n_per_group, float* mean, float* invstd, float* var_scratch,
                          float* running_mean, float* running_var, float momentum, float eps, void* stream);

/* ---- convolutions between a map with many channels and the image side with 1..8 (csrc/vs_conv_thin.hip).  Replaces, for the first encoder
 * layer and the last decoder layer of every convolutional family -- nn.Conv2d(nc, 64, 4, 2, 1) (reference networks/conv.py:119),
 * nn.ConvTranspose2d(64, nc, 4, 2, 1) (conv.py:264-267), nn.Conv2d(nc, 64, 3, 1, 1) (conv.py:130-133, 345-350), nn.ConvTranspose2d(64, nc, 3, 1, 1)
 * (conv.py:300-303), nn.Conv2d(64, nc, 3, 1, 1) (conv.py:420-426) -- the forward, input-gradient and weight-gradient calls whose GEMM form has a
 * dimension of 1..8.  big [B][C][H][W], thin [B][M][S H][S W], (k, S) = (3, 1) or (4, 2), pad 1; a big pixel p = (y, x) meets the thin pixels
 * S p + t - 1 = (S y + ty - 1, S x + tx - 1).  Weights are read from the caller's 16-bit tensor: element (c, m, t) at w[c w_sc + m w_sm + t'],
 * t' = t, or k^2 - 1 - t with `flip` (a stride-1 Conv2d seen from its output side).
 *   vs_conv_thin_expand   out_big[c][p] = bias[c] + sum_{m,t} thin[m][S p + t - 1] w(c, m, t)      Conv2d forward with few input channels (w_sc = M k^2,
 *                         w_sm = k^2); ConvTranspose2d input gradient with few output channels (same strides); Conv2d k3 s1 input gradient with
 *                         few output channels (w_sc = k^2, w_sm = C k^2, flip)
 *   vs_conv_thin_reduce   out_thin[m][q] = bias[m] + sum_{c,t: S p + t - 1 = q} big[c][p] w(c, m, t)  ConvTranspose2d forward with few output channels
 *                         (w_sc = M k^2, w_sm = k^2); Conv2d k3 s1 forward with few output channels (w_sc = k^2, w_sm = C k^2, flip); M <= 4 (k 3) / 2 (k 4)
 *   vs_conv_thin_wgrad    out[c o_sc + m o_sm + t'] = (addend ? addend[..] : 0) + sum_{maps,p} big[c][p] thin[m][S p + t - 1]: both weight gradients
 *                         (Conv2d few inputs: big = dz, o_sc = M k^2, o_sm = k^2; ConvTranspose2d few outputs: big = x, same strides; Conv2d k3 s1 few
 *                         outputs: big = x, thin = dz, o_sc = k^2, o_sm = C k^2, flip); fp32 partials in `ws`, summed in a fixed order
 * C a multiple of 32, W in {8, .., 128} dividing 512, H W a multiple of 512, 16-bit operands, 16-byte aligned.                            */
int vs_conv_thin_supported(int compute, int B, int C, int H, int W, int M, int k, int stride, int pad);
int vs_conv_thin_expand(int compute, const void* thin, const void* w, int64_t w_sc, int64_t w_sm, int flip, const float* bias, void* out, int out_dtype,
                        int B, int C, int H, int W, int M, int k, int stride, void* stream);
int vs_conv_thin_reduce(int compute, const void* big, const void* w, int64_t w_sc, int64_t w_sm, int flip, const float* bias, void* out, int out_dtype,
                        int B, int C, int H, int W, int M, int k, int stride, void* stream);
size_t vs_conv_thin_wgrad_workspace_bytes(int B, int C, int H, int W, int M, int k);
int vs_conv_thin_wgrad(int compute, const void* big, const void* thin, float* ws, size_t ws_bytes, const float* addend, float* out, int64_t o_sc,
                       int64_t o_sm, int flip, int B, int C, int H, int W, int M, int k, int stride, void* stream);

/* Frame metrics of the evaluation scripts (test/mnist/test.py:136-142): for every plane pair (pred, target) [planes, H, W] fp32
 * mse[plane] = mean squared error (PSNR = 10 log10(1 / mse) follows on the host as in the reference) and ssim[plane] = mean over the
 * (H - 10) x (W - 10) "valid" window positions of the SSIM index of utils/ssim.py:81-111 (11 x 11 Gaussian window of the given
 * sigma, c1 = (k1 max_val)^2, c2 = (k2 max_val)^2) -- what `_ssim_wrapper` (test/utils.py:19-24) returns per (sample, frame, channel).
 * One launch, planes up to ~80 x 80 (VS_ERR_UNSUPPORTED beyond); either output may be NULL.                                  */
int vs_frame_metrics(const float* pred, const float* target, int64_t planes, int H, int W, float max_val, float k1, float k2, float sigma,
                     float* mse, float* ssim, void* stream);

/* One batch of Moving-MNIST training sequences rendered on the device (reference: data/moving_mnist.py:112-175 `__getitem__` +
 * `_compute_trajectory`, :177-255 `_process_collision`, deterministic mode as main.py:81-82 constructs it).
 * digits [n_digits_total, digit_h, digit_w] uint8 in HBM; init [batch, num_digits, 5] int32 = (digit index, start row, start column,
 * row speed, column speed) per digit, i.e. the five np.random.randint draws of the reference in its order; out [batch, seq_len, 1,
 * frame, frame] (fp32 or a 16-bit type): digits added at their bounced positions, clipped at 255, divided by 255.
 * Trajectories are computed in IEEE double, operation for operation as the reference does in Python floats: bit-identical frames. */
int vs_moving_mnist_batch(const uint8_t* digits, int64_t n_digits_total, int digit_h, int digit_w, const int32_t* init, int batch,
                          int num_digits, int seq_len, int frame_size, void* out, int out_dtype, void* stream);

/* Decoder input of the auto-encoding pair and of every rollout step in one launch (mlp_encdec.py:43-48 mixing applied at
 * model.py:74-83): z [B, 1+n, Cz] = mix(s [B, Cs], [t_rand [B, Ct] ; t_codes [B, n, Ct]]), mixing 0 = concat (Cz = Cs + Ct),
 * 1 = mul (Cz = Cs = Ct); out fp32, out_bf16 (may be NULL) the same values rounded for the decoder's first bf16 GEMM.
 * bwd: dz [B, 1+n, Cz] fp32 -> ds [B, Cs] (summed over the frames), dt_rand [B, Ct], dt_codes [B, n, Ct].                   */
int vs_mix_codes_fwd(const float* s, const float* t_rand, const float* t_codes, int64_t B, int n, int Cs, int Ct, int mixing, float* out,
                     void* out_lowp, int lowp_dtype, void* stream);
int vs_mix_codes_bwd(const float* dz, const float* s, const float* t_rand, const float* t_codes, int64_t B, int n, int Cs, int Ct,
                     int mixing, float* ds, float* dt_rand, float* dt_codes, void* stream);

/* All four training losses and their weighted sum in one pass (train.py:117-149, MLP-family layout):
 *   total = l_ae mse(frames[:,0], full[:,idx[0]]) + l_s mean((s_old-s_new)^2) + l_pred mse(frames[:,1:], full[:,idx[1:]])
 *           + l_t t_reg,   t_reg = 0.5 mean_b sum_c t0^2 (average_tloss: 0.5 mean_{b,c} t0^2);  lambdas = {l_ae, l_s, l_t, l_pred}
 *   (host).  n_s = number of spatial-code elements (0: no spatial term, s_old/s_new may be NULL).  Target frames: idx [G] on the
 *   device, or idx = NULL and frame 0 <-> full[:, t_random_dev[0] - ae_shift], frame g <-> full[:, first_forecast + g - 1] (the
 *   random window end of train.py:72-75 read on the device: no index tensor to build per step).
 *   fwd: out [10] floats on the device: [4] total, [5] ae, [6] zero-order, [7] pred, [8] t_reg ([0..3], [9] scratch).
 *   bwd: grad_total = upstream gradient of `total` (1 float ON THE DEVICE); writes dframes [B,G,D], ds_old, ds_new [n_s], dt0.  */
int vs_train_losses_fwd(const float* frames, const float* full, const int32_t* idx, const int32_t* t_random_dev, int ae_shift,
                        int first_forecast, int64_t B, int G, int T, int64_t D,
                        const float* s_old, const float* s_new, int64_t n_s, const float* t0, int64_t Bt, int64_t Ct,
                        int average_tloss, const float* lambdas, float* out, void* stream);
int vs_train_losses_bwd(const float* frames, const float* full, const int32_t* idx, const int32_t* t_random_dev, int ae_shift,
                        int first_forecast, int64_t B, int G, int T, int64_t D,
                        const float* s_old, const float* s_new, int64_t n_s, const float* t0, int64_t Bt, int64_t Ct,
                        int average_tloss, const float* lambdas, const float* grad_total, float* dframes, float* ds_old,
                        float* ds_new, float* dt0, int frames_act, void* dz, int dz_dtype, void* stream);
/* Both of the above in ONE pass over the frames, for an upstream gradient known when the forward runs (a recorded training step
 * passes its resident 1.0 or loss scale: `grad_total`, one float on the device): `out` as vs_train_losses_fwd; dz (the gradient of
 * the producing chain's pre-activation, compute dtype) / ds_old / ds_new / dt0 as vs_train_losses_bwd writes them.  D % 4 == 0.
 * `out` must hold 16 + 2 * 4096 floats here: the frame sums go through per-workgroup partials (no float atomics: reproducible). */
int vs_train_losses_fwd_grad(const float* frames, const float* full, const int32_t* idx, const int32_t* t_random_dev, int ae_shift,
                             int first_forecast, int64_t B, int G, int T, int64_t D, const float* s_old, const float* s_new, int64_t n_s,
                             const float* t0, int64_t Bt, int64_t Ct, int average_tloss, const float* lambdas, float* out,
                             const float* grad_total, float* ds_old, float* ds_new, float* dt0, int frames_act, void* dz, int dz_dtype,
                             void* stream);

/* The decoder's last Linear layer with the frame losses in its epilogue -- replaces, in the recorded MLP-family step, the pair
 * [last vs_gemm of MLPDecoder.forward (networks/mlp_encdec.py:43-50) ; vs_train_losses_fwd_grad (train.py:85-86, 117-149)]:
 *   frames[r = (b, g), :] = act(A[r, :] W^T + bias) is compared with full[b, target(g), :] while in registers and never stored;
 *   dz [M, N] (dz_dtype) = grad_total * k_g * (frames - target) * act'(frames), ds_old / ds_new / dt0 and out[0..8] as
 *   vs_train_losses_fwd_grad (out holds 16 + 2 * 4096 floats; the frame sums are added from one pair per workgroup, in order).
 * A [M = B * G, K] and W [N, K] 16-bit row-major (K contiguous), bias fp32 [N] or NULL, N % 4 == 0, target frames resolved on the
 * device from t_random_dev (frame 0 <-> t - ae_shift, frame g <-> first_forecast + g - 1).  VS_ERR_UNSUPPORTED when the problem does
 * not run on the 256 x 256 tile kernel: store the frames with vs_gemm and call vs_train_losses_fwd_grad.                          */
int vs_gemm_frame_loss(int compute, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* W, int64_t ldw,
                       const float* bias, int act, const float* full, const int32_t* t_random_dev, int ae_shift, int first_forecast,
                       int G, int T, const float* s_old, const float* s_new, int64_t n_s, const float* t0, int64_t Bt, int64_t Ct,
                       int average_tloss, const float* lambdas, const float* grad_total, void* dz, int dz_dtype, float* ds_old,
                       float* ds_new, float* dt0, float* out, void* stream);

/* Adam update of up to 64 fp32 tensors in one launch (reference: train.py:156-158 `optimizer.step()` on
 * torch.optim.Adam(lr, betas): weight_decay 0, amsgrad off; same operation order as torch's single-tensor path:
 *   m += (1-b1)(g-m); v = b2 v + (1-b2) g g; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),  t = step[0] + 1).
 * `step` is an int32 ON THE DEVICE (capturable into a hipGraph); it is NOT modified here -- several vs_adam_multi calls
 * of one optimizer step (more than 64 tensors) share it, then vs_adam_step_increment bumps it once.
 * skipped (host array or NULL): optimizer steps a tensor sat out without gradient -- torch counts steps per parameter, so
 * its t is step[0] + 1 - skipped[i].
 * shadow_bf16 (array or NULL, entries may be NULL): bf16 copy of the updated parameter written in the same pass (the
 * operand copy the bf16 forward pass reads).  Tensors contiguous; 16-byte aligned ones take the vector path.
 * grad_dtype (host array or NULL = all fp32): VS_BF16 entries read their gradient as bf16 -- the data-parallel wire image of a
 * weight gradient, consumed without a cast back to fp32.                                                                  */
int vs_adam_multi(int n_tensors, float* const* params, const void* const* grads, const int32_t* grad_dtype, float* const* exp_avg,
                  float* const* exp_avg_sq, void* const* shadow_bf16, const int64_t* numel, const int32_t* skipped, int32_t* step,
                  double lr, double beta1, double beta2, double eps, void* stream);
int vs_adam_step_increment(int32_t* step, void* stream);

/* fp16 training with dynamic loss scaling (reference: train.py:96-97 `scaler = torch.cuda.amp.GradScaler()`, train.py:151-155
 * `scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()` under --torch_amp), kept on the device so that the
 * whole step stays recordable into a hipGraph.  `scale_state` is 4 fp32 words on the device:
 *   [0] loss scale S, [1] found_inf of the current step, [2] growth tracker (clean steps since the last change), [3] steps skipped.
 * vs_check_finite_multi : found_inf[0] = 1 if any element of any listed gradient (fp32 or 16-bit) is inf / NaN   (GradScaler.unscale_'s check)
 * vs_adam_multi_scaled  : vs_adam_multi with gradients read as g / S and the whole update suppressed when found_inf is set;
 *                         shadow copies are written as `shadow_dtype` (VS_BF16 | VS_F16).  scale_state NULL = plain vs_adam_multi.
 * vs_adam_step_increment_scaled : the step count does not advance on a skipped step.
 * vs_loss_scale_update  : GradScaler.update(): S *= backoff and tracker = 0 on overflow, else tracker += 1 and S *= growth every
 *                         `growth_interval` clean steps; clears found_inf.                                                        */
int vs_check_finite_multi(int n_tensors, const void* const* grads, const int32_t* grad_dtype, const int64_t* numel, float* found_inf,
                          void* stream);
int vs_adam_multi_scaled(int n_tensors, float* const* params, const void* const* grads, const int32_t* grad_dtype, float* const* exp_avg,
                         float* const* exp_avg_sq, void* const* shadow, int shadow_dtype, const int64_t* numel, const int32_t* skipped,
                         int32_t* step, double lr, double beta1, double beta2, double eps, const float* scale_state, void* stream);
int vs_adam_step_increment_scaled(int32_t* step, const float* scale_state, void* stream);
/* Caps the grid of the following vs_adam_multi* launches of this process at `blocks` workgroups (0: the default, 4096), returns
 * the previous cap.  A bucket updated WHILE backward is still running is launched with a small grid (2 workgroups per CU) so
 * that it streams HBM in the background and leaves the wave slots to the GEMMs on the critical path.                         */
int vs_adam_set_max_blocks(int blocks);
/* Weight gradient + Adam in one launch (replaces, for one 2-D parameter, the weight-gradient vs_gemm of a Linear layer -- reference
 * mlp.py:66-75 backward -- followed by its share of optimizer.step(), train.py:156-158):  G = alpha * A * B^T with the operand
 * conventions of vs_gemm is the gradient of the contiguous fp32 parameter `param` [M, N]; the epilogue updates param / exp_avg /
 * exp_avg_sq exactly as vs_adam_multi would from a stored G (`step` = the group's device step count BEFORE the increment,
 * `skipped` as there) and rewrites the 16-bit operand copy `shadow` (NULL: none).  G is never stored.  16-bit compute types;
 * operands must fit the LDS-DMA loader (16-byte aligned, leading dimensions / extents multiples of 8): VS_ERR_UNSUPPORTED
 * otherwise and nothing is launched.                                                                                          */
int vs_gemm_adam(int compute, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, int layout_a, const void* B, int64_t ldb,
                 int layout_b, float alpha, float* param, float* exp_avg, float* exp_avg_sq, void* shadow, int shadow_dtype,
                 const int32_t* step, int32_t skipped, double lr, double beta1, double beta2, double eps, void* stream);
int vs_loss_scale_update(float* scale_state, float growth_factor, float backoff_factor, int growth_interval, void* stream);

/* Fused frame losses (train.py:85-86 ae_loss MSE and train.py:139 forecast MSE in one pass over the decoded frames).
 * frames [B, G, D] fp32: per sample the auto-encoding reconstruction (g = 0) followed by the G-1 forecasts; full [B, T, D]
 * fp32: every observed frame; idx [G] int32 ON THE DEVICE: frame g is compared with full[:, idx[g]].
 *   fwd: sums[0] = sum of squared errors of frame 0, sums[1] = of frames 1..G-1 (float atomics: last-bit nondeterminism).
 *   bwd: dframes[b, g, :] = coef[g == 0 ? 0 : 1] * (frames - target); coef [2] on the device = 2/N_k times the upstream
 *        gradient of loss k, so no host synchronisation is needed.  D must make rows 16-byte aligned for speed.          */
int vs_frames_sse_fwd(const float* frames, const float* full, const int32_t* idx, int64_t B, int G, int T, int64_t D, float* sums,
                      void* stream);
int vs_frames_sse_bwd(const float* frames, const float* full, const int32_t* idx, int64_t B, int G, int T, int64_t D, const float* coef,
                      float* dframes, void* stream);

/* Round 4: the conv families' code losses and the weighted total in one pass (reference train.py:38-42 zero_order_loss -- with skip connections
 * over the code and every skip tensor --, :141-149 t_reg and total): pairs (a[j], b[j]) of count[j] elements (a multiple of 8, 16-byte aligned,
 * fp32 or 16-bit: read where they lie, no concatenation), t0 [t_count] fp32, the raw sums of vs_frames_sse_fwd for the two frame terms.
 *   _fwd   out[0..4] = total, ae, zero, pred, t_reg; partial: vs_code_losses_chunks() floats (per-chunk sums, finished in a fixed order)
 *   _bwd   for the upstream gradient g[0] (device): da[j] / db[j] (NULL: not wanted) in dtype[j], dt0, and coefs[4] = the coefficient pairs
 *          vs_frames_sse_bwd takes for the auto-encoding and the forecast frame stacks.  One launch. */
/* Round 4: decoder inputs of a batched rollout (reference conv.py:228, 388-394: torch.cat([skip, out], 1) with the skip / spatial code of the B
 * sequences shared by the n frame calls that run as one batch): out [n B][Ca + Cb][HW] = cat(a [B][Ca][HW] repeated over the frames, x [n B][Cb][HW])
 * in out's type, one pass (no repeated copy of a); _bwd: da = sum over the frames of dout's first Ca channels (frame order), dx = the others.
 * HW a multiple of 8, tensors 16-byte aligned; da or dx may be NULL. */
int vs_cat_bcast_fwd(const void* a, int a_dtype, const void* x, int x_dtype, void* out, int out_dtype, int B, int n, int Ca, int Cb, int64_t HW, void* stream);
int vs_cat_bcast_bwd(const void* dout, int dout_dtype, void* da, int a_dtype, void* dx, int x_dtype, int B, int n, int Ca, int Cb, int64_t HW, void* stream);
int64_t vs_code_losses_chunks(int n_pairs, const int64_t* count, int64_t t_count);
int vs_code_losses_fwd(int n_pairs, const void* const* a, const void* const* b, const int* dtype, const int64_t* count, const float* t0, int64_t t_count,
                       const float* sse_ae, const float* sse_pred, float scale_ae, float scale_pred, float l_ae, float l_s, float l_pred, float l_t,
                       float inv_s, float inv_t, float* partial, float* out, void* stream);
int vs_code_losses_bwd(int n_pairs, const void* const* a, const void* const* b, void* const* da, void* const* db, const int* dtype, const int64_t* count,
                       const float* t0, float* dt0, int64_t t_count, const float* g, float scale_ae, float scale_pred, float l_ae, float l_s, float l_pred,
                       float l_t, float inv_s, float inv_t, float* coefs, void* stream);

/* dz[i] = dy[i] * act'(y[i]) evaluated from the activation OUTPUT y (see vs_gemm mask semantics).
 * Backward of the trailing activation of a chain (mlp_encdec.py:49 last_activation, conv.py:230).  */
int vs_act_bwd(const void* dy, int dy_dtype, const void* y, int y_dtype, void* dz, int dz_dtype, int act,
               int64_t n, void* stream);

/* y[i] = act(x[i]) (out of place or in place).  networks/utils.py:50-72.                             */
int vs_act_fwd(const void* x, int x_dtype, void* y, int y_dtype, int act, int64_t n, void* stream);

/* dst[c*rows + r] = (dst_dtype) src[r*cols + c]: transposed (and converted) weight copies for the backward rollout. */
int vs_transpose_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int rows, int cols, void* stream);

/* ------------------------------------------------------------------------------------------------
 * vs_mlp_rollout_{fwd,bwd}: the residual latent integrator rolled over time in one persistent launch.
 *
 * Replaces the python loop of SeparableNetwork.get_forecast (networks/model.py:78-83) around
 * MLPResnet.forward / MLPResBlock.forward (networks/resnet.py:22-50):
 *     for t in 1..n-1:  for b in blocks:  r = W3_b relu(W2_b relu(W1_b x + b1_b) + b2_b) + b3_b ;  x = x + r
 * Forward:
 *   x0        [B, C] fp32         initial temporal code (E_t output)
 *   weights   host array of 3*n_blocks device pointers {W1 [H,C], W2 [H,H], W3 [C,H]} per block, each PRE-PACKED by
 *             vs_pack_rollout_weight in the compute type;  biases: host array of 3*n_blocks fp32 device pointers
 *   t_codes   [B, n, C] fp32      every code of the rollout, t_codes[:, 0] = x0 (the layout get_forecast returns)
 *   residuals [n-1, n_blocks, B, C] fp32 or NULL   (the `t_residuals` the reference returns)
 *   xin_save [nb, n-1, B, C], h1_save / h2_save [nb, n-1, B, H]  compute type: inputs of the weight-gradient GEMMs
 *   m1_save / m2_save [nb, n-1, Bp, P, 32] uint32, Bp = B rounded up to 16, P = vs_mlp_rollout_parts(...): ReLU sign
 *             bits of h1 / h2, opaque to the caller (written by _fwd, read by _bwd of the same sizes), so the backward
 *             kernel reads a few words per step instead of the activations
 *   workspace: exchange area of vs_mlp_rollout_workspace_bytes(...) bytes, ZERO-FILLED ONCE by the caller when it creates it and then
 *             left to the library (the weight-stationary form numbers its launches through per-slab epoch words inside it and
 *             never clears it; the slab form clears it itself; an all-zero area is always a valid state); its last 16 bytes hold a
 *             timeout flag.  bf16, C <= 32, H in {128, 256, 512}, ceil(B/16) * n_blocks * H/64 <= 224: weight-stationary
 *             pipelined form -- one workgroup per (16-row slab, block, 64-column part) keeps its weight fragments in
 *             registers for the whole rollout and hands [16, C] partials to the workgroups of the next block through
 *             the exchange area.  Otherwise the slab form:  When the hidden size
 *             allows it the H x H layer of every 16-row slab is split over P workgroups (P CUs stream 1/P of the
 *             weights each); they all-reduce one [16, C] partial per block-step through this area with epoch-tagged
 *             8-byte granules (placement independent, bounded spins).  NULL / too small => P = 1 (no split).
 * Backward (through time):
 *   grad_t_codes [B, n, C] fp32   gradient wrt every code;  weights_t: host array of 3*n_blocks device pointers
 *             {W3^T [H,C], W2^T [H,H], W1^T [C,H]} per block, packed with vs_pack_rollout_weight(transpose = 1)
 *   dx0 [B, C] fp32; dr_save [nb, n-1, B, C], dh2_save / dh1_save [nb, n-1, B, H] compute type.
 *   Weight gradients are then vs_gemm(dh1_save[b] (S), xin_save[b] (S)) etc. with K = (n-1)*B, bias gradients
 *   vs_colsum of the same buffers.
 * P workgroups per 16 batch rows; partial sums are combined in a fixed order: results are bitwise reproducible.
 * Limits: n_blocks <= 8; LDS footprint (grows with H) must fit 160 KiB (H <= ~2048 in bf16).
 */
/* Pre-pack of one integrator weight for the rollout kernels: the logical matrix L[N][K] (L = src if transpose == 0,
 * with src fp32 [N,K];  L = src^T if transpose == 1, with src fp32 [K,N]) is converted to the compute type and laid
 * out in MFMA-fragment order, one contiguous 1 KiB piece per (16-column tile, k-step), zero padded, so that every
 * wave-level weight load of the rollout is a single fully coalesced 1 KiB read.  dst must hold
 * vs_rollout_packed_elems(compute, N, K) elements of the compute type.  Re-run after each optimizer step.          */
size_t vs_rollout_packed_elems(int compute, int N, int K);
int vs_pack_rollout_weight(int compute, const float* src, int transpose, int N, int K, void* dst, void* stream);
/* The same for several weights in one launch (<= 48 jobs): src[i] fp32 [rows, cols], logical L = src or src^T [N[i], K[i]]. */
int vs_pack_rollout_weights(int compute, int n_jobs, const float* const* src, const int* transpose, const int* N, const int* K,
                            void* const* dst, void* stream);

int vs_mlp_rollout_parts(int compute, int B, int C, int H);
size_t vs_mlp_rollout_workspace_bytes(int compute, int B, int C, int H);
/* Round 5: the exchange mode of the weight-stationary integrator (reference resnet.py:22-50 unrolled over time, model.py:74-86) is a
 * START-UP decision.  With 8 / 16 / ... row slabs the ring of a slab publishes its granules with plain stores that only reach ONE XCD's
 * L2: it leans on a dispatch property HIP does not promise (workgroups 8 apart share an XCD).
 *   _get   1 when a launch of this geometry would take the XCD-local stores now (VS_ROLLOUT_XCD_LOCAL != 0, 8 k slabs, not refused)
 *   _set   0: agent-scope (sc1) stores for the rest of the process -- what the caller does when its probe rollout (a first launch with a
 *          short spin limit on a fresh workspace) came back with the error word raised; 1: allow again
 * VS_ROLLOUT_XCD_LOCAL=2 (test aid) keeps the plain stores but deals the slabs out so that every ring is spread over the XCDs.            */
int vs_mlp_rollout_xcd_local_get(int compute, int B, int C, int H, int n_blocks);
int vs_mlp_rollout_xcd_local_set(int allowed);
/* The exchange guard: ONE 4-byte device word (caller-owned, zero = fine) registered for the process.  While registered, every kernel with
 * a bounded in-launch exchange (vs_mlp_rollout_fwd / _bwd: bit 1; vs_conv3_img16_bn_fwd / _bwd: bit 2) raises this word on a time-out
 * INSTEAD of the word inside its own workspace, and every optimizer launch issued afterwards (vs_adam_multi, vs_adam_multi_scaled,
 * vs_gemm_adam, vs_adam_step_increment*) reads it first and leaves parameters, moments, operand copies and the step count untouched
 * while it is non-zero (the pointer is taken when a launch is issued or recorded): `optimizer.step()` (reference train.py:156-158) never
 * applies an update computed from a timed-out exchange.  The caller reads and clears the word (it is sticky).  NULL unregisters.        */
int vs_exchange_guard_set(void* word);
void* vs_exchange_guard_get(void);
/* Optional companion of the guard: a second caller-owned 4-byte device word that vs_adam_step_increment* bumps by one every time the guard
 * made it skip a step (reference train.py:156-158: `optimizer.step()` did not happen), so that the host can report HOW MANY steps a sticky
 * time-out cost between two reads.  NULL unregisters; the caller reads and clears it.                                                     */
int vs_exchange_skip_counter_set(void* word);
int vs_mlp_rollout_fwd(int compute, int B, int C, int H, int n_blocks, int n_steps, const float* x0,
                       const void* const* weights, const float* const* biases, float* t_codes, float* residuals,
                       void* xin_save, void* h1_save, void* h2_save, uint32_t* m1_save, uint32_t* m2_save, void* workspace,
                       size_t workspace_bytes, void* stream);

int vs_mlp_rollout_bwd(int compute, int B, int C, int H, int n_blocks, int n_steps, const float* grad_t_codes,
                       const void* const* weights_t, const void* h1_save, const void* h2_save,
                       const uint32_t* m1_save, const uint32_t* m2_save, float* dx0,
                       void* dr_save, void* dh2_save, void* dh1_save, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Convolutions as im2col-free implicit GEMMs on NCHW tensors (csrc/vs_conv.hip).  x [B,Cin,H,W], stride/pad equal
 * in both directions, dilation 1, groups 1 -- the only forms the reference uses.  Storage type of x / w / dy is the
 * compute type; biases and weight gradients are fp32.
 *
 *   vs_conv2d_*            nn.Conv2d          w [Cout,Cin,kh,kw]   y [B,Cout,OH,OW], OH = (H + 2 pad - kh)/stride + 1
 *                          conv.py:119-122 (k4 s2 p1), :147-165,300-317,326-343,362-382,402-417 and resnet.py:57-59
 *                          (k3 s1 p1), :170 (k4 s1 p0)
 *   vs_conv_transpose2d_*  nn.ConvTranspose2d w [Cin,Cout,kh,kw]   y [B,Cout,OH,OW], OH = (H - 1) stride - 2 pad + kh
 *                          conv.py:258,295 (k4 s1 p0), :260-263 (k4 s2 p1), :318 (k3 s1 p1)
 *   *_fwd   : y = conv(x, w) + bias (bias may be NULL)
 *   *_dgrad : dx = gradient wrt x given dy (dy has the shape of y)
 *   The two TRANSPOSED-form contractions (vs_conv_transpose2d_fwd, vs_conv2d_dgrad) take `w_packed`, the weight
 *   pre-packed by vs_conv_pack_weight (channel dims swapped, taps grouped per output-parity phase for stride 2,
 *   converted to the compute type; refresh it after each optimizer step).  Supported there: stride 1 (any k, pad) and
 *   stride 2 with k - 2*pad == 2 (the reference's k4 s2 p1), run as 4 dense parity phases.
 *   *_wgrad : dw (fp32, same shape as w) = gradient wrt w; uses split-K when the reduction B*OH*OW is long: pass a
 *             workspace of vs_conv_wgrad_workspace_bytes(...) bytes (NULL = no split, slower but correct)
 * In every call B, Cin, H, W, Cout describe the FORWARD op (x's shape and the weight's channel counts).
 */
/* Workspace for every conv entry point below (forward, input gradient, weight gradient of Conv2d(Cin, Cout, k, stride, pad) on
 * [B, Cin, H, W] and of the ConvTranspose2d with the same numbers).  With a workspace of this size, contractions with >= 64
 * output channels run as "gather once into a transient column matrix, then dense MFMA GEMM" (2-3x faster than gathering
 * inside the MFMA loop); with workspace = NULL (or too small) they gather inside the loop.  The result is the same either way. */
size_t vs_conv_workspace_bytes(int compute, int B, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad);
size_t vs_conv_wgrad_workspace_bytes(int B, int Cin, int OH, int OW, int Cout, int kh, int kw);
/* w: fp32 master weight [D0][D1][kh][kw] (Conv2d: D0 = Cout, D1 = Cin; ConvTranspose2d: D0 = Cin, D1 = Cout);
 * dst: vs_conv_packed_elems(...) elements of the compute type.                                                     */
size_t vs_conv_packed_elems(int D0, int D1, int kh, int kw, int stride, int pad);
int vs_conv_pack_weight(int compute, const float* w, int D0, int D1, int kh, int kw, int stride, int pad, void* dst, void* stream);
int vs_conv2d_fwd(int compute, const void* x, const void* w, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int W,
                  int Cout, int kh, int kw, int stride, int pad, void* workspace, size_t workspace_bytes, void* stream);
int vs_conv2d_dgrad(int compute, const void* dy, const void* w_packed, void* dx, int dx_dtype, int B, int Cin, int H, int W, int Cout, int kh,
                    int kw, int stride, int pad, void* workspace, size_t workspace_bytes, void* stream);
int vs_conv2d_wgrad(int compute, const void* dy, const void* x, float* dw, int B, int Cin, int H, int W, int Cout, int kh, int kw,
                    int stride, int pad, void* workspace, size_t workspace_bytes, void* stream);
/* The weight gradients with an `accumulate` flag: dw += (non-zero) instead of dw =.  A module applied many times per step
 * (ConvResBlock inside the rollout loop of model.py:78-83: one call per predicted frame) adds each call's contribution in the
 * GEMM / split-K epilogue instead of a separate add launch per call and parameter.                                          */
int vs_conv2d_wgrad_acc(int compute, const void* dy, const void* x, float* dw, int B, int Cin, int H, int W, int Cout, int kh, int kw,
                        int stride, int pad, void* workspace, size_t workspace_bytes, int accumulate, void* stream);
int vs_conv_transpose2d_wgrad_acc(int compute, const void* dy, const void* x, float* dw, int B, int Cin, int H, int W, int Cout, int kh,
                                  int kw, int stride, int pad, void* workspace, size_t workspace_bytes, int accumulate, void* stream);
int vs_conv_transpose2d_fwd(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H,
                            int W, int Cout, int kh, int kw, int stride, int pad, void* workspace, size_t workspace_bytes, void* stream);
int vs_conv_transpose2d_dgrad(int compute, const void* dy, const void* w, void* dx, int dx_dtype, int B, int Cin, int H, int W, int Cout,
                              int kh, int kw, int stride, int pad, void* workspace, size_t workspace_bytes, int cols_from_wgrad, void* stream);
int vs_conv_transpose2d_wgrad(int compute, const void* dy, const void* x, float* dw, int B, int Cin, int H, int W, int Cout, int kh,
                              int kw, int stride, int pad, void* workspace, size_t workspace_bytes, void* stream);

/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) on `planes` = B*C planes of H x W (reference: conv.py:517, the ResNet18
 * stem of the chairs encoder); output (H + 2 - 3) / 2 + 1 per side.  Backward gathers over the overlapping windows (arg-max =
 * first maximum in window order, as ATen), no atomics.                                                                  */
int vs_maxpool3s2_fwd(const void* x, int x_dtype, void* y, int y_dtype, int64_t planes, int H, int W, void* stream);
int vs_maxpool3s2_bwd(const void* x, int x_dtype, const void* dy, int dy_dtype, void* dx, int dx_dtype, int64_t planes, int H, int W,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm2d + activation, pooling, upsampling (csrc/vs_norm.hip); x is NCHW [B,C,HW], dtypes VS_F32 | VS_BF16.
 *
 * GROUPS: the batch of B samples is `groups` consecutive groups of B/groups samples and every group is normalised
 *                 with its own statistics -- one group per reference CALL.  This is what lets the n+1 decoder calls of
 *                 a training step (model.py:74-83) run as one batch over time with exactly the reference's per-call
 *                 BatchNorm semantics.  mean / invstd / dgamma / dbeta are [groups, C] (sum dgamma/dbeta over groups).
 * vs_bn_stats   : per-call batch statistics of nn.BatchNorm2d in training mode (conv.py:41-60): mean[g][c], invstd[g][c]
 *                 = 1/sqrt(biased var + eps); when running_mean/var are non-NULL they are updated in place, group by
 *                 group in call order, with `momentum` and the unbiased variance, exactly like `groups` sequential ATen
 *                 calls (var_scratch: [groups, C] floats).
 * vs_bn_act_fwd : y = act(gamma (x - mean) invstd + beta).  Eval mode: pass running_mean and 1/sqrt(running_var+eps).
 * vs_bn_act_bwd : given dy = dL/dy, recomputes z = gamma xhat + beta, dz = dy act'(z), and returns dbeta = sum dz,
 *                 dgamma = sum dz xhat and dx = gamma invstd (dz - dbeta/N - xhat dgamma/N)  (training != 0) or
 *                 dx = gamma invstd dz (training == 0).
 * vs_chan_sum   : out[c] = sum_{b,pix} x[b,c,pix]   (bias gradient of a convolution)
 * vs_maxpool2_* : nn.MaxPool2d(2,2) (conv.py:151-169,330-335) on `planes` = B*C planes of H x W (even); backward
 *                 routes dy to the first maximum in window scan order, like ATen.
 * vs_upsample2_*: nn.Upsample(scale_factor=2, mode='nearest') (conv.py:296-314,371-377,406-413); H, W are the INPUT size.
 */
int vs_bn_stats(const void* x, int x_dtype, int B, int C, int64_t HW, int groups, float* mean, float* invstd, float* var_scratch,
                float* running_mean, float* running_var, float momentum, float eps, void* stream);
/* Training-mode BatchNorm2d + activation of ONE reference call on a small tensor (slabs B*HW <= 8192 per channel: the SST
 * integrator's 8 x 16 x 16 maps, conv.py:41-60 blocks inside resnet.py:53-88) in one launch: statistics, running-statistics update
 * (momentum, unbiased variance; NULL = not tracked), y = act(gamma * x_hat + beta); mean / invstd [C] are kept for vs_bn_act_bwd.
 * vs_bn_train_fwd_small_supported tells whether a tensor is served (else: vs_bn_stats + vs_bn_act_fwd, same results up to summation
 * order).  vs_bn_act_bwd takes the matching one-launch path by itself.                                                      */
int vs_bn_train_fwd_small_supported(int x_dtype, int B, int C, int64_t HW);
int vs_bn_train_fwd_small(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma, const float* beta, int act, float* mean,
                          float* invstd, float* running_mean, float* running_var, float momentum, float eps, int B, int C, int64_t HW,
                          void* stream);
/* the same for `groups` calls stacked along the batch axis (per-call statistics, mean / invstd [groups][C]); var_scratch [groups][C] is needed when
 * running statistics are tracked (they are folded in call order by a second launch) */
int vs_bn_train_fwd_small_groups(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma, const float* beta, int act, float* mean,
                                 float* invstd, float* var_scratch, float* running_mean, float* running_var, float momentum, float eps, int B, int C,
                                 int64_t HW, int groups, void* stream);
/* vs_bn_train_fwd_small on the split slabs of vs_conv3_img16: z = round(sum of the slabs + bias[c]) in the 16-bit z_dtype is written
 * (vs_bn_act_bwd needs it), everything else as above.  skip / xnew (fp32, both or neither) and xnew16 (z_dtype, optional): the tail of
 * a residual block (ConvResBlock.forward resnet.py:66-70) -- xnew = skip + y and its 16-bit copy for the next block.           */
int vs_bn_train_fwd_small_slabs(const float* slabs, int nslabs, const float* bias, void* z, int z_dtype, void* y, int y_dtype, const float* gamma,
                                const float* beta, int act, float* mean, float* invstd, float* running_mean, float* running_var,
                                float momentum, float eps, const float* skip, float* xnew, void* xnew16, int B, int C, int64_t HW,
                                void* stream);
/* Backward of the same block in one launch (training mode, one call, 16-bit z): the upstream gradient is the sum of split slabs
 * (slabs != NULL: what a vs_conv3_img16 input-gradient launch left) or dy_a (+ dy_b, fp32: the two outputs of a residual block,
 * resnet.py:66-70); accumulate != 0 adds d gamma / d beta to the vectors given (the parameter's pending gradient of this pass).  */
int vs_bn_act_bwd_small_ex(const void* dy_a, int dy_a_dtype, const float* dy_b, const float* slabs, int nslabs, const void* z, int z_dtype,
                           const float* mean, const float* invstd, const float* gamma, const float* beta, int act, float* dgamma,
                           float* dbeta, int accumulate, void* dx, int dx_dtype, int B, int C, int64_t HW, void* stream);
/* Round 4: training-mode BatchNorm2d (+ activation) forward with every (call group, channel) slab read ONCE (reference conv.py:41-60): slabs of
 * 8 193 .. 131 072 16-bit elements live in the registers of a 1024-thread workgroup between the statistics and the apply phase (vs_bn_stats +
 * vs_bn_act_fwd read the tensor twice).  mean / invstd [groups][C]; running estimates folded in call order (groups > 1: var_scratch [groups][C]).
 * The backward counterpart (x and dy resident, slabs up to 65 536 elements) is taken by vs_bn_act_bwd / _gsum by itself.  VS_BN_SLAB=0: never. */
int vs_bn_train_fwd_slab_supported(int x_dtype, int Bg, int C, int64_t HW);
int vs_bn_train_fwd_slab(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma, const float* beta, int act, float* mean, float* invstd,
                         float* var_scratch, float* running_mean, float* running_var, float momentum, float eps, int B, int C, int64_t HW, int groups,
                         void* stream);
int vs_bn_act_fwd(const void* x, int x_dtype, void* y, int y_dtype, const float* mean, const float* invstd, const float* gamma,
                  const float* beta, int act, int B, int C, int64_t HW, int groups, void* stream);
int vs_bn_act_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* invstd, const float* gamma,
                  const float* beta, int act, int training, int groups, float* dgamma, float* dbeta, void* dx, int dx_dtype, int B,
                  int C, int64_t HW, void* stream);
/* vs_bn_act_bwd that additionally returns the PARAMETER gradients, i.e. dgamma / dbeta summed over the call groups ([C] each; with the
 * n + 1 decoder calls of a step stacked, nn.BatchNorm2d's weight receives the sum of n + 1 per-call gradients, conv.py:41-60): written
 * by workgroup 0 of the apply pass (or one small launch behind the one-launch form) instead of two reduction launches by the caller.   */
int vs_bn_act_bwd_gsum(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, int act, int training, int groups, float* dgamma, float* dbeta, void* dx, int dx_dtype, int B,
                       int C, int64_t HW, float* dgamma_sum, float* dbeta_sum, void* stream);
/* vs_bn_stats + vs_bn_act_fwd with the running-statistics fold moved INTO the apply pass (one launch less per BatchNorm call): vs_bn_stats_ub leaves the
 * unbiased variances in ubvar [groups][C]; vs_bn_act_fwd_running applies mean / invstd and folds (mean, ubvar) into running_mean / running_var in call order */
int vs_bn_stats_ub(const void* x, int x_dtype, int B, int C, int64_t HW, int groups, float* mean, float* invstd, float* ubvar, float eps, void* stream);
int vs_bn_act_fwd_running(const void* x, int x_dtype, void* y, int y_dtype, const float* mean, const float* invstd, const float* gamma, const float* beta,
                          int act, int B, int C, int64_t HW, int groups, const float* ubvar, float* running_mean, float* running_var, float momentum,
                          void* stream);
int vs_chan_sum(const void* x, int x_dtype, int B, int C, int64_t HW, float* out, void* stream);
/* the same sum with per-chunk partial sums in a workspace of vs_chan_sum_workspace_bytes: no atomics (a 1-channel map serialises 1024 of them on
 * one address), fixed summation order */
size_t vs_chan_sum_workspace_bytes(int B, int C, int64_t HW);
int vs_chan_sum_ws(const void* x, int x_dtype, int B, int C, int64_t HW, void* ws, size_t ws_bytes, float* out, void* stream);
int vs_maxpool2_fwd(const void* x, int x_dtype, void* y, int y_dtype, int64_t planes, int H, int W, void* stream);
int vs_maxpool2_bwd(const void* x, int x_dtype, const void* dy, int dy_dtype, void* dx, int dx_dtype, int64_t planes, int H, int W,
                    void* stream);
int vs_upsample2_fwd(const void* x, int x_dtype, void* y, int y_dtype, int64_t planes, int H, int W, void* stream);
int vs_upsample2_bwd(const void* dy, int dy_dtype, void* dx, int dx_dtype, int64_t planes, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VARSEP_HIP_H */
