/*
 * libt2h_hip.so -- C ABI of the MI355X (gfx950) kernels behind the Text2Human
 * sampling hot path (sample_from_parsing / sample_from_pose).
 *
 * The reference (yumingj/Text2Human) is pure PyTorch and has NO FFI / operator
 * plug-in interface; every entry point below replaces a group
 * of eager PyTorch op call sites, cited as reference file:line.  The host side
 * (text2human_amd python package) binds these with ctypes; INTEGRATION.md shows the stub
 * a reference maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless named host_*;
 *  - `stream` is a hipStream_t passed as void*; all calls are asynchronous on
 *    it, never synchronise and never allocate;
 *  - activations of the conv stacks are NHWC fp32 ("pixel rows": [B*H*W, C]),
 *    transformer activations are [B*T, C] fp32, indices are int64 like the
 *    reference's LongTensors, masks are uint8;
 *  - every entry returns 0 on success or a negative t2h_status; the message
 *    for the calling thread is available from t2h_last_error().
 */
#ifndef T2H_HIP_H
#define T2H_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum t2h_status {
  T2H_OK = 0,
  T2H_ERR_INVALID = -1,   /* bad argument (shape / alignment / NULL) */
  T2H_ERR_LAUNCH = -2,    /* hipLaunch / runtime error */
  T2H_ERR_UNSUPPORTED = -3
};

int t2h_version(void);
const char* t2h_last_error(void);

/* ------------------------------------------------------------------ GEMM ---
 * C[M,N] = epi( alpha * A'[M,K] * B[N,K]^T + bias[N] ) + residual[M,N]
 * on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 fma chains).
 *
 * A' is either the plain row-major matrix A (a_mode 0) or the implicit im2col
 * view of an NHWC image (a_mode 1: 3x3 taps, K = 9*Cin ordered [tap][cin]),
 * optionally preceded by a per-(image, channel) affine + activation prologue
 * (GroupNorm apply + swish fused into the operand load).
 *
 * Replaces: every nn.Linear of the sampler (models/archs/transformer_arch.py:
 * 41-49,70,84-89,271), 1x1 convs (vqgan_arch.py:590-595,627-634, post-quant
 * convs models/sample_model.py:230,240), 3x3 convs incl. nearest-x2 Upsample
 * and asym-padded stride-2 Downsample (vqgan_arch.py:526-551,573-580,954,997),
 * torch.bmm of AttnBlock (vqgan_arch.py:645-656), mmcv ConvModule conv+BN+ReLU
 * (models/archs/unet_arch.py:160-171, fcn_arch.py:297-305; BN folded by the
 * loader).
 */
typedef struct t2h_gemm_args {
  const float* A;         /* plain: [M,K] lda; conv: NHWC input [Bimg,Hin,Win,>=Cin], pixel stride lda */
  const float* B;         /* weights [N,K] row-major (ldb) or, b_trans=1, [K,N] (ldb) */
  float* C;               /* [M,N] ldc */
  const float* bias;      /* [N] or NULL */
  const float* residual;  /* [M,N] ldr or NULL; added after the activation, or before it if res_pre */
  const float* pro_scale; /* [n_img, pro_ld] or NULL: a' = act(a*scale + shift) */
  const float* pro_shift;
  int32_t M, N, K;
  int32_t lda, ldb, ldc, ldr;
  int32_t a_mode;         /* 0 plain, 1 conv3x3 */
  int32_t b_trans;        /* 0 / 1 (plain A, no prologue only) */
  int32_t pro_act;        /* 0 none, 1 swish x*sigmoid(x) */
  int32_t pro_rows;       /* plain mode: rows per image (H*W) for the prologue table */
  int32_t pro_ld;         /* channels per table row */
  int32_t epi_act;        /* 0 none, 1 GELU(erf), 2 ReLU */
  int32_t res_pre;        /* 1: residual is added BEFORE the activation (per-pixel bias map) */
  float alpha;
  int32_t Hin, Win, Cin;  /* conv: input image geometry (pre-upsample) */
  int32_t Hout, Wout;     /* conv: output geometry; M = n_img*Hout*Wout */
  int32_t stride, pad;    /* (1,1) same conv; (2,0) asym-padded downsample */
  int32_t ups;            /* 1: input is nearest-upsampled x2 on the fly */
  int32_t batch;          /* >=1: independent problems (blockIdx.z) */
  int64_t strideA, strideB, strideC; /* element strides between problems */
  /* t2h_conv_split_f32 only: if set, the epilogue also writes per-channel (sum, sum of squares) of the
   * FINAL output values over every 128-row tile, in fp64: gn_part_out[img][chunk][2][N], chunk = tile
   * index inside the image (rows per image / 128 chunks) -- the partial layout
   * t2h_groupnorm_finalize_f32 reduces, so the GroupNorm that follows needs no pass over the tensor */
  double* gn_part_out;
  /* t2h_gemm_f32, conv mode only (round 6): a caller-owned workspace lets the launch split K across workgroups -- the
   * deep levels of the index-prediction / parsing UNets (unet_arch.py:470-481,657-674: a handful of pixels per image,
   * K = 9 * 512 .. 9 * 1024) are 8-32 tiles that each stream megabytes of weights.  Slice s writes its partial tile to
   * splitk_ws[s][M][N]; a second kernel of the same call sums the slices in slice order (bit-reproducible) and applies
   * alpha / bias / residual / activation.  ksplit 0: the library chooses from (K, N, Hout * Wout) only -- never from
   * the number of images, so an image's values do not depend on its batch --, 1 = no split, > 1 forces that many
   * slices (needs splitk_ws_floats >= ksplit * M * N; any a_mode, batch 1).  splitk_ws NULL: never split. */
  float* splitk_ws;
  int64_t splitk_ws_floats;
  int32_t ksplit;
} t2h_gemm_args;

int t2h_gemm_f32(const t2h_gemm_args* args, void* stream);
/* the number of K slices t2h_gemm_f32 would use for `args` (1 = no split): callers size the workspace with it */
int t2h_gemm_ksplit(const t2h_gemm_args* args);
/* id of the tile configuration the dispatcher picks for `args` (profiling
 * labels): 0 64x64, 1 128x32, 2 128x64, 3 128x128, 4 128x64/K64, 5 128x128/K64,
 * 6 128x64/8 waves, 7 128x64/K64/8 waves, 8 128x128/K64/8 waves */
int t2h_gemm_tile_config(const t2h_gemm_args* args);
/* tuning hook: force a configuration id for every following GEMM of the CALLING THREAD whose
 * shape supports it (-1 = automatic); returns the previous setting */
int t2h_gemm_force_config(int cfg);

/* The stride-1 convolutions of the hierarchical VQGAN decode (3x3 'same', 3x3 after nearest-x2
 * upsample, 1x1) on the fp16 matrix cores at fp32-class accuracy.  Arguments as t2h_gemm_f32 except:
 * A points to the SPLIT-ROW activations [pixel][Cin/32][2][32] fp16 (t2h_gn_apply_split_f32 /
 * t2h_split_rows_f32; lda unused), B to the split-row weights [N][K/32][2][32]; no prologue tables
 * (GroupNorm + swish are applied by the pass that writes A); a_mode 1 geometry always (a 1x1
 * convolution is K == Cin, pad 0); pixels per image must be a multiple of 128; no b_trans / batch / alpha / stride 2 / GELU.
 * C, residual, bias stay fp32.  Replaces t2h_gemm_f32 at vqgan_arch.py:597-617,529-534,636-661,
 * 1000-1033,1136-1151 (opt-out: T2H_SPLIT_CONV=0). */
int t2h_conv_split_f32(const t2h_gemm_args* args, void* stream);
/* The same 3x3 convolutions (3x3 'same', 3x3 after nearest-x2 upsample) with GroupNorm apply + swish + the fp16
 * split folded into the operand staging: a workgroup owns a 16 x 16-pixel output tile x 128 output channels, stages
 * the 18 x 18-pixel halo of 32 input channels ONCE in LDS (fp32 read, activated, split) and serves the nine taps from
 * it -- no t2h_gn_apply_split_f32 pass (a full read + write of the normalised tensor), no per-tap re-read of the
 * input.  Arguments as t2h_gemm_f32's conv mode: A = fp32 NHWC input (pixel stride lda >= Cin), pro_scale /
 * pro_shift = the GroupNorm tables [n_img, pro_ld] with pro_act 1 (swish), or both NULL (plain convolution);
 * B = split-row weights [N][9*Cin/32][2][32] as for t2h_conv_split_f32; K = 9*Cin, stride 1, pad 1, ups 0 / 1;
 * Hout and Wout multiples of 16; bias / residual / gn_part_out as t2h_conv_split_f32 (a 128-pixel chunk of
 * gn_part_out is half a tile).  K is summed [channel group][tap] (t2h_conv_split_f32: [tap][channel group]): the
 * two agree to fp32 rounding, not bit for bit.  overflow_flag: the caller's sticky word (bit 0 is set when an
 * activated value leaves fp16's range).  Replaces t2h_gn_apply_split_f32 + t2h_conv_split_f32 at
 * vqgan_arch.py:597-617,529-534,1000-1033 at the decoders' large levels (the caller chooses by the image geometry,
 * never by the batch: ops.conv_halo_ok; opt-out: T2H_HALO_CONV=0).  Default kernel: weight tiles by LDS-DMA, Cin <= 512
 * with tables, an image and the weights < 2 GiB each. */
int t2h_conv_halo_f32(const t2h_gemm_args* args, int32_t* overflow_flag, void* stream);
int t2h_conv_halo_force_variant(int v); /* tuning / tests (thread-local): 1 = LDS-DMA weight tiles, two fragment sets, staged conversion (default); 0 = the first, register-staged kernel; returns the old value */
/* 3x3 'same' convolution with 1..4 output channels (the decoders' conv_out, vqgan_arch.py:997,1026-1033) on
 * the vector ALU, exact fp32: out[pixel][co] = bias[co] + sum over taps, channels of
 * act(x * scale[img] + shift[img]) * w[co][tap][c] (scale NULL = no prologue; act 1 = swish), zero padding of
 * the ACTIVATED tensor.  x rows [n_img*H*W, >= Cin] (stride ldx), w packed [Cout][9][Cin] like t2h_gemm_f32's
 * conv mode, out rows (stride ldo >= Cout).  Cin % 32 == 0. */
int t2h_conv3x3_small_f32(const float* x, int32_t ldx, const float* w, const float* bias, const float* scale,
                          const float* shift, int32_t tbl_ld, int32_t act, float* out, int32_t ldo, int32_t n_img,
                          int32_t H, int32_t W, int32_t Cin, int32_t Cout, void* stream);
int t2h_conv_split_force_tile(int rows); /* tuning / tests (thread-local): 128 or 256 output pixels per tile, 0 auto; returns the old value */
/* out_split[row] = split( act( x[row] * scale[img] + shift[img] ) ): GroupNorm apply (tables of
 * t2h_groupnorm_tables_f32; NULL = plain split) + swish (act 1) of fp32 NHWC rows in one pass
 * (vqgan_arch.py:510-517,599-600,609-610,637,1026-1027) */
int t2h_gn_apply_split_f32(const float* x, int32_t ldx, const float* scale, const float* shift, int32_t tbl_ld,
                           uint16_t* out_split, int64_t rows, int32_t rows_per_img, int32_t C, int32_t act,
                           int32_t* overflow_flag, void* stream);

/* ------------------------------------------------- split-precision GEMM -----
 * Same contraction on the fp16 matrix cores at fp32-class accuracy: every fp32
 * operand is carried as two fp16 planes (x = h + l / 2048, h = fp16(x),
 * l = fp16((x - h) * 2048); |x| < 65504) and a product is the three partial products
 * hh + (hl + lh) / 2048 (v_mfma_f32_32x32x16_f16, fp32 accumulate).  Operands are
 * "split rows": [rows][K/32][2][32] fp16 (128 B per row and 32-wide K tile), written
 * by t2h_split_rows_f32 / the C_split epilogue.
 * Replaces the same nn.Linear call sites as t2h_gemm_f32 (default path of the sampler).
 * Dispatch (automatic; t2h_gemm_split_force_config overrides): 256x128 tiles with a ping-pong LDS-DMA
 * loop when they give >= 192 tiles (128x192 on the same loop where that takes fewer rounds of the 256
 * CUs: q|k|v at M = 4096), 128x64 with an in-block K split when N = 512 at M = 4096, a
 * few-rows kernel (16x16x32 MFMA straight from global memory, K split over 8 waves) for M <= 64,
 * 128x64 / 128x128 otherwise.  Operands must span < 2 GiB each (32-bit byte offsets). */
typedef struct t2h_gemm_split_args {
  const uint16_t* A;      /* split rows [M][K/32][2][32] */
  const uint16_t* B;      /* split rows [N][K/32][2][32] (weights, repacked once) */
  float* C;               /* fp32 output [M,N] ldc, or NULL */
  uint16_t* C_split;      /* split-row output [M][N/32][2][32], or NULL */
  const float* bias;      /* [N] or NULL */
  const float* residual;  /* fp32 [M,N] ldr or NULL (added after the activation) */
  int32_t M, N, K;
  int32_t ldc, ldr;
  int32_t epi_act;        /* 0 none, 1 GELU(erf), 2 ReLU */
  /* optional "V transposed" routing for the q|k|v projection (transformer_arch.py:41-49):
   * output columns >= vt_col0 (the value heads, vt_hd columns per head) are NOT written to
   * C / C_split but, plus bias, as the two fp16 planes to Vt[B][H][2][vt_hd][vt_T] with row
   * index = b * vt_T + key; inside every group of 32 keys, key k sits at position
   * 16(k>>4) + 8((k>>2)&1) + 4((k>>3)&1) + (k&3), the order in which t2h_mha_split_f32's
   * P*V matrix instruction contracts them.  NULL = off. */
  uint16_t* Vt;
  int32_t vt_col0, vt_T, vt_hd;
  int32_t* overflow_flag; /* the caller's sticky overflow word (below); required with C_split / Vt */
  /* "x8" operands (fmt = 1; 0 = the two-fp16-plane rows above): A and B rows are [rows][K/32][hi16: 32 fp16 | hi8:
   * 32 e4m3 | lo8: 32 e4m3] (still 128 bytes per (row, K tile)), hi8 / lo8 = e4m3(h s), e4m3(l s) with a power of
   * two s per tensor; the hi*hi product runs on v_mfma_f32_32x32x16_f16, BOTH cross terms of a K tile in ONE
   * v_mfma_scale_f32_32x32x64_f8f6f4 (A = [hi8 | lo8], B = [lo8 | hi8] along its K = 64), and the cross-term
   * accumulator is merged with lo_mul = 2^-11 / (s_A s_B) (0 = 2^-11).  out_fmt = 1 writes C_split in the x8
   * format with scale out_scale (0 = 1); Vt planes and out_fmt = 0 outputs are unchanged.  Writers raise bit 1 of
   * the overflow word when |x| s >= 448 (t2h_split_rows_x8_f32, t2h_layernorm_x8_f32, t2h_mha_split_f32 y_fmt 1). */
  int32_t fmt, out_fmt;
  float lo_mul, out_scale;
  /* split over K across workgroups (0 / 1 = off; 2 or 4; x8 operands on the ping-pong tile configurations 8 / 10 / 11
   * only): slice s of K is computed by its own workgroups and written as a PARTIAL fp32 tile to C + s * M * ldc (C
   * holds ksplit * M rows; bias and residual go with slice 0); the caller sums the slices in a fixed order.  Built
   * for the proj / fc2 experiment of round 6 (profiles/r06_fc2_128x128_splitk.log); the dispatcher never picks it. */
  int32_t ksplit;
} t2h_gemm_split_args;

int t2h_gemm_split_f32(const t2h_gemm_split_args* args, void* stream);
/* Sticky overflow flag of every split-row producer (this GEMM's C_split / Vt epilogues, t2h_split_rows_f32,
 * t2h_layernorm_split_f32, t2h_gn_apply_split_f32, the attention outputs): ONE int32 word in device memory
 * that the CALLER allocates (zeroed) and passes as `overflow_flag`; a producer ORs 1 into it when a value about
 * to be written as split rows has |x| >= 65504 (the fp16 planes would hold inf / NaN).  The library never
 * allocates, reads or synchronises on it: a caller that drives several streams / models gives each its own
 * word.  t2h_split_overflow_async enqueues on `stream` a copy of *flag to *host_out (use pinned memory) and, if
 * reset != 0, a clear of the flag behind the copy; the value is valid once the caller has synchronised the
 * stream.  The host side clears the flag at the start of a sampling run / decode and checks it at the end. */
int t2h_split_overflow_async(int32_t* flag, int32_t* host_out, int32_t reset, void* stream);
/* measurement hook (bench.py): the NEXT t2h_gemm_split_f32 launch of the calling thread records its own
 * start / end into the two hipEvent_t (hipExtLaunchKernelGGL: the kernel's timestamps, what rocprofv3's
 * kernel trace reports), then the hook disarms.  NULL, NULL disarms. */
int t2h_gemm_split_time_next_launch(void* start_event, void* stop_event);
/* measurement hook (tools/gemm_phase_timing.py, bench.py): the NEXT t2h_gemm_split_f32 launch of the calling thread
 * stores phase stamps into dev_int64_buf[workgroups][16] (int64; the caller zeroes it and sizes it for the launch's
 * grid, 4096 workgroups always suffice at the sampler's shapes): per workgroup s_memrealtime (100 MHz) in slots 0..3 and
 * s_memtime (shader clock) in slots 8..11 at entry / prologue done / main loop done / epilogue stores issued -- phase
 * lengths and the clock the CU really ran at in each phase.  Disarms after the launch; NULL disarms. */
int t2h_gemm_split_probe_next_launch(void* dev_int64_buf);
/* id of the tile configuration the dispatcher picks for `args` (profiling labels; ids as t2h_gemm_split_force_config) */
int t2h_gemm_split_tile_config(const t2h_gemm_split_args* args);
int t2h_gemm_split_force_config(int cfg); /* tuning / tests: tile configuration 0..3, 5, 6, 8 / 10 / 11 (ping-pong LDS-DMA, 256x128 / 128x192 / 128x128), 9 (few-rows kernel), -1 auto;
                                              thread-local: it affects launches of the calling thread only */
/* fp32 [rows, C] (row stride ldx) -> split rows */
int t2h_split_rows_f32(const float* x, int32_t ldx, uint16_t* out, int64_t rows, int32_t C, int32_t* overflow_flag,
                       void* stream);
/* producers that emit split rows directly: LayerNorm (transformer_arch.py:93-95)
 * and the attention output (transformer_arch.py:65-67) */
int t2h_layernorm_split_f32(const float* x, const float* gamma, const float* beta, uint16_t* y_split,
                            int32_t rows, int32_t C, float eps, int32_t* overflow_flag, void* stream);
/* the same producers for the x8 format (t2h_gemm_split_args.fmt = 1): rows [rows][C/32][hi16 | hi8 | lo8], 8-bit
 * planes scaled by the power of two `scale` */
int t2h_split_rows_x8_f32(const float* x, int32_t ldx, uint16_t* out, int64_t rows, int32_t C, float scale,
                          int32_t* overflow_flag, void* stream);
int t2h_layernorm_x8_f32(const float* x, const float* gamma, const float* beta, uint16_t* y_x8, int32_t rows,
                         int32_t C, float eps, float scale, int32_t* overflow_flag, void* stream);
int t2h_mha_noncausal_split_f32(const float* qkv, uint16_t* y_split, int32_t B, int32_t T,
                                int32_t n_head, int32_t* overflow_flag, void* stream);
/* the same attention (transformer_arch.py:52-67, causal=False, head dim 64) with both
 * matrix products as three fp16 partial products: q and k are read as split rows from qk_split
 * ([B*T][ld_cols/32][2][32], q at columns [0, C), k at [C, 2C), C = 64 n_head -- the C_split
 * output of the q|k|v projection) and v from the Vt planes the same projection wrote
 * (t2h_gemm_split_args.Vt); output as fp32 rows y [B*T, C] and / or split rows y_split. */
int t2h_mha_split_f32(const uint16_t* qk_split, int32_t ld_cols, const uint16_t* vt, float* y,
                      uint16_t* y_split, int32_t B, int32_t T, int32_t n_head, int32_t* overflow_flag, void* stream);
/* ... with y_split written in the x8 format (8-bit planes scaled by the power of two `y_scale`) */
int t2h_mha_split_x8_f32(const uint16_t* qk_split, int32_t ld_cols, const uint16_t* vt, uint16_t* y_x8, float y_scale,
                         int32_t B, int32_t T, int32_t n_head, int32_t* overflow_flag, void* stream);
/* Two forms, chosen by the number of rounds of the 256 CUs each needs: 128-query workgroups whose two wave groups
 * take the two key halves and merge (what fills the chip at B = 8), or 256-query workgroups whose eight waves each
 * walk all keys (B >= 16: no merge, K / Vt tiles shared by eight waves).  Tuning / tests (thread-local): 1 = all
 * keys, 2 = key halves, 0 = automatic; returns the old value. */
int t2h_mha_split_force_form(int form);

/* ------------------------------------------------------ normalisation ------
 * LayerNorm over the last dim (eps 1e-5): transformer_arch.py:80-81,93-95,231,270 */
int t2h_layernorm_f32(const float* x, const float* gamma, const float* beta,
                      float* y, int32_t rows, int32_t C, float eps, void* stream);

/* GroupNorm(32 groups, eps 1e-6) statistics of an NHWC tensor turned into the
 * per-(image, channel) scale/shift tables consumed by the GEMM prologue
 * (vqgan_arch.py:515-517).  workspace: n_img*chunks*2*C doubles. */
int64_t t2h_groupnorm_workspace_bytes(int32_t n_img, int32_t HW, int32_t C);
int t2h_groupnorm_tables_f32(const float* x, int32_t ldx, const float* gamma,
                             const float* beta, float* scale, float* shift,
                             int32_t n_img, int32_t HW, int32_t C, int32_t groups,
                             float eps, void* workspace, void* stream);
/* the second half of it: tables from per-chunk partial sums part[n_img][chunks][2][C] (fp64), as
 * t2h_conv_split_f32's epilogue writes them (t2h_gemm_args.gn_part_out) */
int t2h_groupnorm_finalize_f32(const double* part, int32_t chunks, const float* gamma, const float* beta,
                               float* scale, float* shift, int32_t n_img, int32_t HW, int32_t C,
                               int32_t groups, float eps, void* stream);

/* AttnBlock's attention (vqgan_arch.py:645-656: bmm -> * C^-0.5 -> softmax -> bmm) flash-style: the
 * N x N score matrix is never written.  qkv rows [n_img * N, >= 3C] (row stride ld) hold q | k | v of one
 * head of width C (256 or 512); out rows [n_img * N, C] (stride ldo).  Exact fp32 matrix instructions;
 * N % 32 == 0.  Used by the decoders' AttnBlocks (N = 512 / 2048, 2048 / 8192 at 1024x512). */
int t2h_spatial_attention_f32(const float* qkv, int32_t ld, float* out, int32_t ldo, int32_t n_img, int32_t N,
                              int32_t C, float scale, void* stream);
/* in-place row softmax of [rows, n] (AttnBlock, vqgan_arch.py:647; the encoders' AttnBlocks keep the
 * materialised form: t2h_gemm_f32 batched -> this -> t2h_gemm_f32 batched) */
int t2h_softmax_rows_f32(float* x, int32_t rows, int32_t n, int32_t ld, void* stream);

/* ------------------------------------------------------ sampler ------------
 * x[B*T, C] = tok_emb[idx] + pos_emb[t] + segm_emb[segm] + texture_emb[tex]
 * (transformer_arch.py:251-266) */
int t2h_embed_sum4_f32(const int64_t* idx, const int64_t* segm, const int64_t* tex,
                       const float* tok_emb, const float* pos_emb,
                       const float* segm_emb, const float* tex_emb, float* x,
                       int32_t B, int32_t T, int32_t C, void* stream);

/* non-causal multi-head attention over a packed [B*T, 3*C] q|k|v buffer
 * (CausalSelfAttention.forward with causal=False, transformer_arch.py:37-67).
 * T % 128 == 0, head_dim == 64. */
int t2h_mha_noncausal_f32(const float* qkv, float* y, int32_t B, int32_t T,
                          int32_t n_head, void* stream);

/* one step of the absorbing-diffusion unmasking schedule
 * (models/sample_model.py:286-292,301-302): changes = rand < 1/t & ~unmasked;
 * unmasked |= changes; head_count[h] += #changed tokens of texture h.  With
 * changed_rows (or NULL): the changed token rows are also appended to that list (any
 * order) and counted in head_count[n_heads]. */
int t2h_unmask_step(const float* rand, int32_t t, uint8_t* unmasked, uint8_t* changes,
                    const int64_t* tex, int32_t* head_count, int32_t n, int32_t* changed_rows,
                    int32_t n_heads, void* stream);

/* texture-routed categorical sampling of the changed tokens of ONE head
 * (models/sample_model.py:304-317 + ln_f and head_list[h] of
 * transformer_arch.py:270-271): for rows with changes && tex==head:
 *   logits = W_head * LN_f(hidden[row]); x0 = argmax(softmax(logits/temp)/expo[row]);
 *   x_t[row] = x0 + 1024*head; out_idx[row] = x0.
 * expo is the reference's full [n, n_class] Exp(1) draw for this head. */
int t2h_sample_head(const float* hidden, const float* lnf_gamma, const float* lnf_beta,
                    const float* w_head, const float* expo, const uint8_t* changes,
                    const int64_t* tex, int32_t head, float temp, int64_t* x_t,
                    int64_t* out_idx, int32_t n, int32_t C, int32_t n_class,
                    void* stream);

/* the same for ALL heads in one launch over the compact list of changed rows: row r is
 * sampled with head h = tex[r], weights w_heads[h], noise expo[h] (the reference's per
 * ACTIVE head draw, in head order; NULL for heads that drew none), out_idx[h][r]. */
#define T2H_MAX_HEADS 32
typedef struct t2h_sample_heads_args {
  const float* hidden;      /* [n, C] transformer output before ln_f */
  const float* lnf_gamma;
  const float* lnf_beta;
  const float* w_heads;     /* [n_heads][n_class][C] */
  const float* expo[T2H_MAX_HEADS];
  const int32_t* rows;      /* changed rows (t2h_unmask_step) */
  const int64_t* tex;
  int64_t* x_t;
  int64_t* out_idx;         /* [n_heads][n] */
  float temp;
  int32_t n_rows, n, C, n_class, n_heads;
  float* logits_ws;         /* optional scratch [n_rows][n_class]: the head GEMV of a row is then spread
                               over 8 workgroups and the race runs in a second launch (same results) */
  /* philox_grid_threads != 0 (needs logits_ws): the Exp(1) noise of head h is NOT read from expo[h] but
   * computed in the kernel as element (row * n_class + j) of the tensor torch's
   * `torch.empty(n, n_class).exponential_()` would have drawn with generator seed philox_seed at
   * generator offset philox_offset[h] (grid_threads = 256 * the grid ATen launches for that numel);
   * the caller advances the generator by the draw's increment per ACTIVE head, in head order. */
  int32_t hidden_compact;   /* != 0: `hidden` holds only the listed rows, hidden[i] = row rows[i] (the
                               last layer's row-wise tail was evaluated for the changed rows only) */
  uint64_t philox_seed;
  uint64_t philox_offset[T2H_MAX_HEADS];
  uint32_t philox_grid_threads;
  /* per-LISTED-ROW noise (needs logits_ws), for row lists that mix sampling steps (the schedule of
   * t2h_unmask_schedule regrouped so that every sample advances through its own active steps):
   * row_philox_offset[i] = generator offset of the exponential_ draw that rows[i]'s head makes at
   * rows[i]'s step (replaces philox_offset[head]; needs philox_grid_threads), or
   * expo_rows[(expo_slot ? expo_slot[i] : i)][n_class] = that row of the explicit draw.  NULL = off. */
  const uint64_t* row_philox_offset;
  const float* expo_rows;
  const int32_t* expo_slot;
  const uint64_t* philox_seed_dev; /* if set, the seed is read from device memory (a captured launch sequence
                                      replayed for runs with different seeds); else philox_seed */
  const int32_t* rng_rows;         /* optional, with row_philox_offset: rng_rows[i] = the row of the reference's
                                      [n, n_class] exponential_ tensor whose elements rows[i] draws (a host that
                                      reorders the samples of a batch keeps every row's own noise); NULL = rows[i] */
} t2h_sample_heads_args;
/* max |x| as the bits of the fp32 maximum, atomicMax'ed into *out_bits (the caller zeroes it; uint order = float order
 * for non-negative values, so the result does not depend on the order of the updates): of an fp32 matrix, and of the
 * fp16 hi plane of split rows / x8 rows.  Load-time plumbing of the x8 format's per-tensor scales (weights' own maxima,
 * activation maxima of the checkpoint-only calibration evaluation) -- not on the sampling path. */
int t2h_absmax_f32(const float* x, int32_t ldx, int64_t rows, int32_t C, uint32_t* out_bits, void* stream);
int t2h_split_rows_absmax(const uint16_t* rows_split, int64_t rows, int32_t C, uint32_t* out_bits, void* stream);
/* dst[i] = src[rows[i]], rows of row_bytes (multiple of 16) bytes */
int t2h_gather_rows(const void* src, const int32_t* rows, void* dst, int32_t n_rows, int32_t row_bytes, void* stream);
/* element-by-element reproduction of `torch.empty(numel).exponential_()` on the device generator
 * (seed, offset as the generator holds them BEFORE the draw; grid_threads as above) */
int t2h_philox_exponential_f32(uint64_t seed, uint64_t offset, uint32_t grid_threads, float* out, int64_t numel,
                               void* stream);
/* the same for `torch.rand(numel)` (uniform_(0, 1): curand_uniform with the bounds reversed, 1 -> 0) */
int t2h_philox_uniform_f32(uint64_t seed, uint64_t offset, uint32_t grid_threads, float* out, int64_t numel,
                           void* stream);
/* The whole unmasking schedule of one sample_fn call (models/sample_model.py:279-292,301-306) in one
 * launch: the schedule depends on the `rand([B,T])` draws only, never on the transformer, and the
 * generator offset of every draw only on how many heads sampled at the earlier steps.  Reproduces
 * the `steps` rand draws of torch's device generator (seed, `offset` = its offset before the first
 * draw; rand_grid_threads / rand_inc = ATen's grid and offset increment for an n-element draw,
 * expo_inc = the increment of one [n, n_class] exponential_ draw) and writes
 *   step_of_row[n]        the step t (steps .. 1) at which token row i is unmasked,
 *   head_mask[steps + 1]  bit h of head_mask[t] set iff a token of texture h is unmasked at step t
 * (head_mask[0] = 0).  The generator offset before the rand of step t is then
 * offset + sum over s > t of (rand_inc + popcount(head_mask[s]) * expo_inc); the exponential_ draw
 * of the i-th active head (ascending) at step t starts rand_inc + i * expo_inc after it.
 * steps <= 4096, n_heads <= 32. */
int t2h_unmask_schedule(uint64_t seed, uint64_t offset, uint32_t rand_grid_threads, uint32_t rand_inc,
                        uint32_t expo_inc, const int64_t* tex, int32_t n, int32_t steps, int32_t n_heads,
                        int32_t* step_of_row, uint32_t* head_mask, void* stream);
/* Round cursor of a schedule laid out as padded tables [rounds][maxr] (a round's list padded with
 * copies of one of its own rows: sampling a row twice writes the same token twice): copies round
 * r = *round_ctr of rows_tbl (and of aux64_tbl = per-row generator offsets / aux32_tbl = per-row explicit
 * noise slots, either may be NULL) into cur_rows / cur_aux64 / cur_aux32 and sets *round_ctr = r + 1.
 * With it one round of engine.sample_tokens is a fixed launch sequence with fixed arguments -- a
 * hipGraph captured once and replayed per round. */
int t2h_schedule_advance(const int32_t* rows_tbl, const int64_t* aux64_tbl, const int32_t* aux32_tbl,
                         int32_t* round_ctr, int32_t* cur_rows, int64_t* cur_aux64, int32_t* cur_aux32,
                         int32_t maxr, void* stream);
int t2h_sample_heads(const t2h_sample_heads_args* args, void* stream);

/* Sampler training-time forward (models/transformer_model.py:212-274, forward only).
 * q_sample: mask[b,i] = u[b,i] < t[b] / num_timesteps; x_t = mask ? mask_id : x0.
 * masked_ce_heads: sum over the 18 heads of F.cross_entropy(logits_h, gt_h, ignore_index=-1,
 * reduction='none') -- only the head of a token's own texture has a target != -1 -- for the
 * masked tokens: ce_rows[row] (0 elsewhere) and ce_samples[b] = sum over the T rows of b.
 * gt_lists: [n_heads][B*T] int64 (-1 = not of that texture), as t2h_vq_argmin_tex_f32 writes. */
int t2h_q_sample(const int64_t* x0, const float* u, const int64_t* t, int32_t num_timesteps,
                 int64_t mask_id, int64_t* x_t, uint8_t* mask, int32_t B, int32_t T, void* stream);
int t2h_masked_ce_heads(const float* hidden, const float* lnf_gamma, const float* lnf_beta,
                        const float* w_heads, const int64_t* tex, const uint8_t* mask,
                        const int64_t* gt_lists, float* ce_rows, float* ce_samples, int32_t B, int32_t T,
                        int32_t C, int32_t n_class, int32_t n_heads, void* stream);

/* ------------------------------------------------------ quantizers ---------
 * VectorQuantizer.forward distance+argmin, vqgan_arch.py:88-92 (first min wins) */
int t2h_vq_l2_argmin_f32(const float* z, const float* codebook, int64_t* idx,
                         int32_t n, int32_t n_e, int32_t d, void* stream);

/* Encode side: texture-routed quantisation.  VectorQuantizerTexture.forward
 * (vqgan_arch.py:232-268; fold_h = fold_w = 0, z = [n, d] latent rows) and
 * VectorQuantizerSpatialTextureAware.forward (:392-443; z = NHWC map
 * [n / (fold_h fold_w), 2 fold_h, 2 fold_w, d / 4], row (b, i, j) = its 2x2 patch in
 * F.unfold layout [c, kh, kw]).  idx_lists[tex[row]][row] = argmin_j |z - books[tex[row]][j]|^2
 * (expanded form, first minimum wins), every other idx_lists[h][row] = -1.
 * books: [n_books][n_e][d]; d in {256, 1024}. */
int t2h_vq_argmin_tex_f32(const float* z, const float* books, const int64_t* tex, int64_t* idx_lists,
                          int32_t n, int32_t n_books, int32_t n_e, int32_t d, int32_t fold_h,
                          int32_t fold_w, void* stream);

/* VectorQuantizerTexture.get_codebook_entry, vqgan_arch.py:289-309:
 * out[row, :] = books[tex[row]][idx[tex[row]][row]]  (idx_lists: [18, n]) */
int t2h_codebook_gather_tex_f32(const int64_t* idx_lists, const int64_t* tex,
                                const float* books, float* out, int32_t n,
                                int32_t n_books, int32_t n_e, int32_t e_dim,
                                void* stream);

/* VectorQuantizerSpatialTextureAware.get_codebook_entry incl. F.fold(k=2,s=2),
 * vqgan_arch.py:463-486: NHWC out [B, 2h, 2w, C] from [C*4]-entries ([c,kh,kw]) */
int t2h_codebook_gather_fold_f32(const int64_t* idx_lists, const int64_t* tex,
                                 const float* books, float* out, int32_t B,
                                 int32_t h, int32_t w, int32_t n_books, int32_t n_e,
                                 int32_t C, void* stream);

/* texture-routed 1x1 head + argmax of bot_index_prediction
 * (fcn_arch.py:338-346, models/sample_model.py:200-207):
 * out_lists[h][row] = argmax_k(W[h][k,:]*feat[row, h*Cf:(h+1)*Cf] + b[h][k]) if tex[row]==h else -1 */
int t2h_routed_head_argmax(const float* feat, int32_t ldf, const float* w,
                           const float* b, const int64_t* tex, int64_t* out_lists,
                           int32_t n, int32_t n_heads, int32_t Cf, int32_t n_class,
                           void* stream);

/* ------------------------------------------------------ layout / misc ------ */
/* F.one_hot(segm).permute(0,3,1,2) (models/sample_model.py:332-335) as NHWC
 * with the channel dim zero-padded to Cpad */
int t2h_onehot_nhwc_f32(const float* segm, float* out, int64_t n_pix, int32_t n_cls,
                        int32_t Cpad, void* stream);
int t2h_nchw_to_nhwc_f32(const float* x, float* y, int32_t B, int32_t C, int32_t HW,
                         int32_t ldy, void* stream);
int t2h_nhwc_to_nchw_f32(const float* x, int32_t ldx, float* y, int32_t B, int32_t C,
                         int32_t HW, void* stream);
/* nn.MaxPool2d(2) on NHWC (unet_arch.py:432) */
int t2h_maxpool2_nhwc_f32(const float* x, int32_t ldx, float* y, int32_t B, int32_t H,
                          int32_t W, int32_t C, void* stream);
/* nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) on NHWC
 * (unet_arch.py:285-286) */
int t2h_bilinear_up2_nhwc_f32(const float* x, float* y, int32_t B, int32_t H, int32_t W,
                              int32_t C, void* stream);
/* argmax over the channel dim of NHWC rows -> int64 (sample_model.py:436) */
int t2h_argmax_rows_f32(const float* x, int32_t ld, int64_t* out, int64_t rows,
                        int32_t n, void* stream);
/* ((dec+1)/2).clamp(0,1) NHWC[.,3] -> NCHW f32 image and (optional) HWC uint8
 * mul(255).add(0.5).clamp(0,255) (models/sample_model.py:245-254) */
int t2h_image_epilogue(const float* dec, int32_t ldd, float* img_nchw, uint8_t* img_u8,
                       int32_t B, int32_t HW, void* stream);
/* ShapeAttrEmbedding.forward (models/archs/shape_attr_embedding_arch.py:23-35).
 * w0t: first-layer weights transposed and stacked [sum(cls_num), dim] with
 * cls_off[a] = first row of attribute a; b0 [n_attr,dim]; w1 [n_attr,dim,dim];
 * b1 [n_attr,dim]; f0 [out_dim, n_attr*dim]; f1 [out_dim,out_dim]. */
int t2h_shape_attr_embed_f32(const int64_t* attr, const int32_t* cls_off, const float* w0t,
                             const float* b0, const float* w1, const float* b1, const float* f0,
                             const float* fb0, const float* f1, const float* fb1, float* out,
                             int32_t B, int32_t n_attr, int32_t dim, int32_t out_dim, void* stream);
/* per-pixel bias map of the spatially constant attribute channels that ShapeUNet
 * concatenates in front of every encoder stage (unet_arch.py:660-667):
 * out[b,y,x,co] = sum of tapc[b,co,tap] over the 3x3 taps that fall inside the image */
int t2h_tap_bias_map_f32(const float* tapc, float* out, int32_t B, int32_t H, int32_t W,
                         int32_t Cout, void* stream);
/* generate_texture_map rule (models/sample_model.py:443-467) */
int t2h_texture_map(const int64_t* segm, const int64_t* upper, const int64_t* lower,
                    const int64_t* outer, float* mask, int32_t B, int32_t HW,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* T2H_HIP_H */
