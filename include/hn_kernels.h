/*
 * hn_kernels.h — C ABI of the MI355X (gfx950) HyperNeRF render hot path.
 *
 * The reference (songrise/HyperNeRF-torch) has no FFI/plugin boundary: its hot path is plain
 * PyTorch behind Python call signatures.  This header is the boundary a native
 * replacement needs: plain pointers + sizes, caller-allocated buffers, launch on the caller's
 * stream, int status (0 = ok, <0 = argument error, >0 = hipError_t).  No allocation, no host
 * synchronisation, no global mutable state inside — every entry point is re-entrant and
 * graph-capturable.  Random draws (t_rand, u, noise) are explicit input buffers.
 *
 * Each entry point cites the reference code it replaces (paths relative to the reference repo).
 *
 * The dense part of the path (modules.MLP and everything built from it) runs on a small
 * "MLP machine": per 32-point block a wavefront keeps the hidden activation in registers with
 * points on the lane axis and features on the register axis, so an MFMA accumulator tile is the
 * next layer's B operand without touching LDS; weights are pre-packed into MFMA A-fragment order
 * (hn_pack_units) and streamed through LDS by LDS-DMA.  The host describes a network as a short
 * program of ops (HN_OP_* / HN_BOP_*), so every constructor configuration of the reference's
 * modules maps onto the same three kernels (forward, backward-data, weight-gradient).
 */
#ifndef HN_KERNELS_H
#define HN_KERNELS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hnStream_t; /* hipStream_t */

#define HN_VERSION 340   /* 320: HnDwJob carries a second X slot; 321: HN_BOP_AUX w2 = tile word; 330: a level composited
                            from two parts through a merge permutation (HnCompositeArgs.perm, hn_sample_pdf_split); 331: weight-gradient
                            jobs flush to partial slabs + hn_mlp_wgrad_reduce (HnDwBatch.partials, HnDwJob.p_tile); 340: hn_build_config,
                            hn_mlp_wgrad_reduce_adam (the reduce launch applies the optimizer) */

/* numeric modes of the MLP machine */
#define HN_MODE_F32 0  /* v_mfma_f32_32x32x2_f32: exact fp32 products, parity mode (<=1e-4 vs oracle) */
#define HN_MODE_BF16 1 /* v_mfma_f32_32x32x16_bf16: bf16 operands, fp32 accumulate, throughput mode   */
/* OPT-IN, not the mode the bench line is quoted on: HN_MODE_BF16 in every product of the forward and backward-data
 * machines and in every output, but the training stash (layer inputs X, layer gradients dZ — what ONLY the weight
 * gradient reads) is kept in 8 bits: X as OCP e4m3, dZ as OCP e5m2 of 2^dz_scale_log2 * dZ, 1 KiB per 32x32 tile
 * instead of 2; the weight-gradient kernel multiplies them with v_mfma_f32_32x32x16_bf8_fp8 (fp32 accumulate) and
 * undoes the scale.  Halves the stash traffic that bounds a training step; costs rounding noise in dW only
 * (DESIGN.md section 8).  hn_mlp_wgrad* take the scale in their mode argument: HN_MODE_BF16_S8 | dz_scale_log2 << 8. */
#define HN_MODE_BF16_S8 2

#define HN_MAX_SRC 8 /* forward: feature sources 0-3 ; backward: 0-3 same, 4-7 gradient inputs */
#define HN_MAX_DST 4
#define HN_MAX_SLOTS 128
#define HN_OP_WORDS 8
#ifndef HN_CHUNK_UNITS
#define HN_CHUNK_UNITS 32 /* weight stream is consumed in chunks of 32 units of 1 KiB */
#endif
#define HN_DSRC_COMPS 32  /* per-point source-gradient accumulators in the backward machine */

/* ---- forward ops: word0 = opcode ---------------------------------------------------------
 * Machine state per wavefront (32 points): `cur` = current hidden activation (<= 256 features) as
 * MFMA B fragments, `accL` = the last accumulator tile (pre-activation) for the OUT ops.
 *
 * HN_OP_LAYER: one Linear (+bias, +activation).  Input features = [cur (32*K32) | nG groups of 64
 * GENERATED features (feature table)].  For every 32-row output tile t < NT, one at a time:
 *     acc = bias[32t..] + W[32t.., main] . cur + W[32t.., aux] . generated ;  nxt[t] = act(acc)
 * then cur <- nxt (unless NO_COMMIT).  Weight-stream order per tile: K32 main blocks, then 2*nG
 * aux blocks (a "block" = 32 out x 32 in).  Training: relu bit mask -> mask_slot, activation ->
 * stash_out (X of the next layer's dW), generated features -> stash_aux.  The stash is a workspace of the three
 * machine kernels, opaque to the host: bf16 mode stores the operand fragments as they are (the weight-gradient kernel
 * transposes on its LDS read), fp32 mode stores tiles transposed through the matrix core. */
#define HN_OP_LAYER 1    /* w1 = K32 | nG<<8 | NT<<16 | act<<24 | flags<<28 ; w2=bias_off w3=feat_off
                            w4=mask|-1 w5=stash_out|-1 w6=stash_aux|-1 — RESOLVED for the launch: the byte offset
                            of block 0 of that stash region in KiB (masks: in units of 256 B); a region holds, per
                            32-point block, NT tiles (out), 2*nG tiles (aux) or (NT+1)/2 mask words per lane     */
#define HN_ACT_NONE 0
#define HN_ACT_RELU 1
#define HN_LAYER_NO_COMMIT 1 /* flags bit0: leave cur untouched (head layers read by an OUT op)      */
#define HN_LAYER_DIRECT 2    /* flags bit1: the layer's feature groups hold HN_FEAT_ID_DIRECT entries       */
/* dst[w1][p*ld + w2 + i] = act(accL row i) (+ residual src[w5][.. + w6 + i]), i < w3 <= 4           */
#define HN_OP_OUT 4      /* w1=dst  w2=col  w3=n  w4=act(0 none,1 sigmoid)  w5=res_src|-1  w6=res_col
                            w7 = 1 + first staged component the n results are ALSO published to (0 = none): later
                            layers of the same program encode them as generated features (warp -> template) */
#define HN_OP_OUT_WIDE 5 /* w1=dst  w2=col  w3=n (<= 32*NT)  w4=NT   (cur features -> dst)            */

/* ---- backward ops ---------------------------------------------------------------------------
 * State: `cur` = dZ of the layer being differentiated, `cur2` = a second, 32-feature dZ (heads).
 * HN_BOP_LAYER: dH = W^T . dZ for every 32-feature tile t < NT of the layer INPUT, one at a time:
 *     acc = W[:, 32t..]^T . cur (32*K32 dZ features) + W2[:, 32t..]^T . cur2 (if K32b) ;
 *     nxt[t] = acc * relu'(mask) ; then cur <- nxt.  Stream order per tile: K32 blocks, K32b blocks. */
#define HN_BOP_LOAD 1      /* w1=src w2=col w3=n(<=4) w4=act'(1 sigmoid: y from src w5 col w6)
                              w7=stash|-1 ; w3 bit 8 set: destination cur2 instead of cur ;
                              w3 bit 9 set: ADD the source-gradient accumulators of slots 8*((w3>>10)&3) + i — the
                              gradient that later ops of this program (earlier in forward order) left for the
                              components this head published (a NULL src then contributes nothing)         */
#define HN_BOP_LOAD_WIDE 2 /* w1=src  w2=col  w3=n  w4=NT  w5=relu mask|-1  w7=dZ stash|-1 (resolved offsets)   */
#define HN_BOP_LAYER 3     /* w1 = K32 | K32b<<8 | NT<<16 ; w4=mask|-1 w5=dZ stash|-1 (resolved offsets)   */
/* gradient w.r.t. GENERATED input features: per 32-feature tile of nG*64 features,
 * tmp = W_aux^T . (cur | cur2), then the chain rule through the feature table into the per-point
 * source-gradient accumulators (LDS), written to `dsrc` at the end of the program.               */
#define HN_BOP_AUX 4       /* w1 = K32 | K32b<<8 | nG<<16 ; w3=feat_off ; w2 = tile word: bit tt = tile tt (of 2*nG) holds
                              a feature with a gradient — only those tiles are in the weight stream and computed —,
                              bit 8+tt = one of them is trigonometric (else d feature / dx = 1, no factor applied)  */

/* feature table entry (8 bytes).  value(p) = kind(freq * x), x = staged source component `ci` of point p.
 * The distinct (source, column) pairs a program reads are listed once in HnMlpArgs.comps; every workgroup
 * stages them per 32-point block into LDS, so evaluating a feature costs two LDS reads and no global load. */
typedef struct {
  int32_t packed; /* bits 0-7 ci (index into comps) | 12-15 kind | 16-23 grad slot + 1 (0 = no gradient) */
  float freq;
} HnFeat;
#define HN_MAX_COMPS 32 /* staged source components per program */
#define HN_FEAT_ID_DIRECT 5 /* identity feature read straight from global memory: src in bits 8-11, column in
                               bits 24-31 (stand-alone modules with > HN_MAX_COMPS raw input channels) */
#define HN_FEAT_ZERO 0
#define HN_FEAT_ID 1
#define HN_FEAT_SIN 2  /* sin(freq*x)                                                        */
#define HN_FEAT_COS 3  /* cos(freq*x)                                                        */
#define HN_FEAT_SINP 4 /* sin(freq*x + 0.5*3.1415926): the reference's cosine in model_utils.posenc:262 */

typedef struct {
  const float* ptr; /* NULL: the source is absent — its staged components are left to the program itself (HN_OP_OUT
                       publishes head outputs as components, fused level programs) / its gradient input is zero */
  int32_t ld;      /* row stride in floats */
  int32_t per_ray; /* 1: row index = point / samples_per_ray, 0: row index = point */
  const int64_t* gather_idx; /* optional (per_ray sources): row index = gather_idx[ray] — the GLO embedding lookup
                                (modules.GLOEmbed, hypernerf/modules.py:155-167) read straight from the table;
                                an index outside [0, gather_rows) stages NaN */
  int32_t gather_rows;
  int32_t pad;
} HnSrc;

typedef struct {
  float* ptr;
  int32_t ld;
  int32_t pad;
} HnDst;

typedef struct {
  uint64_t off; /* byte offset from the stash / mask base for block 0 */
  int32_t nt;   /* stash: 32-feature tiles per block ; mask: dwords per lane per block */
  int32_t pad;
} HnSlot;

/* One launch of the forward or backward machine over n_points points.
 * Replaces, per launch: modules.MLP.forward (hypernerf/modules.py:116-127) and the modules
 * composed from it — TranslationField.warp (hypernerf/warping.py:90-96), HyperSheetMLP.forward
 * (hypernerf/modules.py:331-337), NerfMLP.forward (hypernerf/modules.py:266-298), the encoders
 * model_utils.posenc_orig / posenc (hypernerf/model_utils.py:234-274) fused as generated input
 * features, legacy models/nerf.py:83-124 — and, for the backward machine, their autograd. */
typedef struct {
  int32_t mode;            /* HN_MODE_* */
  int32_t n_points;        /* P */
  int32_t samples_per_ray; /* S (ray(p) = p / S) */
  int32_t training;        /* 0: skip mask/stash writes */
  int32_t n_ops;
  int32_t n_chunks; /* weight stream length in chunks of HN_CHUNK_UNITS KiB */
  int32_t n_dsrc;   /* backward: number of source-gradient components written per point (<=16) */
  int32_t n_bias;   /* floats in `bias` (staged into LDS once per workgroup) */
  int32_t n_feat;   /* entries in `feat` (staged into LDS once per workgroup) */
  int32_t max_groups; /* most feature groups any one layer has (the forward machine is built for <=2 and for 3) */
  const int32_t* ops;    /* device, n_ops * HN_OP_WORDS */
  const void* wstream;   /* device, packed weight units (hn_pack_units) */
  const float* bias;     /* device, packed biases (fp32) */
  const HnFeat* feat;    /* device, feature table */
  void* stash;           /* device, activation / dZ stash (training) */
  uint32_t* masks;       /* device, relu bit masks (training) */
  float* dsrc;           /* backward: [P][n_dsrc] source gradients */
  HnSrc src[HN_MAX_SRC]; /* forward: feature sources ; backward: same + gradient inputs */
  HnDst dst[HN_MAX_DST];
  HnSlot slots[HN_MAX_SLOTS]; /* unused by the kernels since ABI 200 (the op words carry resolved offsets); kept for
                                 hosts that want to hand the layout along */
  uint64_t* prof;        /* diagnostic only (NULL in production): 8 shader-clock sums, see tools/ */
  const int32_t* comps;  /* device, n_comps entries: src << 16 | column */
  int32_t n_comps;
  int32_t embed_reg_mask; /* backward, 0 = off: bit i set = accumulator register i holds (in one lane half or both)
                             a source-gradient slot of the gathered per-ray source; those are summed over the 32
                             points of the block (all of ONE ray: samples_per_ray % 32 == 0 is required) and added to
                             embed_grad[gather_idx[ray]][embed_col[slot]] — GLOEmbed's backward, scatter included */
  float* embed_grad;      /* (gather_rows, embed_dim) fp32, accumulated with float atomics */
  const int64_t* embed_idx;
  int32_t embed_rows, embed_dim;
  int8_t embed_col[HN_DSRC_COMPS]; /* slot -> column of the table row, -1 = not an embedding component */
  int32_t n_trig_comps;   /* forward, bf16: staged components 0 .. n_trig_comps-1 also get x / 2pi staged as hi + lo
                             (every component a trigonometric feature reads must be among them; >= 1 if n_comps > 0) */
  int32_t wide_ops;       /* bit 0: the op program holds HN_OP_OUT_WIDE / HN_BOP_LOAD_WIDE ops; bit 1: its feature table
                             holds HN_FEAT_ID_DIRECT entries (layers flagged HN_LAYER_DIRECT).  Forward launches take
                             the kernel build that carries those paths if either bit is set, backward launches if bit 0
                             is; 0 selects the builds without them (what every render-level program runs: no scratch) */
  int32_t dz_scale_log2;  /* HN_MODE_BF16_S8, backward: the machine carries 2^dz_scale_log2 * dZ (exact: a power of two)
                             so that the e5m2 stash keeps small gradients; source / embedding gradients leave unscaled */
  int32_t trig_lo_planes; /* 1: the lo planes are staged (exactly reduced sine arguments); 0: hi only + one shared zero
                             plane (programs whose staging would not fit into LDS otherwise: hn_mlp_forward returns -6
                             when ring + tables + 8 waves x (n_comps + n_trig + (lo ? n_trig : 1)) x 128 B > 158 KiB) */
  uint64_t* timeline;     /* optional (ABI 330), device uint64[8], zero-initialised by the caller once: the launch times
                             itself — [4] += last workgroup's end - first workgroup's start in ticks of the 100 MHz
                             wall clock, [5] += 1, [0]/[1] start / end of the last run, [6]/[7] first start / last end
                             ever; [2], [3] are tickets the launch leaves at zero.  Lets a step that is replayed as one
                             HIP graph report per-kernel durations of the timed replays themselves.  NULL = off. */
  float* embed_partial;   /* backward, optional (ABI 331): with embed_reg_mask set, the per-block sums of the gathered
                             row's gradient are STORED to embed_partial[block][embed_dim] (columns without a gradient
                             slot are left untouched) instead of added to embed_grad by float atomics;
                             hn_mlp_wgrad_reduce (HnEmbedReduce) then sums them per table row in a fixed order */
} HnMlpArgs;

/* weight packing: one descriptor per 1-KiB unit of a stream */
typedef struct {
  int32_t w_id;       /* index into the pointer table ; -1 = all-zero unit */
  int32_t ld;         /* source row stride (in_features) */
  int32_t r0, c0;     /* source row / col of A-operand (row 0, k 0) */
  int32_t r_end;      /* valid source rows  [.., r_end)  */
  int32_t c_end;      /* valid source cols  [.., c_end)  */
  int32_t k0;         /* first k feature of this unit within the 32-block (bf16: 0/16 ; f32: 4*g steps) */
  int32_t transposed; /* 0: A[row][k] = W[r0+row][c0+k] ; 1: A[row][k] = W[r0+k][c0+row] */
} HnPackUnit;

typedef struct {
  int32_t w_id; /* bias vector id or -1 */
  int32_t n;    /* valid length */
  int32_t off;  /* destination offset (floats) */
  int32_t len;  /* padded length */
} HnPackBias;

/* weight-gradient job: one workgroup (8 waves) accumulates the dW tile grid n_nt x n_kt of one Linear input
 * segment over a block range; pad = gn | gk<<8 | bps<<16: gn x gk (<= 8 waves) is the wave grid, each wave
 * owns a ceil(n_nt/gn) x ceil(n_kt/gk) (<= 4x2) tile rectangle; bps = point blocks per LDS stage
 * (bps * (n_nt+n_kt) tiles <= 32 KiB).  n_nt, n_kt <= 8 tiles (bf16) / 4 tiles (fp32).
 * The k-tiles may come from TWO stash slots (a skip layer's running activation followed by its re-appended encoder
 * input, hypernerf/modules.py:122-124): tiles 0 .. n_kt1-1 from (x_off, x_nt, x_t0), the rest from (x2_off, x2_nt,
 * x2_t0) — one job, one read of the dZ tiles, instead of one job per segment.  n_kt1 == n_kt: one slot. */
typedef struct {
  uint64_t z_off, x_off;  /* stash byte offsets (block 0) of dZ and X slots */
  int32_t z_nt, x_nt;     /* tiles per block of the two slots */
  int32_t z_t0, x_t0;     /* first tile used */
  int32_t n_nt, n_kt;     /* n-tiles of dZ, k-tiles of X handled by this job */
  int32_t blk0, blk1;     /* block range [blk0, blk1) */
  int32_t w_off;          /* offset (floats) of the gradient matrix (row-major (out,in)) in the flat
                             gradient buffer, -1 = none */
  int32_t ld;
  int32_t r0, c0;         /* destination row / col of (tile 0, tile 0) */
  int32_t r_end, c_end;   /* valid bounds */
  int32_t b_off;          /* offset of the bias gradient or -1 (db[r] += sum_p dZ[p][r]) */
  int32_t pad;
  uint64_t x2_off;        /* second X slot: stash byte offset (block 0) */
  int32_t x2_nt, x2_t0;   /* its tiles per block, first tile used */
  int32_t n_kt1;          /* k-tiles taken from the first X slot (the remaining n_kt - n_kt1 from the second) */
  int32_t p_tile;         /* batched launches with HnDwBatch.partials (ABI 331): the job writes its n_nt x n_kt dW tiles —
                             raw accumulators, tile (i, j) at partials + (p_tile + i * n_kt + j) * 1024 floats, element
                             [register quad][lane][4]; bias sums in tile p_tile + n_nt * n_kt — with plain stores instead of adding them to the gradient by float
                             atomics; hn_mlp_wgrad_reduce sums the jobs' slabs and adds each element ONCE */
} HnDwJob;

int hn_version(void);
/* sizeof() of the ABI structs, in the order HnMlpArgs, HnPackUnit, HnPackBias, HnDwJob,
 * HnCompositeArgs, HnFeat, HnSlot, HnSrc — lets a foreign-language binding verify its mirror. */
int hn_abi_sizes(int32_t* out, int n);
/* The build-time tuning knobs of THIS library (ABI 340): out[0..] = ring depth and LDS-DMA pieces per wave and stage of
 * hn_wgrad_kernel, bias-by-MFMA build (0/1), weight-stream chunk in units, block-read build (0/1), asymmetric weight-stream
 * issue (0/1), waves per workgroup of the bf16 machines, cache policy of the stash stream.  Returns the number of
 * entries.  A binding derives its host-side mirrors (job stage cuts, chunk alignment, reduce tables) from these instead of
 * assuming the defaults. */
#define HN_BUILD_CONFIG_N 8
int hn_build_config(int32_t* out, int n);

/* Pack fp32 nn.Linear weights (row-major (out,in), reference layout hypernerf/modules.py:99-102)
 * into MFMA A-fragment order.  ptrs: device array of source pointers. */
int hn_pack_units(int mode, const HnPackUnit* units_dev, int n_units, const float* const* ptrs_dev,
                  void* wstream_dev, const HnPackBias* bias_dev, int n_bias, float* bias_out_dev,
                  hnStream_t stream);

/* The same for up to HN_MAX_PACK_JOBS programs in ONE launch (ABI 331): a training step re-packs the streams of three
 * programs after every optimizer step, and each of those launches costs more in dispatch than in work. */
#define HN_MAX_PACK_JOBS 8
typedef struct {
  const HnPackUnit* units;      /* device */
  const float* const* ptrs;     /* device: the program's pointer table */
  void* wstream;                /* device */
  const HnPackBias* bias;       /* device */
  float* bias_out;              /* device */
  int32_t n_units, n_bias;
} HnPackJob;
int hn_pack_units_multi(int mode, const HnPackJob* jobs_host, int n_jobs, hnStream_t stream);

int hn_mlp_forward(const HnMlpArgs* args, hnStream_t stream);
int hn_mlp_backward(const HnMlpArgs* args, hnStream_t stream);
/* Workspace query (host arithmetic only, no GPU needed): bytes of `stash` and of `masks` that the forward
 * (backward = 0) or backward (= 1) op program `ops_host` (HOST copy of HnMlpArgs.ops) touches for n_points
 * points in a training launch.  A stash is shared by a program's forward, backward and weight-gradient launches:
 * allocate the maximum of the two queries.  In the reference this is torch's autograd saving activations
 * (every nn.Linear / ReLU of hypernerf/modules.py:116-127). */
int hn_mlp_workspace_bytes(const int32_t* ops_host, int n_ops, int backward, int mode, int64_t n_points,
                           int64_t* stash_bytes, int64_t* mask_bytes);

/* dW/db for every Linear of a program: grads (fp32) are ACCUMULATED with float atomics.
 * `mode` of the three hn_mlp_wgrad* launches is a word: bits 0..7 = HN_MODE_*; HN_MODE_BF16_S8: bits 8.. = dz_scale_log2;
 * HN_MODE_BF16 / HN_MODE_F32: bits 8..15 = the LDS stage, in KiB, the host cut the jobs' blocks-per-stage (HnDwJob.pad
 * bits 16..23) for — 0 = not stated.  The kernel's ring is a compile-time constant of the library (2 x 64 KiB); a stated
 * stage that does not fit it is refused with -8 instead of silently loading part of every stage. */
int hn_mlp_wgrad(int mode, const HnDwJob* jobs_dev, int n_jobs, const void* stash_dev,
                 float* grad_base_dev, hnStream_t stream);

/* The same for up to HN_MAX_WGRAD_BATCH programs in ONE launch (a training step runs 6 programs; launched one by
 * one, the small ones cannot fill the chip and every launch pays its own ramp and tail).  Workgroup g works on job
 * g - first(batch) of the batch that holds it — or, with `order_dev` (device, one int32 per job of all batches:
 * batch << 24 | job), job order_dev[g]: the host's global heaviest-first order, which list-schedules 5 % tighter than
 * per-batch order.  `batches` is a HOST array, copied into the kernel arguments. */
#define HN_MAX_WGRAD_BATCH 8
typedef struct {
  const HnDwJob* jobs; /* device */
  const void* stash;   /* device: the stash the jobs' z_off / x_off refer to */
  float* grads;        /* device: base the jobs' w_off / b_off refer to */
  int32_t n_jobs;
  int32_t pad;
  float* partials;     /* device or NULL: workspace of the jobs' dW slabs (HnDwJob.p_tile), 4 KiB per tile; NULL = the
                          jobs add their tiles to `grads` themselves (float atomics) */
} HnDwBatch;
int hn_mlp_wgrad_batched(int mode, const HnDwBatch* batches_host, int n_batches, const int32_t* order_dev,
                         hnStream_t stream);
/* The same with a kernel timeline (HnMlpArgs.timeline; NULL = off). */
int hn_mlp_wgrad_batched_t(int mode, const HnDwBatch* batches_host, int n_batches, const int32_t* order_dev,
                           uint64_t* timeline_dev, hnStream_t stream);


/* Second half of a batched launch whose batches carry `partials` (ABI 331).  A job ends with the flush of its dW
 * rectangle; by float atomics one CU retires ~5 GB/s of them (one 256-B wave-instruction per ~50 ns), 100 us of a
 * 650-us launch at BASELINE config 2 during which the CU streams nothing.  With partials the jobs store their raw
 * accumulator tiles (plain 256-B stores) and this launch — one workgroup per DESTINATION tile (32 x 32 elements of
 * one gradient matrix) — sums the tile over every job that produced a slab for it, in the order of `list`, and adds
 * the sum to the gradient once: 1/12 of the atomics, and a summation order that does not depend on which job finished
 * first.  tiles_dev[t]: destination; its slabs are list_dev[first .. first + count): batch << 28 | slab tile index
 * (tile k of batch b starts at batches[b].partials + k * 1024 floats).  A record with ld == 0 is a BIAS record: w_off =
 * offset of the bias gradient, row0 = its first row, col0 = dZ tiles of the rectangle; a job with b_off >= 0 owns one
 * more slab tile behind its n_nt * n_kt dW tiles, 32 floats per dZ tile (HN_MODE_BF16_S8 keeps its bias atomics). */
typedef struct {
  int32_t batch;         /* whose `grads` receives the tile */
  int32_t w_off, ld;     /* gradient matrix (row-major (out, in)) in that buffer */
  int32_t row0, col0;    /* destination of accumulator element (row 0, col 0) of the tile */
  int32_t r_end, c_end;  /* valid bounds */
  int32_t first, count;  /* slice of list_dev */
} HnDwReduceTile;
/* Optional third kind of work of the same launch: the gradient of ONE gathered GLO table (modules.GLOEmbed's backward,
 * hypernerf/modules.py:155-167) from the per-block partial rows the backward machines of up to HN_MAX_WGRAD_BATCH
 * programs stored (HnMlpArgs.embed_partial): one workgroup per table row sums, program by program and block by block,
 * the rows of the blocks whose ray carries that index and adds the total to `grad` — a single writer per element, a
 * fixed order: with the slabs above the whole gradient of a step is bit-reproducible. */
typedef struct {
  float* grad;          /* (rows, dim) gradient of the table */
  int32_t rows, dim;
  uint32_t col_mask;    /* columns that carry a gradient (bit c = column c); dim <= 32 */
  int32_t n_src;
  const float* partial[HN_MAX_WGRAD_BATCH];   /* [n_blocks][dim] */
  const int64_t* idx[HN_MAX_WGRAD_BATCH];     /* ray -> table row */
  int32_t n_blocks[HN_MAX_WGRAD_BATCH];
  int32_t samples_per_ray[HN_MAX_WGRAD_BATCH];
} HnEmbedReduce;
int hn_mlp_wgrad_reduce(int mode, const HnDwReduceTile* tiles_dev, int n_tiles, const uint32_t* list_dev,
                        const HnDwBatch* batches_host, int n_batches, const HnEmbedReduce* embed_host,
                        hnStream_t stream);

/* The same launch, ALSO applying the optimizer (ABI 340; reference: train.py:116-131 configure_optimizers ->
 * utils.get_optimizer's torch.optim.Adam, utils/__init__.py:23-41, stepped right after loss.backward()): a workgroup
 * that has summed its destination tile updates the parameters behind it in registers (hn_adam_step's arithmetic,
 * operation for operation) and leaves the gradient buffer zeroed (zero_grad) or completed.  Single-GPU steps only: with an
 * all-reduce between backward and the optimizer the two-launch form stays.  `rest_dev`: the elements of the arena that
 * no tile / bias record / table row of this launch covers, as (start, len) ranges — each is handled by one more
 * workgroup, so that EVERY element of [0, n) is updated exactly once (the host builds and proves the partition).  All
 * `batches[i].grads` must be `grads`, the table gradient must lie inside [grads, grads + n): -9 otherwise. */
typedef struct {
  int64_t start;   /* first element (floats from the arena base) */
  int32_t len, pad;
} HnAdamRange;
typedef struct {
  float* params; float* grads; float* exp_avg; float* exp_avg_sq;   /* flat arena buffers of n floats, same layout */
  int64_t n;
  const float* hyper;   /* device: [lr, beta1, beta2, eps, weight_decay, grad_scale, -, -], as hn_adam_step */
  float* step;          /* device: [updates done, ticket], as hn_adam_step */
  const HnAdamRange* rest;   /* device */
  int32_t n_rest, zero_grad;
} HnAdamFuse;
int hn_mlp_wgrad_reduce_adam(int mode, const HnDwReduceTile* tiles_dev, int n_tiles, const uint32_t* list_dev,
                             const HnDwBatch* batches_host, int n_batches, const HnEmbedReduce* embed_host,
                             const HnAdamFuse* adam_host, hnStream_t stream);

/* ---- per-ray kernels --------------------------------------------------------------------- */

/* Stratified coarse samples + points.  model_utils.sample_along_rays (hypernerf/model_utils.py:6-41),
 * legacy models/rendering.py:189-207.  lower/upper: (n) per-bin bounds or (B,n) when per_ray_bounds;
 * t_rand (B,n) or NULL (z = lower).  Unfused fp32 ops in the reference's order (bit-exact z). */
int hn_sample_along_rays(const float* origins, const float* dirs, int ray_ld, const float* lower,
                         const float* upper, int per_ray_bounds, const float* t_rand, float scale,
                         int n_rays, int n, float* z_out, float* pts_out, hnStream_t stream);

/* Legacy nerf_pl sampling with per-ray near/far taken from columns 6,7 of the (B,>=8) ray rows
 * (models/rendering.py:189-207): z = near*(1-t)+far*t (or the disparity form), optional perturbation
 * z = lower + (upper-lower)*(scale*t_rand), pts = o + d*z.  t_vals / one_minus_t: (n) from the host. */
int hn_sample_legacy(const float* rays, int ray_ld, const float* t_vals, const float* one_minus_t,
                     int use_disp, const float* t_rand, float scale, int n_rays, int n, float* z_out,
                     float* pts_out, hnStream_t stream);

/* Stand-alone positional encoders: model_utils.posenc_orig (hypernerf/model_utils.py:234-246),
 * models/nerf.py:4-38 Embedding, and model_utils.posenc (:255-274, jax_cos=1: cos taken as
 * sin(x + 0.5*3.1415926)).  out = [x if identity][sin(f_k x), cos(f_k x)]_k, blocks of c channels.
 * Forward: out != NULL, g_out == NULL.  Backward: g_out != NULL, g_x receives d/dx. */
int hn_posenc(const float* x, int64_t n, int c, const float* freqs, int n_freqs, int identity, int jax_cos,
              float* out, const float* g_out, float* g_x, hnStream_t stream);

/* Softplus/relu density + alpha compositing.  model_utils.volumetric_rendering
 * (hypernerf/model_utils.py:43-107, 319-362) with NerfModel.query_template's noise+softplus
 * (hypernerf/models.py:485-491); legacy variant models/rendering.py:144-170.
 * variant 0 = hypernerf (softplus, last dist 1e7|1e-7, eps 1e-5 in cumprod, acc drops last sample)
 * variant 1 = legacy    (relu(sigma+noise), last dist 1e10, eps 1e-10, opacity = full sum)
 * variant 2 = hypernerf constants, `raw` is an already activated density (stand-alone
 *             model_utils.volumetric_rendering) */
typedef struct {
  int32_t variant, n_rays, n_samples, white_bg, sample_at_infinity, warped_ld;
  int32_t has_dust;     /* filter_sigma (models.py:35-63): drop densities below dust_threshold */
  float dust_threshold;
  const float* rgb;    /* (B,S,3) */
  const float* raw;    /* (B,S) raw density */
  const float* noise;  /* (B,S) standard-normal draws (or NULL); the kernel adds noise_scale * noise to the raw density
                          (model_utils.noise_regularize, model_utils.py:300-317: randn * noise_std) */
  const float* z;      /* (B,S) */
  const float* dirs;   /* (B,3) row stride ray_ld */
  int64_t ray_ld;
  const float* warped; /* (B,S,warped_ld) or NULL */
  float* out_rgb;      /* (B,3) */
  float* out_depth;    /* (B) */
  float* out_acc;      /* (B) */
  float* out_weights;  /* (B,S) */
  float* out_med_depth;  /* (B) or NULL */
  float* out_med_points; /* (B) or NULL */
  /* backward only */
  const float* g_rgb;     /* (B,3) or NULL */
  const float* g_depth;   /* (B) or NULL */
  const float* g_acc;     /* (B) or NULL */
  const float* g_weights; /* (B,S) or NULL */
  float* d_rgb;           /* (B,S,3) */
  float* d_raw;           /* (B,S) */
  const float* keep;      /* (B,S) 0/1 density mask (filter_sigma's bounding box) or NULL; forward and backward */
  float noise_scale;      /* noise_std; 1 for draws that arrive already scaled */
  int32_t split;          /* with `perm`: samples per ray held by part 0 (the rest, n_samples - split, by part 1) */
  /* A level evaluated in TWO parts (ABI 330; NULL perm = one part, everything above as before).  The fine level of
   * NerfModel.forward re-evaluates the coarse level's samples (hypernerf/models.py:752-768, model_utils.py:206-232:
   * sort(cat(z_coarse, z_new))); the warp field / hyper sheet results of those points already exist, so the host
   * evaluates the OLD samples (part 0: rgb, raw, warped — (B, split, .) in the coarse level's order) and the NEW
   * samples (part 1: rgb1, raw1, warped1 — (B, n_samples - split, .) in draw order) separately and composites them
   * through the merge permutation of hn_sample_pdf_split: sorted sample s of ray b is entry k = perm[b][s] of
   * cat(part 0, part 1) of that ray.  z, noise, keep, weights and g_weights stay in sorted order.  The backward writes
   * d_rgb / d_raw (part 0) and d_rgb1 / d_raw1 (part 1) in the parts' own order. */
  const int32_t* perm;    /* (B,S) int32 or NULL */
  const float* rgb1;      /* (B,S-split,3) */
  const float* raw1;      /* (B,S-split) */
  const float* warped1;   /* (B,S-split,warped_ld) or NULL */
  float* d_rgb1;          /* backward */
  float* d_raw1;          /* backward */
  float* out_warped;      /* forward, optional: (B,S,warped_ld) = the parts' warped rows in sorted order (the level's
                             `warped_points`, hypernerf/models.py:654-669) */
} HnCompositeArgs;
int hn_composite_forward(const HnCompositeArgs* a, hnStream_t stream);
int hn_composite_backward(const HnCompositeArgs* a, hnStream_t stream);

/* Median depth from given compositing weights: model_utils.compute_opaqueness_mask / compute_depth_index /
 * compute_depth_map (hypernerf/model_utils.py:319-362).  weights, z: (B, S) row-major.  Outputs (each may be NULL):
 * index (B) int64 = first sample whose inclusive weight sum reaches `threshold` (0 if none does: argmax of an
 * all-zero mask), depth (B) = z at that sample (0 if none), mask (B, S) = 1 at that sample, 0 elsewhere. */
int hn_depth_index(const float* weights, const float* z, int n_rays, int n_samples, float threshold,
                   int64_t* out_index, float* out_depth, float* out_mask, hnStream_t stream);

/* Inverse-CDF hierarchical sampling + merge-sort + points.  model_utils.piecewise_constant_pdf /
 * sample_pdf (hypernerf/model_utils.py:160-232) == legacy models/rendering.py:14-55,225-233.
 * weights: first used column of a (B, *) array with row stride w_ld, n_bins columns used;
 * bins: (B, n_bins+1) bin edges, or NULL = midpoints of z (then n_bins == n_coarse-2, and the caller
 * passes weights+1, i.e. coarse weights[:, 1:-1], as hypernerf/models.py:752-755 does);
 * z: (B, n_coarse) sorted coarse depths merged with the new samples, or NULL = no merge;
 * u: (B, n_fine).  Outputs (each may be NULL): z_all (B, n_coarse+n_fine) sorted, pts (.., 3),
 * inds (B, n_fine) int64 = searchsorted(cdf, u, right=True), z_samples (B, n_fine) unsorted.
 * The pdf normaliser is an fp64 sequential sum rounded once to fp32 and the cdf an fp64 sequential
 * prefix sum rounded per entry (== CPU torch.cumsum), so indices are bit-reproducible. */
int hn_sample_pdf(const float* weights, int w_ld, const float* bins, int n_bins, const float* z,
                  int n_coarse, const float* u, const float* origins, const float* dirs, int ray_ld,
                  int n_rays, int n_fine, float* z_all, float* pts, int64_t* inds, float* z_samples,
                  hnStream_t stream);
/* The same (ABI 330), for a fine level that evaluates only its NEW samples through the warp field (HnCompositeArgs.perm):
 * two more outputs, each may be NULL — perm (B, n_coarse+n_fine) int32: sorted position s holds entry perm[b][s] of
 * cat(z[b], z_samples[b]) (k < n_coarse: coarse sample k; else new sample k - n_coarse; equal depths keep that
 * order), and pts_new (B, n_fine, 3) = o + z_samples * d in draw order.  z_all / pts / inds / z_samples are bit-for-bit
 * those of hn_sample_pdf (the sort carries the index as a payload; keys are compared first).  perm needs z and z_all. */
int hn_sample_pdf_split(const float* weights, int w_ld, const float* bins, int n_bins, const float* z,
                        int n_coarse, const float* u, const float* origins, const float* dirs, int ray_ld,
                        int n_rays, int n_fine, float* z_all, float* pts, int64_t* inds, float* z_samples,
                        int32_t* perm, float* pts_new, hnStream_t stream);

/* A level's compositing (hn_composite_forward) AND the inverse-CDF sampling of the next level from its weights
 * (hn_sample_pdf_split's fused form: bins = midpoints of a->z, weights = columns 1 .. S-2 of the level's weights) as ONE
 * launch (ABI 340; reference call order models.py:744-768): the wave that composites a ray draws its fine samples, the
 * weights reach the sampler through LDS.  Outputs and arithmetic of both entry points, bit for bit; a->perm must be NULL. */
int hn_composite_sample_pdf(const HnCompositeArgs* a, const float* u, const float* origins, const float* dirs, int ray_ld,
                            int n_fine, float* z_all, float* pts, int64_t* inds, float* z_samples, int32_t* perm,
                            float* pts_new, hnStream_t stream);

/* GLO embedding lookup (modules.GLOEmbed, hypernerf/modules.py:155-167) and its gradient:
 * d_table[idx[b]] += sum_s d_embed[b, s, col0 : col0+dim]. */
int hn_embed_gather(const float* table, const int64_t* idx, int n_rays, int dim, int n_rows,
                    float* out, hnStream_t stream);
int hn_embed_backward(const float* d_points, int ld, int col0, const int64_t* idx, int n_rays,
                      int n_samples, int dim, int n_rows, float* d_table, hnStream_t stream);

/* Random draws of a render step in one launch (replaces the reference's torch.rand / torch.randn calls:
 * model_utils.py:31 t_rand, :226 u, :300-317 density noise).  Counter-based Philox4x32-10; state_dev = uint64[3]
 * {seed, offset, ticket (must start 0)} in device memory: the kernel advances `offset` itself, so a launch captured
 * in a HIP graph draws new numbers on every replay.  kind 0 = U[0,1) (24-bit, as torch.rand), 1 = N(0,1). */
#define HN_MAX_DRAWS 8
typedef struct HnDraw {
  float* ptr;      /* device buffer */
  int64_t n;       /* floats to fill */
  int32_t kind;
  int32_t pad;
} HnDraw;
int hn_random_fill(const HnDraw* draws_host, int n_draws, uint64_t* state_dev, hnStream_t stream);

/* The head of a render step as ONE launch (ABI 340): the weight streams of the step's programs (hn_pack_units_multi's
 * jobs), all its random draws (hn_random_fill's table and device state), the coarse samples placed from the t_rand draw
 * (hn_sample_along_rays: reference hypernerf/model_utils.py:6-41, same three separately rounded operations) and the int64
 * image ids of the (B, 9) ray rows (prepare_ray_dict's `rays[:, 8].type(torch.long)`, model_utils.py:365-404).  Any part
 * may be empty (n_pack = 0, n_draws = 0, p = NULL / t_rand_draw = -1 / n_ids = 0). */
typedef struct {
  int32_t t_rand_draw;     /* index into `draws` of the uniform (n_rays x n) buffer that is t_rand, or -1: no sampling */
  int32_t n_rays, n, ray_ld, per_ray_bounds;
  float scale;
  const float* origins; const float* dirs;   /* (n_rays, >= 3) rows, stride ray_ld */
  const float* lower; const float* upper;    /* (n) or (n_rays, n) bin bounds */
  float* z_out; float* pts_out;              /* (n_rays, n), (n_rays, n, 3) | NULL */
  const float* ids_src; int64_t* ids_dst;    /* ids_dst[i] = (int64) ids_src[i * ids_ld], i < n_ids */
  int32_t ids_ld, n_ids;
} HnPrologue;
int hn_render_prologue(int mode, const HnPackJob* pack_jobs_host, int n_pack, const HnDraw* draws_host, int n_draws,
                       uint64_t* state_dev, const HnPrologue* p_host, hnStream_t stream);


/* The reference's loss head (losses.py:4-14): loss = mean((coarse - gt)^2) [+ mean((fine - gt)^2)] over (B,3) pixels,
 * one launch forward (one workgroup; the result is written, not accumulated) and one launch backward:
 * d_coarse = d_fine-style 2 (pred - gt) / n * g_loss[0] (g_loss on the device; NULL = 1).  fine / d_fine may be NULL. */
int hn_mse_loss_forward(const float* coarse, const float* fine, const float* gt, int64_t n, float* loss_out,
                        hnStream_t stream);
int hn_mse_loss_backward(const float* coarse, const float* fine, const float* gt, int64_t n, const float* g_loss,
                         float* d_coarse, float* d_fine, hnStream_t stream);
/* Forward that also writes the gradients hn_mse_loss_backward would write for a root gradient of exactly 1 (bit for bit):
 * a training step whose loss is the root of the backward pass needs no second launch. */
int hn_mse_loss_forward_grad(const float* coarse, const float* fine, const float* gt, int64_t n, float* loss_out,
                             float* d_coarse, float* d_fine, hnStream_t stream);

/* torch.optim.Adam (the reference's default optimizer, utils/__init__.py get_optimizer) over ONE flat fp32 buffer
 * (ParamArena): p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps), m,v updated first, L2 weight decay added to the
 * gradient.  `hyper_dev`: 8 floats ON THE DEVICE, [lr, beta1, beta2, eps, weight_decay, grad_scale, 0, 0] — read by the
 * kernel, so a launch captured in a HIP graph follows a learning-rate schedule (utils/__init__.py:43-46) without
 * re-capture; grad_scale multiplies the gradient first (1/world after a SUM all-reduce).  `step_dev`: TWO 4-byte words
 * on the device, [0] = number of updates done so far as a float (advanced by the launch itself: the last block to
 * finish stores it), [1] = a ticket counter the launch leaves at zero.
 * zero_grad != 0 clears `grads` in the same pass.  Buffers 16-byte aligned. */
int hn_adam_step(float* params_dev, float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev, long long n,
                 const float* hyper_dev, float* step_dev, int zero_grad, hnStream_t stream);

/* All rays of one H x W image on the device: get_ray_directions + get_rays (+ get_ndc_rays when `ndc`)
 * (datasets/ray_utils.py:5-93) and the ray-row layout of datasets/llff.py:244-264:
 * rays[(j*W + i)] = [origin(3), direction(3), near, far(, image_id)], row_floats = 8 or 9.
 * c2w: (3,4) row-major fp32 on the device.  `ndc_near` is get_ndc_rays' near plane (the reference passes 1.0). */
int hn_generate_rays(int H, int W, float focal, const float* c2w_dev, int ndc, float ndc_near, float near,
                     float far, float image_id, int row_floats, float* rays_dev, hnStream_t stream);

/* SE(3) exponential-map warp of warping.SE3Field.warp (hypernerf/warping.py:226-238 with rigid_body.exp_se3,
 * rigid_body.py:55-83, applied per point): theta = |w|, a = w/theta, b = v/theta,
 * y = p + sin(theta) a x p + (1-cos(theta)) a x (a x p) + theta b + (1-cos(theta)) a x b + (theta-sin(theta)) a x (a x b).
 * w, v, points, out: (n_points, 3) fp32 row-major.  The backward writes the gradients w.r.t. w, v and points
 * (each may be NULL). */
int hn_se3_apply_forward(const float* w, const float* v, const float* points, int n_points, float* out,
                         hnStream_t stream);
int hn_se3_apply_backward(const float* w, const float* v, const float* points, const float* g_out, int n_points,
                          float* d_w, float* d_v, float* d_points, hnStream_t stream);

/* The same warp with a row stride per operand (w and v = the two halves of the field's (P, 6) head output, g_out =
 * columns of the template's source-gradient tensor: no slicing copies on either side), and — forward — an optional
 * second output rows_out (P, 3 + H), row stride rows_ld = [y | table[idx[p / samples_per_ray]]]: the `warped_points`
 * tensor of an axis-aligned-plane level (hypernerf/models.py:533-534, 578-581) without index_select + cat; an index
 * outside [0, n_rows) writes NaN, as the machine's own gather does.  out (P, 3) contiguous may be NULL if rows_out is
 * given; d_w / d_v / d_points may each be NULL. */
int hn_se3_warp_forward(const float* w, int w_ld, const float* v, int v_ld, const float* points, int p_ld,
                        int n_points, float* out, float* rows_out, int rows_ld, const float* table,
                        const int64_t* idx, int H, int n_rows, int samples_per_ray, hnStream_t stream);
int hn_se3_warp_backward(const float* w, int w_ld, const float* v, int v_ld, const float* points, int p_ld,
                         const float* g_out, int g_ld, int n_points, float* d_w, int dw_ld, float* d_v, int dv_ld,
                         float* d_points, hnStream_t stream);

/* Box calibration for a benchmark line (no reference counterpart: measurement infrastructure).  Both kernels time
 * themselves into t_dev (device uint64[16], zeroed by the caller; layout of HnMlpArgs.timeline in [0..7]).
 * hn_calib_mfma: every wave of 512 workgroups x 8 waves runs `iters` x 8 independent v_mfma_f32_32x32x16_bf16 from
 * registers — FLOPs = 512 * 8 * iters * 8 * 32768; t[8] / t[9] = shader-clock (s_memtime) / 100 MHz wall-clock ticks of
 * one wave's loop, i.e. the sustained shader clock = 1e8 * t[8] / t[9] Hz.
 * hn_calib_stream: 256 workgroups stream n_bytes (a multiple of 64 KiB is read) once through LDS-DMA. */
int hn_calib_mfma(int iters, float* sink_dev, uint64_t* t_dev, hnStream_t stream);
int hn_calib_stream(const void* buf_dev, long long n_bytes, float* sink_dev, uint64_t* t_dev, hnStream_t stream);
/* pattern 0: as hn_calib_stream (the workgroups share one moving window); 1: every workgroup streams a contiguous region
 * of its own, front to back — 256 far-apart sequential streams, the pattern of hn_wgrad_kernel's jobs. */
int hn_calib_stream_pattern(const void* buf_dev, long long n_bytes, int pattern, float* sink_dev, uint64_t* t_dev,
                            hnStream_t stream);
/* The DMA protocol of hn_wgrad_kernel alone: a ring of `stages` (3, 4, 5) stages of 32 KiB, counted vmcnt wait for the
 * oldest stage, proto 0: + a workgroup barrier per stage (the kernel's), proto 1: no barrier. */
int hn_calib_ring(const void* buf_dev, long long n_bytes, int stages, int proto, float* sink_dev, uint64_t* t_dev,
                  hnStream_t stream);

/* Debug/probe: runs one MFMA of each kind on identifiable data (layout self-test on real hardware). */
int hn_probe_mfma(float* out_bf16_acc, float* out_f32_acc, float* out_glds, hnStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
