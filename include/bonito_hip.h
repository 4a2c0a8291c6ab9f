/* bonito_hip.h -- C ABI of libbonito_hip.so, the MI355X (gfx950) engine for bonito's
 * chunked-signal inference hot path:  signal chunks -> encoder -> CRF scores -> decode.
 *
 * This is the drop-in boundary.  Each entry point replaces one FFI/library call the reference makes
 * on this path (paths relative to /root/reference):
 *
 *   bh_encoder_create / bh_encoder_forward
 *        koi.lstm.update_graph(encoder, batchsize, chunksize, quantize)        bonito/crf/model.py:240-246
 *        + SeqdistModel.forward -> self.encoder(x) (cuDNN/cuBLAS/flash-attn)   bonito/crf/model.py:193-194
 *        + transformer use_koi (NTC, expand_blanks=False output)               bonito/transformer/model.py:136-146
 *        + bonito.ctc Model.forward (QuartzNet + log_softmax)                  bonito/ctc/model.py:35-37,195-207
 *   bh_beam_search
 *        koi.decode.beam_search(scores, beam_width, beam_cut, scale, offset, blank_score)
 *                                                                              bonito/crf/basecall.py:36-40
 *   bh_crf_viterbi / bh_crf_logz / bh_crf_posteriors
 *        koi.ctc.{logZ_cu_sparse, fwd_scores_cu_sparse, bwd_scores_cu_sparse}, SequenceDist.posteriors
 *        behind CTC_CRF.logZ / viterbi / decode_batch                          bonito/crf/model.py:47-67,98-103,196-199
 *   bh_ctc_greedy_decode / bh_ctc_beam_search
 *        fast_ctc_decode.viterbi_search / beam_search                          bonito/ctc/model.py:39-46
 *   bh_linear, bh_conv1d_*, bh_lstm_layer, ...  (operator level, used by the parity tests)
 *        torch.nn.Linear / Conv1d / LSTM kernels behind bonito/nn.py:27-38,222-241,396-415
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types.  Device pointers are raw HIP device addresses.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All launches are asynchronous;
 *     nothing here synchronises the device unless documented.
 *   - every function returns 0 on success, non-zero on error; bh_last_error() returns a
 *     thread-local message.  Nothing ever falls back to a CPU path.
 *   - caller owns input/output buffers; engines own their weights (copied at create) and workspace.
 *   - an engine handle is not re-entrant: one handle per GPU worker thread (the reference drives
 *     compute_scores from a single ThreadIterator, bonito/multiprocessing.py:20-24).
 */
#ifndef BONITO_HIP_H
#define BONITO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BH_ABI_VERSION 1

/* activations (bonito.nn `layers` registry names: swish, tanh, relu; nn.py:22-23,54-56) */
enum { BH_ACT_NONE = 0, BH_ACT_SWISH = 1, BH_ACT_TANH = 2, BH_ACT_RELU = 3 };

/* layer kinds understood by bh_encoder_create (bonito.nn registry names in comments) */
enum {
    BH_LAYER_CONV = 1,        /* convolution  (nn.py:222; BatchNorm folded by the caller as nn.py:447-454) */
    BH_LAYER_LSTM = 2,        /* lstm         (nn.py:396) */
    BH_LAYER_LINEAR_CRF = 3,  /* linearcrfencoder (nn.py:269) */
    BH_LAYER_CLAMP = 4,       /* clamp        (nn.py:60) -- fused into the producing kernel */
    BH_LAYER_TRANSFORMER = 5, /* transformerencoderlayer (transformer/model.py:83) */
    BH_LAYER_UPSAMPLE = 6,    /* linearupsample (nn.py:140) */
    BH_LAYER_TCS_BLOCK = 7,   /* reserved */
    BH_LAYER_CTC_DECODER = 8, /* bonito.ctc Decoder: 1x1 conv + log_softmax (ctc/model.py:195-207); final layer */
    BH_LAYER_DWCONV = 9,      /* depthwise half of TCSConv1d (ctc/model.py:99-103): groups = channels, no bias */
    BH_LAYER_RESIDUAL_PROJ = 10, /* Block.residual (ctc/model.py:166-167): 1x1 conv + folded BN of the block input,
                                  * kept aside and added by the next BH_LAYER_CONV whose `add_residual` is set */
    BH_LAYER_LINEAR = 11      /* linear (nn.py:27-52): y = x W^T + b on the feature axis, any layout
                               * (dna_r10.4.1@v4.0.toml:101-104: 1024 -> 256 between the LSTM stack and the CRF head) */
};

/* One layer of an encoder.  All weight pointers are HOST fp32 arrays in torch's native layout; the
 * engine packs / rounds them to fp16 once at create time.  Unused fields are 0 / NULL. */
typedef struct bh_layer {
    int32_t kind;
    int32_t in_size;     /* conv: in channels; lstm/linear: in features; transformer: d_model */
    int32_t out_size;    /* conv: out channels; lstm: hidden; linear_crf: n_base^(state_len+1) (+ blanks if no blank_score) */
    int32_t winlen, stride, padding;   /* conv */
    int32_t activation;  /* BH_ACT_* */
    int32_t reverse;     /* lstm */
    int32_t nhead, dim_ff, win_left, win_right;   /* transformer */
    int32_t scale_factor;               /* upsample */
    int32_t groups;      /* conv: 1 */
    int32_t add_residual;               /* conv (pointwise): add the pending residual projection before the activation */
    int32_t quantize;    /* lstm: 1 = 8-bit recurrent path Q8-1 where the kernel covers the shape (koi's `quantize`, crf/model.py:245) */
    int32_t reserved_i[1];
    float scale;         /* linear_crf: multiply after activation (0 = none) */
    float clamp_lo, clamp_hi;           /* clamp */
    float blank_score;   /* linear_crf with fixed blank (scores keep the 4S koi layout) */
    float alpha;         /* transformer: deepnorm_alpha */
    float eps;           /* transformer: rmsnorm eps */
    float reserved_f[2];
    const float* w0;     /* conv [Cout][Cin/groups][K]; lstm W_ih [4H][I]; linear [out][in]; transformer Wqkv [3D][D] */
    const float* b0;     /* conv bias [Cout]; lstm b_ih [4H]; linear bias; transformer: NULL */
    const float* w1;     /* lstm W_hh [4H][H]; transformer out_proj.weight [D][D] */
    const float* b1;     /* lstm b_hh [4H]; transformer out_proj.bias [D] */
    const float* w2;     /* transformer ff.fc1.weight [2F][D] */
    const float* w3;     /* transformer ff.fc2.weight [D][F] */
    const float* w4;     /* transformer norm1.weight [D] */
    const float* w5;     /* transformer norm2.weight [D] */
} bh_layer_t;

typedef struct bh_encoder bh_encoder_t;

const char* bh_last_error(void);
int bh_abi_version(void);
/* sizeof(bh_layer_t) as compiled into the library (bindings check their struct mirror against it) */
size_t bh_sizeof_layer(void);
/* number of visible HIP devices, or <0 on error */
int bh_device_count(void);

/* ---- encoder engine -------------------------------------------------------------------------- */
/* Build an engine for a linear chain of layers on HIP device `device`.  Workspace is sized for
 * batches of up to max_batch chunks of up to max_chunk samples (batch is padded to 16 internally). */
int bh_encoder_create(const bh_layer_t* layers, int n_layers, int device, int max_batch, int max_chunk,
                      bh_encoder_t** out);
void bh_encoder_destroy(bh_encoder_t* enc);
/* output geometry for chunks of L samples: T output steps, C scores per step, stride = L-per-step */
int bh_encoder_output_shape(const bh_encoder_t* enc, int L, int* T, int* C, int* stride);
/* signal: device fp16 [N][L] (the reference's [N,1,L] batch, bonito/crf/basecall.py:33).
 * scores: device fp16, contiguous [N][T][C] (the layout koi.decode.beam_search consumes). */
int bh_encoder_forward(bh_encoder_t* enc, const void* signal, int N, int L, void* scores, void* stream);
/* one text line per layer naming the kernels the engine launches for it (measurement / logging) */
int bh_encoder_describe(const bh_encoder_t* enc, char* buf, size_t bytes);
/* "lstm_exchange" (1 default: the workgroup-shared fp16 recurrent kernel hands h_t over through a small L2-resident ring
 * buffer and writes the output tensor separately; 0: through the sentinel-filled output tensor - the per-wave
 * lstm_layer_fused_kernel for H <= 512, the wide kernel's older hand-off above that; same bytes, tests).
 * "lstm_pair" (1 default: a batch of more rings than one launch of that kernel holds - more than 512 chunks at H = 384 - is served
 * two rings per workgroup on one register-resident copy of the weights, lstm_layer_wgx2_kernel; 0: one launch per 32 rings). */
/* tuning / test options (results never change): "lstm_fused" (3 default: narrowest applicable fused kernel; 2, 1, 0 = older
 * variants down to projection-by-GEMM), "lstm_force_slow" (0/1: write-through exchange), "lstm_wide" (1/0), "lstm_prefill"
 * (1 default: sentinel fill of the next recurrent layer's buffer on a side stream), "attn_ring" (1/0), "lstm_tune" (bit mask) */
int bh_encoder_set_option(bh_encoder_t* enc, const char* name, int value);
/* debug: read back the LSTM workspace (XCD agreement slots, per-wave cycle statistics when lstm_tune bit 2 is set) */
int bh_encoder_debug_read(bh_encoder_t* enc, void* host, size_t bytes, size_t offset);
/* Device-side timeouts. The persistent recurrent kernels bound every spin; a kernel that gives up raises a flag and finishes with
 * INVALID output. Flags are kept PER FORWARD: every bh_encoder_forward gets a ticket (0, 1, 2, ... per engine,
 * bh_encoder_last_ticket right after the call), zeroes its own slot in front of its kernels and ends with a 4-byte copy of the slot
 * into pinned host memory on its stream. Once the caller has observed the completion of forward `ticket` (event, decoded outputs on
 * the host, stream synchronise), bh_encoder_error_flag_at(ticket) says - without a device round trip - whether THAT forward timed
 * out; other forwards in flight are not consumed or cleared by the query (the product pipeline retries exactly the flagged batch).
 * At most 64 forwards of one engine may be in flight / unqueried (-1 for a ticket whose slot has been recycled).
 * bh_encoder_error_flag: non-zero iff any forward since the last bh_encoder_check has been SEEN to time out (host side only).
 * bh_encoder_check: synchronises `stream`, returns the same and forgets it (reference seam: bonito/crf/basecall.py:27-45 has
 * nothing that can time out; a drop-in must neither abort nor emit calls decoded from invalid scores). */
int bh_encoder_check(bh_encoder_t* enc, void* stream);
int bh_encoder_error_flag(const bh_encoder_t* enc);
long bh_encoder_last_ticket(const bh_encoder_t* enc);
int bh_encoder_error_flag_at(const bh_encoder_t* enc, long ticket);
/* The caller has handled the timeout of forward `ticket` (it re-ran the batch): forget that flag - it is then reported neither by
 * bh_encoder_error_flag / bh_encoder_check nor when its slot is recycled. 0, or -1 for a ticket whose slot has been recycled already.
 * bh_encoder_forward may run on one host thread while the four flag queries run on another (all flag state is atomic). */
int bh_encoder_ack(bh_encoder_t* enc, long ticket);

/* Per-kernel-class timing with HIP events recorded on the forward's stream (measurement only).
 * After enabling, every bh_encoder_forward appends spans; profile_read synchronises on them and returns
 * accumulated milliseconds and span counts per class, then clears. */
enum { BH_PROF_CONV = 0, BH_PROF_LSTM_GEMM = 1, BH_PROF_FILL = 2, BH_PROF_LSTM_REC = 3, BH_PROF_CRF_LINEAR = 4,
       BH_PROF_ATTENTION = 5 /* Wqkv + rotary, out_proj, the first residual norm */, BH_PROF_MLP = 6 /* fc2 + the second residual norm */,
       BH_PROF_OTHER = 7, BH_PROF_MLP_FC1 = 8 /* fc1 + SwiGLU alone: one launch per span */,
       BH_PROF_ATTENTION_CORE = 9 /* the attention kernel alone: one launch per span */, BH_PROF_CLASSES = 10 };
int bh_encoder_profile(bh_encoder_t* enc, int enable);
int bh_encoder_profile_read(bh_encoder_t* enc, float* ms /*[BH_PROF_CLASSES]*/, int* spans /*[BH_PROF_CLASSES]*/);

/* ---- CRF decode ------------------------------------------------------------------------------ */
/* Bytes of device workspace bh_crf_viterbi needs. */
size_t bh_crf_viterbi_workspace(int N, int T, int state_len);
/* Max-semiring best path (CTC_CRF.viterbi, crf/model.py:98-103).
 * scores fp16 with element strides (stride_n, stride_t); layout_5s=1: C=5S with the stay score in
 * column 5j (expand_blanks layout); layout_5s=0: C=4S koi layout + scalar blank_score.
 * moves[N][T] in {0,1}; path[N][T] in {0..4} (0 = no emission, else 1+base); best[N] path score (or NULL). */
int bh_crf_viterbi(const void* scores, int N, int T, int state_len, int layout_5s, float blank_score,
                   long stride_n, long stride_t, void* workspace, int8_t* moves, int8_t* path,
                   float* best, void* stream);

/* CTC_CRF.reverse_complement (crf/model.py:84-96): permute scores so that decoding yields the reverse-complement
 * strand.  layout_5s=0: koi layout (stride_n = T*4S, stride_t = 4S for contiguous NTC); layout_5s=1: [T][N][5S]
 * (stride_n = 5S, stride_t = N*5S).  Out of place. */
int bh_crf_reverse_complement(const void* in, void* out, int N, int T, int state_len, int layout_5s,
                              long stride_n, long stride_t, void* stream);
/* CTC_CRF.logZ (crf/model.py:47-52), Log semiring, on contiguous koi-layout scores: logz[N] (device double).
 * workspace: bh_beam_search_workspace(N, T, state_len) bytes. */
int bh_crf_logz(const void* scores, int N, int T, int state_len, float blank_score, void* workspace, double* logz,
                void* stream);

/* Signal ingest on the device: replaces Read.__init__'s numpy work (bonito/reader.py:122-166 normalisation + trim, the pA
 * scaling of bonito/pod5.py:52-67) and util.chunk + the fp16 cast (bonito/util.py:142-161, crf/basecall.py:31) for raw
 * int16 reads, with the reference's arithmetic reproduced bit for bit. All pointers are device pointers.
 *   raw: concatenated int16 samples of n_reads reads, offsets[n_reads + 1]; cal_scale / cal_offset: per-read calibration
 *   (pA = cal_scale * (raw + cal_offset)); strategy 0 = quantile scaling with (quantile_a, quantile_b, shift_mult,
 *   scale_mult), 1 = fixed (fixed_shift, fixed_scale). Outputs per read: shift, scale (fp64), weak (bit0: shift is the
 *   literal 10, bit1: scale is the literal 1.0 -- NumPy's promotion then differs), trim (first sample of the read proper).
 * bh_signal_chunks writes normalised fp16 rows [n_chunks][chunk_samples]: row i = read chunk_read[i], samples
 *   chunk_start[i] .. (chunk_len[i] >= chunk_samples) or the chunk_len[i] available samples tiled (short reads). */
int bh_signal_normalise(const int16_t* raw, const long* offsets, const float* cal_scale, const float* cal_offset, int n_reads,
                        int strategy, double quantile_a, double quantile_b, double shift_mult, double scale_mult,
                        double fixed_shift, double fixed_scale, int do_trim, double* shift, double* scale, int* weak, int* trim,
                        void* stream);
int bh_signal_chunks(const int16_t* raw, const long* offsets, const float* cal_scale, const float* cal_offset,
                     const double* shift, const double* scale, const int* weak, const int* chunk_read, const long* chunk_start,
                     const long* chunk_len, int n_chunks, int chunk_samples, void* out, void* stream);

/* Operator level (parity tests): one recurrent layer of the 8-bit path Q8-1 (koi's `quantize`, bonito/crf/model.py:245; the
 * arithmetic is defined in oracle/lstm_q8_ref.py) from fp32 HOST weights W_ih, W_hh [4H][H] and bias [4H] (b_ih + b_hh, or NULL).
 * x: device fp16 [T][N][H], N % 16 == 0, quantised with the static scale 127 / bound; h16_out: device fp16 [T][N][H];
 * hq_frag (or NULL): device int8 output in MFMA fragment order [T][N/16][ceil(H/64)][64][16]; sums (or NULL): device int32
 * [T][N][4H][2] = the exact integer sums (input part, recurrent part) per gate row. variant: see "lstm_q8_variant". Synchronises. */
int bh_lstm_q8_layer(const void* x, float bound, const float* w_ih, const float* w_hh, const float* bias, int T, int N, int H,
                     int reverse, int variant, void* h16_out, int8_t* hq_frag, int32_t* sums, void* stream);

/* Process-wide knobs (measurement / tuning hooks, no reference counterpart).
 *   "beam_fork": -1 auto (default: where the scan is a kernel of its own - up to 64 states, and 1024 states, where the beam kernel
 *                is one wave per chunk), 0 = run the posterior scan behind the beam kernel on the caller's stream,
 *                1 = run it next to the beam kernel on an internal helper stream (joined before finalize).
 *   "beam_fuse": -1 (default) = auto: fused for <= 256 states; 1 = the forward / posterior scan runs as a second wave inside the beam kernel's workgroups and reads
 *                the score and guide rows from the blocks the beam wave stages in LDS (the score tensor is read from HBM once
 *                for both); 0 = separate crf_forward_post_kernel as in round 1 (then "beam_fork" applies).
 *   "beam_cpw": chunks per workgroup of the fused beam kernel at 256 states: 0 (default) = the smallest of 1 / 2 / 4 that lets all
 *                chunks of the call be resident at once (5 / 6 / 8 chunks per CU; the kernels are latency chains, so chunks in
 *                flight per CU are what counts: 2048 x 1667 steps 12.0 -> 8.7 ms on MI355X); 1, 2, 4 force a geometry. Same bytes.
 *   "attn_waves": 0 (default) = automatic, 8 / 12 = waves per workgroup of the ring attention kernel (query blocks of 128 / 192; same results).
 *   "beam_select": 0 (default) = top-W selection by histogram + exact boundary ranking, 1 = MSB-first radix search
 *                (the same beams either way; kept for regression tests and A/B timing).
 *   "conv_ws": 1 (default) = weight-stationary kernel for the 384-channel / 19-tap convolution, 0 = generic implicit GEMM.
 *   "conv_fuse": 1 (default) = conv1 -> conv2 -> conv3 of a 384-channel LSTM stack as one kernel (intermediates in LDS), 2 = also
 *                for the 96-channel stacks, 0 = three kernels. "conv_fs": 1 (default) = feature-split instances of the implicit-GEMM
 *                kernel for layers with a multiple of 64 output channels, 0 = position-split. "conv_lds_kb": LDS a workgroup of that
 *                kernel may take for its input span (default 64). All of them: identical bytes (tests).
 *   "gemm_path": 0 auto (default), 1 = register-staged 128x128x64 kernel only, 2 = never a 256x256x64 kernel, 3 = never the
 *                four-wave kernel (gemm_w4_kernel; the eight-wave one where it applies), 5 = the four-wave kernel for every legal shape.
 *   "gemm_tile16": 1 (default) = the four-wave kernel's K-tile stream on 16x16x32 MFMAs with its epilogues in the accumulator layout,
 *                0 = the 32x32x16 stream of rounds 4-5 (results agree to fp16 rounding: another accumulation order).
 *                "gemm_order": 1 (default) = an XCD sweeps all feature groups of a block of token tiles before it moves on, 0 = token
 *                blocks fastest; "gemm_gf": feature tiles per block of that order (0 = automatic). Same bytes either way.
 *   "lstm_q8_variant": geometry of the 8-bit recurrent kernel chosen at bh_encoder_create: 0 (default) = 12 / 16 units per wave,
 *                one workgroup per CU; 1 = 4 units per wave, three workgroups per CU; 2 = 12 units per wave compiled for two
 *                workgroups per CU, so that the recurrent kernels of two engines (two batches in flight) share every CU and each
 *                hides the other's exchange round trip (both: hidden size 384 only).
 *   "lstm_max_spins": bound of the recurrent kernels' exchange spin loops (default 1000000; < 0 restores it). Tests lower it
 *                to provoke the timeout path (bh_encoder_error_flag / bh_encoder_check). */
int bh_set_option(const char* name, int value);

/* Posterior decoding = SeqdistModel.decode_batch (crf/model.py:196-199): Viterbi over log(edge posteriors + 1e-8).
 * scores: contiguous koi layout [N][T][4S]; moves/path as bh_crf_viterbi.
 * workspace: bh_crf_posterior_viterbi_workspace(N, T, state_len) bytes. */
size_t bh_crf_posterior_viterbi_workspace(int N, int T, int state_len);
int bh_crf_posterior_viterbi(const void* scores, int N, int T, int state_len, float blank_score, void* workspace,
                             int8_t* moves, int8_t* path, void* stream);

/* Beam-search decode: drop-in for koi.decode.beam_search(scores, beam_width=32, beam_cut=100.0, scale=1.0,
 * offset=0.0, blank_score=2.0) (bonito/crf/basecall.py:27,36-40).  scores: device fp16 contiguous
 * [N][T][4^(state_len+1)] (koi layout).  Outputs are DEVICE int8 [N][T], zero where nothing is emitted:
 * sequence = ASCII base at emitting steps, qstring = 33 + round(q), moves in {0,1}; qfloat (optional,
 * device fp32 [N][T]) receives the un-rounded q.  workspace: bh_beam_search_workspace(N, T, state_len) bytes.
 * Algorithm "BS-2" (round 5: the guide and the posterior scan in the linear domain, a deterministic exponential shared with the oracle;
 * DESIGN.md 5); bit-exact against oracle/crf_oracle.c for sequence and moves, q within 1e-3 of its fp64 posteriors.
 * Limits: 1 <= state_len <= 5, 1 <= beam_width <= 32, beam_cut >= 1, T < 131072 steps per chunk. */
size_t bh_beam_search_workspace(int N, int T, int state_len);
int bh_beam_search(const void* scores, int N, int T, int state_len, int beam_width, float beam_cut,
                   float blank_score, float q_scale, float q_offset, void* workspace, int8_t* sequence,
                   int8_t* qstring, int8_t* moves, float* qfloat, void* stream);

/* ---- operator level (parity tests, custom pipelines) ----------------------------------------- */
/* out[m][n] = clamp(act(X[m][:] . W[n][:] + bias[n]) * scale); fp16 X [M][ldx], W [N][ldw], out [.][ldo].
 * gated=1: SwiGLU epilogue over interleaved rows (out has N/2 columns).
 * row remap: out_row = (m / row_div) * row_s_hi + (m % row_div) * row_s_lo   (row_div=0: identity);
 * rows with (m % row_div) >= row_lim are skipped (row_lim=0: none) -- used to drop batch padding. */
int bh_linear(const void* X, const void* W, const float* bias, void* out, int M, int N, int K, int ldx,
              int ldw, int ldo, int act, float scale, float clamp_lo, float clamp_hi, int gated,
              int row_div, long row_s_hi, long row_s_lo, int row_lim, void* stream);
/* first convolution, Cin = 1: signal fp16 [N][Lin] -> out (n*os_n + t*os_t + c), w fp32 [Cout][K] on device */
int bh_conv1d_first(const void* signal, const float* w, const float* bias, void* out, int N, int Lin,
                    int Cout, int K, int stride, int pad, int act, float clamp_lo, float clamp_hi,
                    long os_n, long os_t, void* stream);
/* packed conv weight size in halves and host-side packer: torch [Cout][Cin][K] fp32 -> [Cout16][Kp] fp16 */
size_t bh_conv1d_packed_halves(int Cin, int Cout, int K);
int bh_conv1d_pack(const float* w, int Cin, int Cout, int K, uint16_t* packed);
/* channel-minor implicit-GEMM conv: in fp16 [N][Lin][Cin] -> out (n*os_n + t*os_t + c) */
int bh_conv1d(const void* in, const void* wpacked, const float* bias, void* out, int N, int Lin, int Cin,
              int Cout, int K, int stride, int pad, int act, float clamp_lo, float clamp_hi, long os_n,
              long os_t, void* stream);
/* Greedy CTC decode of R reads in one launch: replaces fast_ctc_decode.viterbi_search(probs, alphabet,
 * qstring=True, qscale, qbias) (bonito/ctc/model.py:39-42).  logp: device fp32 [sum T_r][classes]
 * log-probabilities, offsets: device int64 [R+1].  Outputs (device, compacted per read at offsets[r]):
 * labels (1..classes-1), qual (phred chars), path (step of each base), count[R]. */
int bh_ctc_greedy_decode(const float* logp, const long* offsets, int R, int classes, float qscale, float qbias,
                         int8_t* labels, int8_t* qual, int* path, int* count, void* stream);
/* CTC prefix beam search of R reads in one launch: replaces fast_ctc_decode.beam_search(probs, alphabet, beam_size=5,
 * beam_cut_threshold=1e-3) (bonito/ctc/model.py:44).  Same buffers as bh_ctc_greedy_decode (no qualities);
 * workspace: bh_ctc_beam_search_workspace(sum T_r, R, classes, beam_size) device bytes. beam_size <= 16. */
size_t bh_ctc_beam_search_workspace(long total_steps, int R, int classes, int beam_size);
int bh_ctc_beam_search(const float* logp, const long* offsets, int R, int classes, int beam_size, float threshold,
                       void* workspace, int8_t* labels, int* path, int* count, void* stream);
/* depthwise conv (TCSConv1d.depthwise, ctc/model.py:99-103): in/out fp16 channel-minor [N][L][C], w fp32 device [C][K] */
int bh_dwconv1d(const void* in, const float* w, void* out, int N, int Lin, int C, int K, int stride, int pad,
                void* stream);
/* rotary cos/sin table (host): out[t][i][0..1] = cos, sin(t * 10000^(-2i/dim)), i < dim/2, fp32
 * (flash_attn.layers.rotary.RotaryEmbedding(dim, interleaved=False), bonito/transformer/model.py:55,73) */
int bh_rotary_table(int T, int dim, float* out);
/* rotary + sliding-window attention on packed qkv (flash_attn_qkvpacked_func(qkv, window_size=(l, r)),
 * bonito/transformer/model.py:58-66): qkv fp16 [N*T][3*nhead*head_dim] -> out fp16 [N*T][nhead*head_dim];
 * cos_sin = device copy of bh_rotary_table(T, head_dim). head_dim must be 64. */
int bh_attention(const void* qkv, void* out, const float* cos_sin, int N, int T, int nhead, int head_dim,
                 int win_left, int win_right, void* stream);
/* The same windowed attention on packed qkv whose q and k ALREADY carry the rotary embedding and whose q is scaled by log2(e) / sqrt(head_dim)
 * (what the engine's Wqkv epilogue writes): the persistent ring-buffer kernel the engine runs for windows with left <= 128 and
 * left + right <= 256 (flash_attn_qkvpacked_func proper, bonito/transformer/model.py:60; softmax in base 2 on pre-scaled scores). */
int bh_attention_prerotated(const void* qkv, void* out, int N, int T, int nhead, int head_dim, int win_left, int win_right, void* stream);
/* out = rmsnorm(a + alpha * x) * w (fp32 statistics): RMSNorm(x, residual) of bonito/transformer/model.py:110-111,125-128 */
int bh_rmsnorm_residual(const void* a, const void* x, const float* w, void* out, long M, int D, float alpha,
                        float eps, void* stream);
/* recurrent weights: torch W_hh [4H][H] fp32 (host) -> MFMA-fragment order fp16 (host, 4*H*H halves) */
int bh_lstm_pack_whh(const float* whh, int H, uint16_t* packed);
/* host helper of the formatting stage = koi.decode.to_str before the text decode (bonito/crf/basecall.py:48-55): copies the
 * non-zero bytes of src[0..n) to dst (capacity n) in order and returns how many there were. No device work. */
long bh_host_compact(const int8_t* src, long n, char* dst);
/* host helper of the chunking stage: rows [row0, row0 + nrows) of util.chunk(signal[0..T), chunksize, overlap)
 * (bonito/util.py:142-161; T >= chunksize) cast to fp16 (round to nearest even) into dst[nrows][chunksize].
 * Returns nrows, or < 0 on bad arguments. No device work. */
long bh_host_chunk_rows(const float* signal, long T, int chunksize, int overlap, long row0, long nrows, uint16_t* dst);

/* One basecalled read -> its FASTQ (mode 0) / FASTA (1) / unaligned SAM (2) record, in one call: the stitching of
 * bonito/util.py:164-183 (via crf/basecall.py:13-24: the kept window of every chunk, `reverse` included), koi's to_str
 * (crf/basecall.py:48-55), the rna flip, the mean q-score filter and the record text with the tags of bonito/io.py:135-166
 * (RG:Z, qs:f, ns:i, ts:i, mv:B:c). The read's chunks are n_pieces runs of consecutive rows of decoded planes: piece i = rows
 * [lo[i], lo[i] + rows[i]) of an int8 array [3][n][T] (sequence, qstring, moves) at base[i] whose planes are plane_stride[i] bytes
 * apart. Returns the bytes written to out; 0 = filtered out (empty sequence or mean q < min_qscore; seq_len / mean_q are set
 * regardless); -1 = bad arguments; < -1 = -(bytes of out needed). */
long bh_host_format_read(const int8_t* const* base, const long* plane_stride, const long* lo, const long* rows, int n_pieces,
                         long T, long length, int chunksize, int overlap, int stride, int reverse, int rna, int mode,
                         double min_qscore, const char* read_id, const char* run_id, long num_samples, long trimmed_samples,
                         char* out, long out_cap, long* seq_len, double* mean_q);
/* Mean q-score of a phred string, averaged in error-probability space (bonito/util.py mean_qscore_from_qstring). */
double bh_host_mean_qscore(const char* qstring, long n);
/* pod5 signal codec, inner layer (bonito_amd/pod5.py; replaces the pod5 wheel behind /root/reference bonito/pod5.py:52 `read.signal`):
 * streamvbyte-16 over the zig-zag code of the first differences -> `count` int16 samples. `in` is the zstd-DEcompressed block.
 * Returns the bytes consumed, -1 when the input is too short. Host code, no device involved. */
long bh_host_svb16_decode(const uint8_t* in, long n_in, long count, int16_t* out);
/* one LSTM layer over gates_in = x W_ih^T + b (fp16 [T][N][4H], torch gate order); h_out fp16 [T][N][H].
 * N % 16 == 0.  workspace: bh_lstm_workspace(N, H) device bytes.  err_flag: device int, set non-zero on a
 * device-side timeout.  flags bit 0: force the placement-independent write-through exchange policy;
 * bit 1: force the weight-streaming kernel (always used for H > 512; needs H % 64 == 0, H <= 1024). */
size_t bh_lstm_workspace(int N, int H);
int bh_lstm_layer(const void* gates_in, const void* whh_packed, void* h_out, int T, int N, int H,
                  int reverse, void* workspace, int* err_flag, int flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BONITO_HIP_H */
