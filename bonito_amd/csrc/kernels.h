// Internal launcher prototypes (one per HIP translation unit). The public C ABI in
// include/bonito_hip.h is a thin shell over these (bonito_amd/csrc/abi.cpp, engine.cpp).
// All pointers are device pointers; all launchers are asynchronous on `stream` and return
// 0 on success (error text via bh_last_error()).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// gemm.hip
int bh_k_linear(const void* X, const void* W, const float* bias, void* out, int M, int N, int K,
                int ldx, int ldw, int ldo, int act, float scale, float clamp_lo, float clamp_hi,
                int gated, int row_div, long row_s_hi, long row_s_lo, int row_lim, hipStream_t stream,
                const void* residual = nullptr, int ldres = 0, float res_scale = 1.0f);

void bh_k_linear_force_v1(int on);
void bh_k_linear_stagger(int units);
void bh_k_linear_order(int order);  // gemm_w4_kernel's work order inside an XCD: 0 token blocks fastest, 1 feature groups fastest
void bh_k_linear_gf(int gf);        // ... feature tiles per block (0 = automatic)
void bh_k_linear_tile16(int on);   // gemm_w4_kernel on 16x16x32 MFMAs (1) or 32x32x16 (0)

// conv.hip
int bh_k_conv_first(const void* signal, const float* w, const float* bias, void* out, int N, int Lin,
                    int Lout, int Cout, int K, int stride, int pad, int act, float clamp_lo,
                    float clamp_hi, long os_n, long os_t, hipStream_t stream);
int bh_k_conv_igemm(const void* in, const void* wpk, const float* bias, void* out, int N, int Lin,
                    int Lout, int Cin, int Cout, int K, int stride, int pad, int act, float clamp_lo,
                    float clamp_hi, long os_n, long os_t, hipStream_t stream);

// lstm.hip
int bh_k_lstm_layer(const void* gates_in, const void* whh_packed, void* h_out, int T, int N, int H,
                    int reverse, int* err_flag, hipStream_t stream, int n_rings, int* xcc_ws, int force_slow);
size_t bh_k_lstm_ws_bytes(int N, int H);
int bh_k_lstm_layer_stream(const void* gates_in, const void* whh_packed, void* h_out, int T, int N, int H,
                           int reverse, int* err_flag, hipStream_t stream, int n_rings, int* xcc_ws, int force_slow);
int bh_k_lstm_layer_fused(const void* x, const void* wih_packed, const float* bias, const void* whh_packed, void* h_out,
                          int T, int N, int H, int reverse, int* err_flag, hipStream_t stream, int n_rings, int* xcc_ws,
                          int force_slow);
int bh_k_fill_u16(void* dst, uint16_t value, size_t count, hipStream_t stream);
int bh_k_lstm_set_option(const char* name, int value);     // "lstm_max_spins"
size_t bh_k_lstm_packed_bytes(int H);
int bh_k_lstm_wg_units(int H);
int bh_k_lstm_cta_units(int H);
int bh_k_lstm_layer_cta(const void* x, const void* wih_tiles, const float* bias, const void* whh_tiles, void* h_out, int T, int N,
                        int H, int reverse, hipStream_t stream, int n_rings);

// crf.hip
int bh_k_crf_viterbi(const void* scores, int N, int T, int state_len, int layout_5s, float blank_score,
                     long s_n, long s_t, void* bp_ws, float* alpha_ws, int8_t* moves, int8_t* path,
                     float* best_score, hipStream_t stream);

int bh_k_crf_revcomp(const void* in, void* out, int N, int T, int state_len, int layout_5s, long s_n, long s_t,
                     hipStream_t stream);

// beam.hip
size_t bh_k_beam_workspace(int N, int T, int state_len);
int bh_k_crf_logz(const void* scores, int N, int T, int state_len, float blank, void* workspace, double* logz_out,
                   hipStream_t stream);
int bh_k_beam_search(const void* scores, int N, int T, int state_len, int beam_width, float beam_cut,
                     float blank, float q_scale, float q_offset, void* workspace, int8_t* sequence,
                     int8_t* qstring, int8_t* moves, float* qfloat, hipStream_t stream);

// attention.hip
int bh_k_attention(const void* qkv, void* out, const float* cos_sin, int N, int T, int nhead, int head_dim,
                   int win_left, int win_right, hipStream_t stream);
int bh_k_rmsnorm_residual(const void* a, const void* x, const float* w, void* out, long M, int D, float alpha,
                          float eps, hipStream_t stream);

// ctc.hip
int bh_k_dwconv(const void* in, const float* w, void* out, int N, int Lin, int Lout, int C, int K, int stride,
                int pad, hipStream_t stream);
int bh_k_ctc_head(const void* in, const float* w, const float* bias, void* out, long M, int features, int classes,
                  hipStream_t stream);
int bh_k_ctc_greedy(const float* logp, const long* offs, int R, int C, float qscale, float qbias, int8_t* seq,
                    int8_t* qual, int* path, int* count, hipStream_t stream);
size_t bh_k_ctc_beam_workspace(long total_steps, int R, int C, int beam_size);
int bh_k_ctc_prefix_beam(const float* logp, const long* offs, int R, int C, int beam_size, float threshold, void* workspace,
                         int8_t* labels, int* path, int* count, hipStream_t stream);
size_t bh_k_posterior_viterbi_workspace(int N, int T, int state_len);
int bh_k_posterior_viterbi(const void* scores, int N, int T, int state_len, float blank, void* workspace, int8_t* moves,
                           int8_t* path, hipStream_t stream);
int bh_k_decode_set_option(const char* name, int value);
namespace bh { extern int g_viterbi_quad; }       // crf.hip ("viterbi_quad")
int bh_k_conv_set_option(const char* name, int value);     // "conv_ws", "conv_fs", "conv_lds_kb", "conv_fuse"
// conv1 -> conv2 -> conv3 of an LSTM model's front end in one kernel (conv_front3_kernel); _ok: does the shape qualify?
int bh_k_conv_front3_ok(int c1_eff, int K1, int s1, int c2_in_eff, int c2_eff, int K2, int s2, int c3_in_eff, int c3_out, int K3, int s3);
int bh_k_conv_front3(const void* signal, int N, int L0, const float* w1, const float* b1, int K1, int pad1, int act1, float lo1, float hi1,
                     const void* w2pk, const float* b2, int K2, int pad2, int act2, float lo2, float hi2, const void* w3pk,
                     const float* b3, int Cout3, int K3, int stride3, int pad3, int act3, float lo3, float hi3, void* out, long os_n,
                     long os_t, hipStream_t stream);
int bh_k_linear_qkv_rotary(const void* X, const void* W, const float* bias, void* out, int M, int D, int K, const float* cos_sin,
                           int T, float qscale, hipStream_t stream);
int bh_k_attention_prerotated(const void* qkv, void* out, int N, int T, int nhead, int head_dim, int win_left, int win_right,
                              hipStream_t stream);
// signal.hip
int bh_k_signal_normalise(const int16_t* raw, const long* offs, const float* cal_scale, const float* cal_offset, int R,
                          int strategy, double qa, double qb, double shift_mult, double scale_mult, double fixed_shift,
                          double fixed_scale, int do_trim, double* shift, double* scale, int* weak, int* trim, hipStream_t stream);
int bh_k_signal_chunks(const int16_t* raw, const long* offs, const float* cal_scale, const float* cal_offset, const double* shift,
                       const double* scale, const int* weak, const int* chunk_read, const long* chunk_start, const long* chunk_len,
                       int n_chunks, int L, void* out, hipStream_t stream);
int bh_k_lstm_wide_ok(int H);
int bh_k_lstm_layer_wide(const void* gates_perm, const void* whh_tiles, void* h_out, int T, int N, int H, int reverse,
                         int* err_flag, hipStream_t stream, int n_rings, int* xcc_ws, int force_slow, void* ex = nullptr, int R = 0,
                         int arm = 0);
size_t bh_k_lstm_wide_ex_bytes(int N, int H);

// lstm_q8.hip: 8-bit recurrent path Q8-1
int bh_k_lstm_q8_units(int H, int variant);
size_t bh_k_lstm_q8_tile_bytes(int H);
int bh_k_lstm_q8_pack(const float* w, int H, int U, int8_t* packed, float* scale);
int bh_k_quantise_rows(const void* x, void* out, int T, int N, int H, int R, float bound, hipStream_t stream);
int bh_k_lstm_q8_arm(void* ex, int R, int H, hipStream_t stream);
int bh_k_lstm_layer_q8(const void* xq, const void* wih, const void* whh, const float* sx, const float* sh, const float* bias,
                       void* hq_out, void* h16_out, void* ex, int T, int N, int H, int R, int n_rings, int reverse, int* err_flag,
                       hipStream_t stream, int* xcc_ws, int flags, int variant, int* dbg, unsigned max_spins);
unsigned bh_k_lstm_max_spins();
size_t bh_k_lstm_wgx_ex_bytes(int N, int H);
int bh_k_lstm_layer_wgx(const void* x, const void* wih_tiles, const float* bias, const void* whh_tiles, void* h_out, void* ex, int T,
                        int N, int H, int R, int reverse, int* err_flag, hipStream_t stream, int n_rings, int* xcc_ws, int force_slow,
                        int arm);
int bh_k_lstm_layer_wgx2(const void* x, const void* wih_tiles, const float* bias, const void* whh_tiles, void* h_out, void* ex, int T,
                        int N, int H, int R, int reverse, int* err_flag, hipStream_t stream, int n_rings, int* xcc_ws, int force_slow,
                        int arm);      // two rings per workgroup: n_rings up to twice as many
