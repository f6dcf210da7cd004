// C ABI of the kernel launchers (defined in the .hip files of this directory).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "diag.hpp"

extern "C" {
// "release" for the shipped build, "diag" for a library compiled with -DDTF_DIAG=1 (csrc/kernels/diag.hpp)
const char* dtf_build_flavour();
int dtf_conv1_fwd(const void* images, const void* idx, const void* w, const void* bias, void* pooled, void* arg,
                  int B, int cout, int grid, hipStream_t s);
int dtf_conv1_fwd_sampled(const void* images, void* idx_out, const void* state, unsigned n, unsigned hb,
                          unsigned seed, int shuffle, const void* w, const void* bias, void* pooled, void* arg,
                          int B, int cout, int grid, hipStream_t s);
int dtf_conv1_wgrad(const void* images, const void* idx, const void* gpool, const void* arg,
                    void* part, int B, int cout, int grid, hipStream_t s);
int dtf_conv2_pack_weights(const void* w, void* packed, int cin, int cout, hipStream_t s);
int dtf_conv2_fwd(const void* x, const void* wT, const void* bias, void* pooled, void* arg, int B, int cin, int cout,
                  hipStream_t s);
int dtf_conv12_fwd(const void* images, const void* idx, void* idx_out, const void* state, unsigned n, unsigned hb,
                   unsigned seed, int shuffle, const void* w1, const void* b1, void* p1, void* arg1, const void* wT,
                   const void* bias2, void* p2, void* arg2, int B, hipStream_t s);
int dtf_conv2_dgrad(const void* gpool, const void* arg, const void* w, void* dx, int B, int cin,
                    int cout, hipStream_t s);
int dtf_conv2_dgrad_conv1_wgrad_g(const void* gpool, const void* arg2, const void* w, const void* images,
                                  const void* idx, const void* arg1, void* part1, int B, int G, hipStream_t s);
int dtf_conv2_small_batch(void);
int dtf_conv2_dgrad_conv1_wgrad(const void* gpool, const void* arg2, const void* w, const void* images,
                                const void* idx, const void* arg1, void* part1, int B, hipStream_t s);
int dtf_conv2_wgrad(const void* x, const void* gpool, const void* arg, void* part, int B, int cin,
                    int cout, int G, hipStream_t s);
int dtf_conv2_wgrad_nt(const void* x, const void* gpool, const void* arg, void* part, int B, int cin, int cout, int G,
                       int nt, hipStream_t s);
int dtf_gemm_nt(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int lda, int ldb,
                int ldc, int epi, float keep_prob, uint64_t seed, const void* state, int tile, hipStream_t s);
int dtf_gemm_tn_ev(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int tile,
                   hipEvent_t done, hipStream_t s);
int dtf_gemm_tn_g16_ev(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                       hipEvent_t done, hipStream_t s);
int dtf_gemm_tn(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int tile,
                hipStream_t s);
int dtf_gemm_tn_sgd(const void* A, const void* B, void* W, void* shadow, const void* hyper, int M, int N, int K,
                    int lda, int ldb, int ldw, hipStream_t s);
int dtf_gemm_nn_fwd(const void* A, const void* B, void* part, void* out, const void* bias, int M, int N, int K,
                    int lda, int ldb, int S, float keep_prob, uint64_t seed, const void* state, hipStream_t s);
int dtf_transpose_bf16(const void* in, void* out, int batch, int R, int C, hipStream_t s);
int dtf_head_fwd_bwd(const void* h1, const void* wT, const void* bias, const void* labels, const void* idx,
                     void* logits_out, void* loss_rows, void* correct_rows, void* dlogits, void* da, int B, int H,
                     float grad_scale, float inv_keep, int backward, hipStream_t s);
int dtf_head_set_onepass(int on);   // 0: the general head kernel also at H == 1024 (tests, A/B)
int dtf_head_fused_fwd_bwd(const void* part, int S, const void* bias1, float keep_prob, uint64_t seed,
                           const void* state, void* h1, const void* wT, const void* bias, const void* labels,
                           const void* idx, void* logits_out, void* loss_rows, void* correct_rows, void* dlogits,
                           void* da, int B, int H, float grad_scale, float inv_keep, int backward, hipStream_t s);
int dtf_transpose_head_w(const void* w, void* wT, int H, hipStream_t s);
int dtf_dropout_mask(uint64_t seed, uint64_t offset, float keep_prob, void* out, long n, hipStream_t s);
int dtf_metrics_reduce(const void* loss_rows, const void* correct_rows, int n, void* acc, int accumulate, hipStream_t s);
int dtf_gemm_nn_partial(const void* A, const void* B, void* part, int M, int N, int K, int lda, int ldb, int S,
                        hipStream_t s);
int dtf_head_wgrad(const void* h1, const void* dlogits, const void* da, void* part, int B, int H, int KS,
                   hipStream_t s);
// fc1 dgrad GEMM (C = A . B^T, bf16 store) + the head wgrad as extra workgroups of the same launch
int dtf_gemm_nt_head_wgrad(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                           const void* h1, const void* dlogits, const void* da, void* part, int Bsz, int H, int KS,
                           hipStream_t s);
int dtf_sgd_update(void* w, const void* g, void* shadow, const void* hyper, long n, void* state, long cursor_inc,
                   long step_inc, long drop_inc, hipStream_t s);
int dtf_adam_update(void* w, const void* g, void* m, void* v, void* shadow, const void* hyper, long n, void* tstate,
                    int adam_slot, void* state, long cursor_inc, long step_inc, long drop_inc, hipStream_t s);
int dtf_sgd_update_g16(void* w, const void* g, void* shadow, const void* hyper, long n, void* state, long cursor_inc,
                       long step_inc, long drop_inc, hipStream_t s);
int dtf_adam_update_g16(void* w, const void* g, void* m, void* v, void* shadow, const void* hyper, long n, void* tstate,
                        int adam_slot, void* state, long cursor_inc, long step_inc, long drop_inc, hipStream_t s);
int dtf_cast_bf16(const void* w, void* shadow, long n, hipStream_t s);
int dtf_reduce_segments(int count, const void** part, void** out, const int* G, const int* n, const long* stride,
                        hipStream_t s);
int dtf_fused_reduce_sgd(int count, const void** part, const int* G, const int* n, const long* stride,
                         const long* seg_off, void* w, void* g, void* shadow, const void* hyper, long dense_lo,
                         long dense_hi, void* wc2T, long wc2_off, int c1, int c2, void* out_wT, long out_w_off, int H,
                         void* state, long cursor_inc, long step_inc, long drop_inc, hipStream_t s);
int dtf_fused_reduce_adam(int count, const void** part, const int* G, const int* n, const long* stride,
                          const long* seg_off, void* w, void* g, void* m, void* v, void* shadow, const void* hyper,
                          long dense_lo, long dense_hi, void* wc2T, long wc2_off, int c1, int c2, void* out_wT,
                          long out_w_off, int H, void* state, int adam_slot, long cursor_inc, long step_inc,
                          long drop_inc, hipStream_t s);
// ---- data-parallel exchange fabric (dp.hip): peer-to-peer reduce-scatter / sharded update / all-gather of the dense
// gradient and the one-shot exchange + commit of the tail, with every per-step counter in device memory ----
// Mirrored by ctypes in distributed_tensorflow_amd/parallel/dp_fabric.py.
struct DtfDpFabric {
  void* base[8];               // every rank's uncached exchange buffer as mapped into THIS process (base[rank]: own)
  int world, rank;
  int dense_blocks;            // workgroups of the dense exchange kernel (<= 256, the same on every rank)
  int commit_blocks;           // workgroups of the tail kernel that commit the dense shard (<= 256)
  long n_tail, n_tail_max;     // floats, multiples of 4
  long n_dense, csz;           // floats: the dense range (wd1) and this job's shard size n_dense / world
  int ag_bytes;                // what the all-gather carries: 2 = bf16 compute copy, 4 = fp32 weights (fp32 engine)
  int reserved;
  // byte offsets of the areas inside every rank's buffer (dtf_dp_fabric_layout fills them)
  long off_tail_stage, off_tail_flags, off_tail_done, off_dense_a, off_dense_b, off_flag_a, off_flag_b, off_done_a,
      off_done_b;
  // local, ordinary device memory of this rank
  unsigned* seq_tail;          // [64 + commit_blocks] per-workgroup sequence numbers of the tail kernel
  unsigned* seq_dense;         // [dense_blocks] the same for the dense kernel
  int* ctl;                    // [0] poison (sticky) [1] why: 1 a wait timed out, 2 a peer was poisoned
                               // [2] sequence number of the last dense exchange that completed here
                               // [3] test hook: the step with this sequence number raises no flags
  float* gsum;                 // [csz] the summed gradient of this rank's shard (dense kernel -> commit)
  long long timeout_ticks;     // bound of every wait, in 100 MHz wall-clock ticks
};
long dtf_dp_fabric_layout(DtfDpFabric* f);   // fills the offsets; returns the bytes every rank allocates
// dense exchange of one step on `s` (a branch next to the conv backward); adam_t = &state[slot] or NULL for SGD
int dtf_dp_dense_exchange(const DtfDpFabric* f, int adam, const void* grads_dense, const void* w_dense,
                          const void* m_dense, const void* v_dense, void* shadow_dense, const void* hyper,
                          const void* adam_t, hipStream_t s);
// tail exchange + launch-wide verdict + commit (tail update, step state, and -- with_dense -- this rank's shard of the
// dense master weights / moments from f->gsum, plus the foreign fp32 shards when ag_bytes == 4)
int dtf_dp_tail_commit(const DtfDpFabric* f, int adam, int with_dense, void* w, void* g, void* m, void* v, void* shadow,
                       const void* hyper, void* wc2T, long wc2_off, int c1, int c2, void* out_wT, long out_w_off, int H,
                       void* state, int adam_slot, int adam_slot2, long cursor_inc, long step_inc, long drop_inc,
                       long tail_off, long dense_off, hipStream_t s);
int dtf_sample_indices(const void* state, void* idx, int B, unsigned n, unsigned hb, unsigned seed, int shuffle,
                       hipStream_t s);

// ---- exact-fp32 kernel set (--dtype fp32): fp32 activations / weights, f32-input MFMA (csrc/kernels/*_f32.hip) ----
int dtf_conv1_f32_fwd(const void* images, const void* idx, const void* w, const void* bias, void* pooled, void* arg, int B,
                      int cout, int grid, hipStream_t s);
int dtf_conv1_f32_fwd_sampled(const void* images, void* idx_out, const void* state, unsigned n, unsigned hb, unsigned seed,
                              int shuffle, const void* w, const void* bias, void* pooled, void* arg, int B, int cout,
                              int grid, hipStream_t s);
int dtf_conv1_f32_wgrad(const void* images, const void* idx, const void* gpool, const void* arg, void* part, int B, int cout,
                        int grid, hipStream_t s);
int dtf_u8_to_f32_unit(const void* src, void* dst, long n, hipStream_t s);
int dtf_conv2_f32_pack_weights(const void* w, void* packed, int cin, int cout, hipStream_t s);
int dtf_conv2_f32_fwd(const void* x, const void* wpack, const void* bias, void* pooled, void* arg, int B, int cin, int cout,
                      hipStream_t s);
int dtf_conv2_f32_dgrad(const void* gpool, const void* arg, const void* wpack, void* dx, int B, int cin, int cout,
                        hipStream_t s);
int dtf_conv2_f32_wgrad(const void* x, const void* gpool, const void* arg, void* part, int B, int cin, int cout, int G,
                        hipStream_t s);
int dtf_gemm_f32_nn_partial(const void* A, const void* B, void* part, int M, int N, int K, int lda, int ldb, int S,
                            hipStream_t s);
int dtf_gemm_f32_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, hipStream_t s);
int dtf_gemm_f32_nt_head_wgrad(const void* A, const void* B, void* splitk_part, void* C, int M, int N, int K, int lda,
                               int ldb, int ldc, int S, const void* h1, const void* dlogits, const void* da, void* part,
                               int Bsz, int H, int KS, hipStream_t s);
int dtf_gemm_f32_nt_splitk(const void* A, const void* B, void* part, void* C, int M, int N, int K, int lda, int ldb,
                           int ldc, int S, hipStream_t s);
int dtf_gemm_f32_tn_ev(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                       const void* hyper, hipEvent_t done, hipStream_t s);
int dtf_gemm_f32_tn(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                    const void* hyper, hipStream_t s);
// stream-K variant (gemm_f32.hip gemm_f32_sk) and its per-engine uncached workspace; ws == NULL: the plain launch
int dtf_gemm_f32_tn_sk(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                       const void* hyper, void* ws, hipEvent_t done, hipStream_t s);
long dtf_gemm_f32_sk_workspace_bytes();
void* dtf_gemm_f32_sk_workspace_alloc();
void dtf_gemm_f32_sk_workspace_free(void* p);
int dtf_head_f32_fwd_bwd(const void* h1, const void* wT, const void* bias, const void* labels, const void* idx,
                         void* logits_out, void* loss_rows, void* correct_rows, void* dlogits, void* da, int B, int H,
                         float grad_scale, float inv_keep, int backward, hipStream_t s);
int dtf_head_f32_fused_fwd_bwd(const void* part, int S, const void* bias1, float keep_prob, uint64_t seed,
                               const void* state, void* h1, const void* wT, const void* bias, const void* labels,
                               const void* idx, void* logits_out, void* loss_rows, void* correct_rows, void* dlogits,
                               void* da, int B, int H, float grad_scale, float inv_keep, int backward, hipStream_t s);
int dtf_head_f32_wgrad(const void* h1, const void* dlogits, const void* da, void* part, int B, int H, int KS,
                       hipStream_t s);
// asynchronous parameter server over peer-to-peer mailboxes (ps.hip)
int dtf_ps_blocks();
int dtf_ps_max_workers();
int dtf_ps_push(const void* grads, void* dst, void* flags, void* hdr, const void* version, int nr, const long* lo,
                const long* hi, long n_total, unsigned seq, unsigned kind, hipStream_t s);
// resident service loop of a ps task (ps_serve): control block in uncached memory, summary in pinned host memory
long dtf_ps_ctl_bytes();
int dtf_ps_out_words();
int dtf_ps_serve_init(void* ctl, int W, unsigned long long step, hipStream_t s);
int dtf_ps_ctl_set_dropped(void* ctl, unsigned mask, hipStream_t s);
int dtf_ps_serve(int adam, void* ctl, void* own, void* const* boxes, int W, int NB, int p, long n_total, long n_dense,
                 const long* layout9, int nr, const long* lo, const long* hi, void* w, void* m, void* v, const void* hyper,
                 unsigned adam_t0, unsigned long long training_iter, unsigned round0, double budget_s, double idle_s,
                 void* out, hipStream_t s);
int dtf_ps_install(const void* box_flags, const void* box_hdr, const void* box_f32, const void* box_bf16, long n_dense,
                   int nr, const long* lo, const long* hi, long n_total, unsigned seq, void* params, void* shadow,
                   const void* full, const void* full_flags, void* full_dst, void* state, void* version, void* host_hdr,
                   int owner, int* status, double timeout_s, hipStream_t s);
}
