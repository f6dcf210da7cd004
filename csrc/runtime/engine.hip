// Native step executor: the whole training step (sampler, forward, loss, backward with its
// fork/join side streams, slab reductions, optimizer, shadow refresh) is issued from C++ so one
// foreign call from Python enqueues a phase.  This is the replacement for the TF C++ executor that
// runs the reference's `sess.run([optimizer, global_step], feed_dict)` (reference
// .idea/MNISTDist.py:188): Python only decides WHAT to run; launching, stream forks/joins and
// event plumbing are native.  Everything is enqueue-only (no allocation, no sync), so any phase
// can be captured into a HIP graph.
#include "api.h"

extern "C" {

// Plain-old-data description of one engine instance; mirrored by ctypes in engine.py.
struct DtfEngineDesc {
  // model / batch
  int B, c1, c2, hidden, flat, classes;
  int head_ks, c2_g, c1_g, optimizer /*0 sgd, 1 adam*/, shuffle, multi_stream;
  int fc1_splits;    // split-K factor the partial-product buffer fc1_part was sized for (B rows each)
  int c2_nt;         // bf16 conv2 wgrad: 16-wide co tiles per wave (2 | 4), 0 = the kernel's choice by model
  int dgrad_splits;  // fp32 fc1 dgrad at a small batch: K ranges summed from dgrad_part (1: plain GEMM)
  int c1w_imgs;      // fused conv2 dgrad -> conv1 wgrad kernel: images per workgroup (2, or 1 = the small-batch launch
                     // with c1_g == B slabs); 0: two kernels
  unsigned ds_n, ds_hb, sample_seed, fuse_dense_update;  // last: fc1 SGD update inside its wgrad GEMM (1 GPU)
  uint64_t seed;
  float keep_prob;
  int dtype;   // 0: bf16 compute copies + fp32 master weights; 1: exact fp32 end to end (csrc/kernels/*_f32.hip)
  long n_total;
  long off_out_w, off_out_b, off_bd1, off_wd1, off_wc2, off_bc2, off_wc1, off_bc1;  // float offsets in the arena
  long head_stride, c2_stride, c1_stride;
  // device buffers
  void *params, *grads, *shadow, *fc1_part, *wc2T, *adam_m, *adam_v, *state, *hyper;
  void *idx, *p1, *arg1, *p2, *arg2, *h1, *logits, *loss_rows, *correct_rows, *dlogits, *da, *dflat, *dp1;
  void *head_part, *c2_part, *c1_part, *dgrad_part;
  void* dense_g16;   // bf16 [flat * hidden] or NULL: the data-parallel step's wd1 gradient in 16 bits (sync_dp.py grad_comm)
  void* out_wT;  // fp32 [classes][hidden]: transposed output weight for the head kernel
  void *ds_images, *ds_labels;
  // native-owned handles (filled by dtf_engine_init)
  hipStream_t side1, side2;
  hipEvent_t ev_head, ev_fork, ev_join1, ev_join2, ev_dgrad;
  hipEvent_t ev_dense_ready, ev_dense_updated, ev_tail_ready, ev_tail_reduced;  // data-parallel step
  void* sk_ws;   // fp32: uncached stream-K workspace of the fc1 wgrad GEMM (gemm_f32.hip gemm_f32_sk), or NULL
};

#define DTF_TRY(x)            \
  do {                        \
    int rc_ = (x);            \
    if (rc_ != 0) return rc_; \
  } while (0)
#define HIP_TRY(x)                          \
  do {                                      \
    hipError_t e_ = (x);                    \
    if (e_ != hipSuccess) return (int)e_;   \
  } while (0)

static inline float* f32at(void* base, long off) { return (float*)base + off; }
static inline char* bf16at(void* base, long off) { return (char*)base + 2 * off; }
static inline bool is_f32(const DtfEngineDesc* e) { return e->dtype == 1; }
// the fp32 configuration keeps no bf16 copies: optimizer kernels are told so with shadow == NULL
static inline void* shadow_of(const DtfEngineDesc* e) { return is_f32(e) ? nullptr : e->shadow; }

const char* dtf_build_flavour() { return DTF_DIAG ? "diag" : "release"; }

int dtf_engine_init(DtfEngineDesc* e) {
  // side streams only for the backward fork/join (multi_stream): a process has 4 hardware queues by default
  // (GPU_MAX_HW_QUEUES), and the data-parallel step already uses compute + comm + RCCL's own streams
  e->side1 = e->side2 = nullptr;
  if (e->multi_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&e->side1, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&e->side2, hipStreamNonBlocking));
  }
  hipEvent_t* evs[9] = {&e->ev_head,        &e->ev_fork,          &e->ev_join1,      &e->ev_join2,       &e->ev_dgrad,
                        &e->ev_dense_ready, &e->ev_dense_updated, &e->ev_tail_ready, &e->ev_tail_reduced};
  // Every one of these events orders streams of THIS device only, so a device-scope release is all it has to do.
  // The default (system-scope) release writes the caches back when the event is recorded: measured as a 5.5-7.5 us
  // hole in the compute stream behind each record that follows a 12.8 MB producer (the fc1 GEMMs).
  const unsigned flags = hipEventDisableTiming | ((DTF_EXP_DP & 8) ? 0u : (unsigned)hipEventReleaseToDevice);
  for (auto ev : evs) HIP_TRY(hipEventCreateWithFlags(ev, flags));
  e->sk_ws = is_f32(e) ? dtf_gemm_f32_sk_workspace_alloc() : nullptr;   // NULL (no uncached memory): the plain launch
  return 0;
}

int dtf_engine_destroy(DtfEngineDesc* e) {
  hipEvent_t evs[9] = {e->ev_head,        e->ev_fork,          e->ev_join1,      e->ev_join2,       e->ev_dgrad,
                       e->ev_dense_ready, e->ev_dense_updated, e->ev_tail_ready, e->ev_tail_reduced};
  for (auto ev : evs)
    if (ev) (void)hipEventDestroy(ev);
  if (e->side1) (void)hipStreamDestroy(e->side1);
  if (e->side2) (void)hipStreamDestroy(e->side2);
  dtf_gemm_f32_sk_workspace_free(e->sk_ws);
  e->sk_ws = nullptr;
  return 0;
}

// number of HIP streams this engine created itself (the caller's compute / comm streams come on top)
int dtf_engine_stream_count(const DtfEngineDesc* e) { return (e->side1 ? 1 : 0) + (e->side2 ? 1 : 0); }

int dtf_stream_wait_event(hipStream_t s, hipEvent_t ev) { return (int)hipStreamWaitEvent(s, ev, 0); }
int dtf_event_record(hipEvent_t ev, hipStream_t s) { return (int)hipEventRecord(ev, s); }

int dtf_engine_sample(const DtfEngineDesc* e, hipStream_t s) {
  return dtf_sample_indices(e->state, e->idx, e->B, e->ds_n, e->ds_hb, e->sample_seed, e->shuffle, s);
}

// both conv blocks; the 1 -> 32 -> 64 model runs them as ONE kernel (conv1 computed inside conv2's
// staging phase, csrc/kernels/conv2.hip Conv1Fused).  sampled != 0: rows come from the device sample
// cursor (the batch sampler is part of the kernel) and are published in e->idx.
static int forward_conv_impl(const DtfEngineDesc* e, const void* images, const void* idx, int sampled, int n,
                             hipStream_t s) {
  const void* state = sampled ? e->state : nullptr;
  if (n < 1 || n > e->B) return -3;      // every activation buffer holds B rows
  if (is_f32(e)) {
    if (sampled)
      DTF_TRY(dtf_conv1_f32_fwd_sampled(images, e->idx, e->state, e->ds_n, e->ds_hb, e->sample_seed, e->shuffle,
                                        f32at(e->params, e->off_wc1), f32at(e->params, e->off_bc1), e->p1, e->arg1, n,
                                        e->c1, 0, s));
    else
      DTF_TRY(dtf_conv1_f32_fwd(images, idx, f32at(e->params, e->off_wc1), f32at(e->params, e->off_bc1), e->p1, e->arg1,
                                n, e->c1, 0, s));
    return dtf_conv2_f32_fwd(e->p1, e->wc2T, f32at(e->params, e->off_bc2), e->p2, e->arg2, n, e->c1, e->c2, s);
  }
  if (e->c1 == 32 && e->c2 == 64)
    return dtf_conv12_fwd(images, sampled ? nullptr : idx, e->idx, state, e->ds_n, e->ds_hb, e->sample_seed, e->shuffle,
                          f32at(e->params, e->off_wc1), f32at(e->params, e->off_bc1), e->p1, e->arg1, e->wc2T,
                          f32at(e->params, e->off_bc2), e->p2, e->arg2, n, s);
  if (sampled)
    DTF_TRY(dtf_conv1_fwd_sampled(images, e->idx, e->state, e->ds_n, e->ds_hb, e->sample_seed, e->shuffle,
                                  f32at(e->params, e->off_wc1), f32at(e->params, e->off_bc1), e->p1, e->arg1, n, e->c1,
                                  0, s));
  else
    DTF_TRY(dtf_conv1_fwd(images, idx, f32at(e->params, e->off_wc1), f32at(e->params, e->off_bc1), e->p1, e->arg1, n,
                          e->c1, 0, s));
  return dtf_conv2_fwd(e->p1, e->wc2T, f32at(e->params, e->off_bc2), e->p2, e->arg2, n, e->c1, e->c2, s);
}

int dtf_engine_forward_conv(const DtfEngineDesc* e, const void* images, const void* idx, int n, hipStream_t s) {
  return forward_conv_impl(e, images, idx, 0, n, s);
}

// Split-K factor of the fc1 forward GEMM for n rows.  The partial-product buffer (engine.py fc1_part) holds
// fc1_splits * B rows; a ragged evaluation chunk n < B has fewer tiles and would by itself pick MORE splits than
// fit there (e.g. hidden 4096, B = 1024: one split; n = 784: two splits of 784 rows), so the choice is capped.
// The fp32 kernels take up to 32 splits (batch 100: 8 output tiles of 128 x 128, 98 K tiles -> 8 x 25 workgroups
// instead of 8 x 8; the bf16 GEMM keeps the 8 it was tuned with).
static int auto_splits(const DtfEngineDesc* e, int M, int N) {
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  const int S = (256 + tiles - 1) / tiles;
  // bf16: 8 splits, 16 for a single row of tiles (batch <= 128: 0.0709 -> 0.0697 ms at batch 100; 32 splits and
  // 16 splits at batch 256 were measured slower -- the head kernel reads every slab)
  const int cap = is_f32(e) ? 32 : (tiles <= 8 ? 16 : 8);
  return S > cap ? cap : S;
}
static int fc1_splits_for(const DtfEngineDesc* e, int n) {
  const long cap_rows = (long)e->fc1_splits * e->B;
  int S = auto_splits(e, n, e->hidden);
  while (S > 1 && (long)S * n > cap_rows) --S;
  return S;
}

int dtf_engine_forward_dense(const DtfEngineDesc* e, const void* labels, const void* idx, int backward, int n,
                             int train_mode, hipStream_t s) {
  const float keep = train_mode ? e->keep_prob : 1.0f;
  if (n < 1 || n > e->B) return -3;
  if (is_f32(e)) {
    const int S = dtf_gemm_f32_nn_partial(e->p2, f32at(e->params, e->off_wd1), e->fc1_part, n, e->hidden, e->flat, e->flat,
                                          e->hidden, fc1_splits_for(e, n), s);
    if (S <= 0) return S == 0 ? -1 : S;
    return dtf_head_f32_fused_fwd_bwd(e->fc1_part, S, f32at(e->params, e->off_bd1), keep, e->seed, e->state, e->h1,
                                      e->out_wT, f32at(e->params, e->off_out_b), labels, idx, e->logits, e->loss_rows,
                                      e->correct_rows, e->dlogits, e->da, n, e->hidden, 1.0f / n, 1.0f / keep, backward, s);
  }
  // wd1 is consumed in its native [in, out] layout (no transposed copy), split-K; the bias + ReLU +
  // dropout epilogue runs inside the head kernel, which turns the partial sums into h1 on the fly
  const int S = dtf_gemm_nn_partial(e->p2, bf16at(e->shadow, e->off_wd1), e->fc1_part, n, e->hidden, e->flat, e->flat,
                                    e->hidden, fc1_splits_for(e, n), s);
  if (S <= 0) return S == 0 ? -1 : S;
  return dtf_head_fused_fwd_bwd(e->fc1_part, S, f32at(e->params, e->off_bd1), keep, e->seed, e->state, e->h1, e->out_wT,
                                f32at(e->params, e->off_out_b), labels, idx, e->logits, e->loss_rows, e->correct_rows,
                                e->dlogits, e->da, n, e->hidden, 1.0f / n, 1.0f / keep, backward, s);
}

// conv2 dgrad followed by the conv1 filter gradient.  The 1 -> 32 -> 64 model runs both as ONE kernel
// (dx stays in LDS; csrc/kernels/conv2.hip Conv1WgradFused) which writes one slab per workgroup -- (B + 1) / 2 with
// two images per workgroup, B with one (c1w_imgs; engine.py sizes c1_g accordingly).
static int dgrad_conv1_wgrad(const DtfEngineDesc* e, const void* images, const void* idx, hipStream_t s) {
  if (is_f32(e)) {
    DTF_TRY(dtf_conv2_f32_dgrad(e->dflat, e->arg2, f32at(e->wc2T, 25L * e->c1 * e->c2), e->dp1, e->B, e->c1, e->c2, s));
    return dtf_conv1_f32_wgrad(images, idx, e->dp1, e->arg1, e->c1_part, e->B, e->c1, e->c1_g, s);
  }
  if (e->c1 == 32 && e->c2 == 64 && idx && e->c1w_imgs > 0 && e->c1_g == (e->B + e->c1w_imgs - 1) / e->c1w_imgs)
    return dtf_conv2_dgrad_conv1_wgrad_g(e->dflat, e->arg2, bf16at(e->wc2T, 25L * e->c1 * e->c2), images, idx, e->arg1,
                                         e->c1_part, e->B, e->c1_g, s);
  DTF_TRY(dtf_conv2_dgrad(e->dflat, e->arg2, bf16at(e->wc2T, 25L * e->c1 * e->c2), e->dp1, e->B, e->c1, e->c2, s));
  return dtf_conv1_wgrad(images, idx, e->dp1, e->arg1, e->c1_part, e->B, e->c1, e->c1_g, s);
}

// conv2 filter gradient slabs (c2_nt: the bf16 kernel's co tiles per wave, chosen with c2_g in engine.py; 0 = by model)
static int conv2_wgrad(const DtfEngineDesc* e, hipStream_t s) {
  if (is_f32(e)) return dtf_conv2_f32_wgrad(e->p1, e->dflat, e->arg2, e->c2_part, e->B, e->c1, e->c2, e->c2_g, s);
  return dtf_conv2_wgrad_nt(e->p1, e->dflat, e->arg2, e->c2_part, e->B, e->c1, e->c2, e->c2_g, e->c2_nt, s);
}

// head wgrad + slab reduce + fc1 wgrad: finalises the (head | wd1) prefix of the gradient arena.
// reduce == 0 leaves the head slabs unreduced (the fused single-GPU SGD tail sums them itself);
// reduce == -1 does not compute them either: the training step runs the head wgrad inside the fc1
// dgrad launch (backward_conv_impl with head_w != 0).
// done != NULL: that event is complete when the fc1 wgrad GEMM is (bound to the kernel: no marker in the stream).
// g16 != 0: the fc1 weight gradient is written as bf16 into e->dense_g16 instead of the gradient arena.
static int backward_dense_impl(const DtfEngineDesc* e, int reduce, hipStream_t s, hipEvent_t done = nullptr, int g16 = 0) {
  if (reduce >= 0)
    DTF_TRY((is_f32(e) ? dtf_head_f32_wgrad : dtf_head_wgrad)(e->h1, e->dlogits, e->da, e->head_part, e->B, e->hidden,
                                                              e->head_ks, s));
  if (reduce > 0) {
    const void* part[1] = {e->head_part};
    void* out[1] = {f32at(e->grads, e->off_out_w)};
    int G[1] = {e->head_ks}, n[1] = {(int)e->head_stride};
    long st[1] = {e->head_stride};
    DTF_TRY(dtf_reduce_segments(1, part, out, G, n, st, s));
  }
  if (is_f32(e))
    return dtf_gemm_f32_tn_sk(e->p2, e->da, f32at(e->grads, e->off_wd1), e->flat, e->hidden, e->B, e->flat, e->hidden,
                              e->hidden, nullptr, e->sk_ws, done, s);
  if (g16)
    return dtf_gemm_tn_g16_ev(e->p2, e->da, e->dense_g16, e->flat, e->hidden, e->B, e->flat, e->hidden, e->hidden, done, s);
  return dtf_gemm_tn_ev(e->p2, e->da, f32at(e->grads, e->off_wd1), e->flat, e->hidden, e->B, e->flat, e->hidden,
                        e->hidden, 0, done, s);
}
int dtf_engine_backward_dense(const DtfEngineDesc* e, hipStream_t s) { return backward_dense_impl(e, 1, s); }

// fc1 dgrad -> {conv2 dgrad -> conv1 wgrad} || conv2 wgrad (side stream) -> slab reduce.
// record_dgrad != 0: e->ev_dgrad is recorded right after fc1 dgrad (the last reader of bf16 wd1).
static int fc1_dgrad(const DtfEngineDesc* e, int head_w, hipStream_t s) {
  if (is_f32(e)) {
    if (head_w)   // + the output-layer weight gradient as extra workgroups of the same launch
      return dtf_gemm_f32_nt_head_wgrad(e->da, f32at(e->params, e->off_wd1), e->dgrad_part, e->dflat, e->B, e->flat,
                                        e->hidden, e->hidden, e->hidden, e->flat, e->dgrad_splits, e->h1, e->dlogits, e->da,
                                        e->head_part, e->B, e->hidden, e->head_ks, s);
    return dtf_gemm_f32_nt_splitk(e->da, f32at(e->params, e->off_wd1), e->dgrad_part, e->dflat, e->B, e->flat, e->hidden,
                                  e->hidden, e->hidden, e->flat, e->dgrad_splits, s);
  }
  if (head_w)   // + the output-layer weight gradient as extra workgroups of the same launch
    return dtf_gemm_nt_head_wgrad(e->da, bf16at(e->shadow, e->off_wd1), e->dflat, e->B, e->flat, e->hidden, e->hidden,
                                  e->hidden, e->flat, e->h1, e->dlogits, e->da, e->head_part, e->B, e->hidden,
                                  e->head_ks, s);
  return dtf_gemm_nt(e->da, bf16at(e->shadow, e->off_wd1), e->dflat, nullptr, e->B, e->flat, e->hidden, e->hidden,
                     e->hidden, e->flat, 0, 1.0f, 0, nullptr, 0, s);
}

// head_w < 0: fc1 dgrad (and with it the head wgrad) has already been issued by the caller
static int backward_conv_impl(const DtfEngineDesc* e, const void* images, const void* idx, int record_dgrad,
                              int reduce, int head_w, hipStream_t s, bool dgrad_first = false) {
  if (head_w >= 0) DTF_TRY(fc1_dgrad(e, head_w, s));
  if (record_dgrad) HIP_TRY(hipEventRecord(e->ev_dgrad, s));
  hipStream_t s2 = s;
  if (e->multi_stream) {
    HIP_TRY(hipEventRecord(e->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(e->side2, e->ev_fork, 0));
    s2 = e->side2;
  }
  // dgrad_first (one stream): in the data-parallel step the dense exchange (comm stream) starts together with this
  // backward; conv5x5_wgrad_pc fills a CU (twelve waves x 168 registers, 146 KiB of LDS), so exchange workgroups that
  // are not placed before it wait for its 25 us -- next to the dgrad kernel (two 4-wave workgroups per CU) they fit
  // (bench.py --force_dp --no_collapse with the shard of a 2 / 4 / 8-rank job: -3.7 / -1.9 / -0.8 % step time).
  // Without a concurrent exchange the other order is the faster one (the slab sums right behind conv2 wgrad cost
  // +5 us, measured with --force_dp), so it stays the default.  And ONLY when every rank has a GPU to itself: with
  // two ranks time-sliced on one device (the test harness, --rehearse_one_gpu) the exchange workgroups of one rank,
  // placed early and spinning, kept the other rank's kernels off the device for seconds at a time -- 24 steps took
  // 3 s, 36 s, 57 s, > 60 s in four consecutive runs (tools/scratch/sync_p2p_repro.py), 3 s in all eight the other way.
  static int dgrad_first_env = -1;   // DTF_DGRAD_FIRST=0|1 overrides (A/B switch)
  if (dgrad_first_env < 0) {
    const char* v = getenv("DTF_DGRAD_FIRST");
    dgrad_first_env = (v && *v) ? (atoi(v) ? 1 : 0) : 2;
  }
  if (dgrad_first_env != 2) dgrad_first = dgrad_first_env == 1;
  if (!e->multi_stream && dgrad_first) {
    DTF_TRY(dgrad_conv1_wgrad(e, images, idx, s));
    DTF_TRY(conv2_wgrad(e, s));
  } else {
    DTF_TRY(conv2_wgrad(e, s2));
    if (e->multi_stream) HIP_TRY(hipEventRecord(e->ev_join2, s2));
    DTF_TRY(dgrad_conv1_wgrad(e, images, idx, s));
    if (e->multi_stream) HIP_TRY(hipStreamWaitEvent(s, e->ev_join2, 0));
  }
  if (!reduce) return 0;
  // reduce == 2: the head slabs (written by backward_dense with reduce == 0) are summed here too, in
  // the same launch -- the data-parallel step needs head + conv reduced only for its tail collective
  // ... and, as a fourth "segment" of one 4-float slab, this rank's control flags (hyper[8..12): stop / gather
  // requests) are published into the control block at the end of the gradient arena, where the tail collective
  // sums them over the ranks: fresh every step, no launch of their own
  const void* part[4] = {e->c2_part, e->c1_part, e->head_part, f32at(e->hyper, 8)};
  void* out[4] = {f32at(e->grads, e->off_wc2), f32at(e->grads, e->off_wc1), f32at(e->grads, e->off_out_w),
                  f32at(e->grads, e->n_total - 4)};
  int G[4] = {e->c2_g, e->c1_g, e->head_ks, 1}, n[4] = {(int)e->c2_stride, (int)e->c1_stride, (int)e->head_stride, 4};
  long st[4] = {e->c2_stride, e->c1_stride, e->head_stride, 4};
  return dtf_reduce_segments(reduce == 2 ? 4 : 2, part, out, G, n, st, s);
}
int dtf_engine_backward_conv(const DtfEngineDesc* e, const void* images, const void* idx, int record_dgrad,
                             hipStream_t s) {
  return backward_conv_impl(e, images, idx, record_dgrad, 1, 0, s);
}

// whole backward; the dense-layer weight gradients run on side1 next to the conv chain
static int backward_impl(const DtfEngineDesc* e, const void* images, const void* idx, int reduce, hipStream_t s) {
  // the head slabs come out of the fc1 dgrad launch and are reduced with the conv slabs (one launch)
  if (!e->multi_stream) {
    DTF_TRY(backward_dense_impl(e, -1, s));
    return backward_conv_impl(e, images, idx, 0, reduce ? 2 : 0, 1, s);
  }
  HIP_TRY(hipEventRecord(e->ev_head, s));
  HIP_TRY(hipStreamWaitEvent(e->side1, e->ev_head, 0));
  DTF_TRY(backward_dense_impl(e, -1, e->side1));
  HIP_TRY(hipEventRecord(e->ev_join1, e->side1));
  DTF_TRY(backward_conv_impl(e, images, idx, 0, reduce ? 2 : 0, 1, s));
  return (int)hipStreamWaitEvent(s, e->ev_join1, 0);
}
int dtf_engine_backward(const DtfEngineDesc* e, const void* images, const void* idx, hipStream_t s) {
  return backward_impl(e, images, idx, 1, s);
}

static int update_range_src(const DtfEngineDesc* e, long lo, long hi, int advance, int advance_cursor, long step_inc,
                            int adam_slot, const void* gsrc, hipStream_t s, int g16 = 0);
int dtf_engine_update_range(const DtfEngineDesc* e, long lo, long hi, int advance, int advance_cursor, long step_inc,
                            int adam_slot, hipStream_t s) {
  return update_range_src(e, lo, hi, advance, advance_cursor, step_inc, adam_slot, nullptr, s);
}
// gsrc != nullptr: the gradient of [lo, hi) is read there instead of from the gradient arena (the staging buffer of
// the peer-to-peer all-reduce, which then does not have to copy its result into the arena); g16: gsrc holds bf16
static int update_range_src(const DtfEngineDesc* e, long lo, long hi, int advance, int advance_cursor, long step_inc,
                            int adam_slot, const void* gsrc, hipStream_t s, int g16) {
  const long n = hi - lo;
  const void* g = gsrc ? gsrc : (const void*)f32at(e->grads, lo);
  void* st = advance ? e->state : nullptr;
  const long cinc = (advance && advance_cursor) ? e->B : 0;
  if (g16) {
    if (!gsrc || is_f32(e)) return -3;
    if (e->optimizer == 0)
      return dtf_sgd_update_g16(f32at(e->params, lo), g, bf16at(e->shadow, lo), e->hyper, n, st, cinc, step_inc, 1, s);
    return dtf_adam_update_g16(f32at(e->params, lo), g, f32at(e->adam_m, lo), f32at(e->adam_v, lo), bf16at(e->shadow, lo),
                               e->hyper, n, e->state, adam_slot, st, cinc, step_inc, 1, s);
  }
  if (e->optimizer == 0)
    DTF_TRY(dtf_sgd_update(f32at(e->params, lo), g, is_f32(e) ? nullptr : bf16at(e->shadow, lo), e->hyper, n, st, cinc,
                           step_inc, 1, s));
  else
    DTF_TRY(dtf_adam_update(f32at(e->params, lo), g, f32at(e->adam_m, lo), f32at(e->adam_v, lo),
                            is_f32(e) ? nullptr : bf16at(e->shadow, lo), e->hyper, n, e->state, adam_slot, st, cinc,
                            step_inc, 1, s));
  // the head kernel reads the output weight transposed: keep that copy current
  if (lo <= e->off_out_w && e->off_out_w < hi)
    DTF_TRY(dtf_transpose_head_w(f32at(e->params, e->off_out_w), e->out_wT, e->hidden, s));
  return 0;
}

// only wc2 needs a derived copy: its two fragment-ordered packs (2 x 100 KB, conv2.hip); wd1 is
// used in place by both fc1 GEMMs
int dtf_engine_transposes(const DtfEngineDesc* e, int wc2, hipStream_t s) {
  if (wc2 && is_f32(e)) DTF_TRY(dtf_conv2_f32_pack_weights(f32at(e->params, e->off_wc2), e->wc2T, e->c1, e->c2, s));
  else if (wc2) DTF_TRY(dtf_conv2_pack_weights(bf16at(e->shadow, e->off_wc2), e->wc2T, e->c1, e->c2, s));
  return dtf_transpose_head_w(f32at(e->params, e->off_out_w), e->out_wT, e->hidden, s);
}

// Backward + update of the single-GPU SGD step with the fc1 weight update fused into its wgrad
// GEMM (gemm_tn_bf16<SGD>): dW_d1 -- 12.8 of the 13.1 MB of gradients -- is applied straight from
// the accumulators and never written.  fc1 dgrad reads the bf16 wd1, so it runs first.
static int train_tail_fused_dense(const DtfEngineDesc* e, long step_inc, hipStream_t s) {
  DTF_TRY(fc1_dgrad(e, 1, s));
  hipStream_t s1 = s, s2 = s;
  if (e->multi_stream) {
    HIP_TRY(hipEventRecord(e->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(e->side1, e->ev_fork, 0));
    HIP_TRY(hipStreamWaitEvent(e->side2, e->ev_fork, 0));
    s1 = e->side1;
    s2 = e->side2;
  }
  if (is_f32(e))
    DTF_TRY(dtf_gemm_f32_tn_sk(e->p2, e->da, f32at(e->params, e->off_wd1), e->flat, e->hidden, e->B, e->flat, e->hidden,
                               e->hidden, e->hyper, e->sk_ws, nullptr, s1));
  else
    DTF_TRY(dtf_gemm_tn_sgd(e->p2, e->da, f32at(e->params, e->off_wd1), bf16at(e->shadow, e->off_wd1), e->hyper, e->flat,
                            e->hidden, e->B, e->flat, e->hidden, e->hidden, s1));
  DTF_TRY(conv2_wgrad(e, s2));
  DTF_TRY(dgrad_conv1_wgrad(e, e->ds_images, e->idx, s));
  if (e->multi_stream) {
    HIP_TRY(hipEventRecord(e->ev_join1, s1));
    HIP_TRY(hipEventRecord(e->ev_join2, s2));
    HIP_TRY(hipStreamWaitEvent(s, e->ev_join1, 0));
    HIP_TRY(hipStreamWaitEvent(s, e->ev_join2, 0));
  }
  const void* part[3] = {e->head_part, e->c2_part, e->c1_part};
  int G[3] = {e->head_ks, e->c2_g, e->c1_g}, n[3] = {(int)e->head_stride, (int)e->c2_stride, (int)e->c1_stride};
  long st[3] = {e->head_stride, e->c2_stride, e->c1_stride};
  long off[3] = {e->off_out_w, e->off_wc2, e->off_wc1};
  // dense range empty: wd1 is already updated
  return dtf_fused_reduce_sgd(3, part, G, n, st, off, e->params, e->grads, shadow_of(e), e->hyper, e->off_wc2, e->off_wc2,
                              e->wc2T, e->off_wc2, e->c1, e->c2, e->out_wT, e->off_out_w, e->hidden, e->state, e->B, step_inc, 1, s);
}

// one full optimisation step on the resident dataset.  SGD (the reference optimizer) ends in the
// fused tail: slab reductions + update + bf16 shadow + K-contiguous wc2 copy in one launch.
int dtf_engine_train_step(const DtfEngineDesc* e, long step_inc, hipStream_t s) {
  // the batch sampler runs inside conv1 forward (rows from the device cursor, published in e->idx)
  DTF_TRY(forward_conv_impl(e, e->ds_images, nullptr, 1, e->B, s));
  DTF_TRY(dtf_engine_forward_dense(e, e->ds_labels, e->idx, 1, e->B, 1, s));
  if (e->optimizer != 0) {
    // Adam: the same fused tail (slab sums + update of the whole arena + shadows + packs + state advance)
    DTF_TRY(backward_impl(e, e->ds_images, e->idx, 0, s));
    const void* part[3] = {e->head_part, e->c2_part, e->c1_part};
    int G[3] = {e->head_ks, e->c2_g, e->c1_g}, n[3] = {(int)e->head_stride, (int)e->c2_stride, (int)e->c1_stride};
    long st[3] = {e->head_stride, e->c2_stride, e->c1_stride};
    long off[3] = {e->off_out_w, e->off_wc2, e->off_wc1};
    return dtf_fused_reduce_adam(3, part, G, n, st, off, e->params, e->grads, e->adam_m, e->adam_v, shadow_of(e), e->hyper,
                                 e->off_wd1, e->off_out_w, e->wc2T, e->off_wc2, e->c1, e->c2, e->out_wT, e->off_out_w,
                                 e->hidden, e->state, 3, e->B, step_inc, 1, s);
  }
  if (e->fuse_dense_update) return train_tail_fused_dense(e, step_inc, s);
  DTF_TRY(backward_impl(e, e->ds_images, e->idx, 0, s));
  const void* part[3] = {e->head_part, e->c2_part, e->c1_part};
  int G[3] = {e->head_ks, e->c2_g, e->c1_g}, n[3] = {(int)e->head_stride, (int)e->c2_stride, (int)e->c1_stride};
  long st[3] = {e->head_stride, e->c2_stride, e->c1_stride};
  long off[3] = {e->off_out_w, e->off_wc2, e->off_wc1};
  return dtf_fused_reduce_sgd(3, part, G, n, st, off, e->params, e->grads, shadow_of(e), e->hyper, e->off_wd1, e->off_out_w,
                              e->wc2T, e->off_wc2, e->c1, e->c2, e->out_wT, e->off_out_w, e->hidden, e->state, e->B, step_inc, 1, s);
}

// ---------------------------------------------------------------------------------------------
// Data-parallel step choreography (parallel/sync_dp.py).  Arena = [wd1 | head | conv2 | conv1]: the dense
// collective is wd1 alone (12.8 MB), the tail collective everything else (0.3 MB).  The step is three native calls around
// the two collectives that Python issues through torch.distributed (RCCL): every kernel launch,
// stream wait and event record of the overlap schedule happens here, so the host cost of a step is
// 3 foreign calls + 2 all-reduce enqueues.  s = compute stream, comm = dense all-reduce + dense
// update stream, tail = conv-tail all-reduce stream (may equal comm).
//   phase1: sampler+conv forward | wait(dense params of the previous step) | dense forward, head,
//           fc1 dgrad (+ head wgrad), fc1 wgrad -> ev_dense_ready ; comm waits for it.
//           (dgrad BEFORE wgrad: it is the last reader of the old bf16 wd1, so the one event that releases the
//           gradient to the comm stream also orders the dense update behind it; every event recorded on the compute
//           stream costs a hole in it, and this order needs one per step instead of two.)
//                                                                       [Python: all-reduce dense on comm]
//   phase2: conv backward ; comm: update dense range [dense_lo, dense_hi) -> ev_dense_updated.
//           Sharded dense update (sync_dp.py shard_dense): Python has reduce-scattered the gradient, [dense_lo,
//           dense_hi) is this rank's shard, record_updated == 0, and Python all-gathers the updated weights on comm
//           before it records ev_dense_updated itself.
//                                                   [Python: all-reduce tail, on s itself by default]
//   phase3: updates the tail (+ state advance), repacks wc2.  (With a separate tail stream: two event pairs.)
// ---------------------------------------------------------------------------------------------
int dtf_engine_dp_phase1(const DtfEngineDesc* e, int wait_dense_updated, hipStream_t s, hipStream_t comm) {
  DTF_TRY(forward_conv_impl(e, e->ds_images, nullptr, 1, e->B, s));
  if (wait_dense_updated) HIP_TRY(hipStreamWaitEvent(s, e->ev_dense_updated, 0));
  DTF_TRY(dtf_engine_forward_dense(e, e->ds_labels, e->idx, 1, e->B, 1, s));
  DTF_TRY(fc1_dgrad(e, 1, s));              // + the head wgrad as extra workgroups of the same launch
  if (DTF_EXP_DP & 32) {        // no hand-over at all (the dense update then races with its producer: timing only)
    return backward_dense_impl(e, -1, s);
  } else if (DTF_EXP_DP & 16) {
    DTF_TRY(backward_dense_impl(e, -1, s));
    HIP_TRY(hipEventRecord(e->ev_dense_ready, s));
  } else {
    // fc1 wgrad only; the event rides on the kernel
    DTF_TRY(backward_dense_impl(e, -1, s, e->ev_dense_ready, e->dense_g16 != nullptr && !is_f32(e)));
  }
  return (int)hipStreamWaitEvent(comm, e->ev_dense_ready, 0);
}

int dtf_engine_dp_phase2(const DtfEngineDesc* e, long dense_lo, long dense_hi, int dense_adam_slot, int reduce_tail,
                         const void* dense_grad_src, int record_updated, hipStream_t s, hipStream_t comm,
                         hipStream_t tail) {
  DTF_TRY(backward_conv_impl(e, e->ds_images, e->idx, 0, reduce_tail ? 2 : 0, -1, s));
  // (the dense update overwrites the bf16 wd1 that fc1 dgrad reads: that kernel ran before ev_dense_ready)
  if (!(DTF_EXP_DP & 2)) {
    const int g16 = e->dense_g16 != nullptr && !is_f32(e);
    if (g16 && dense_grad_src) return -3;     // the 16-bit gradient lives in e->dense_g16, not in a peer-to-peer staging area
    const void* gsrc = g16 ? (const void*)bf16at(e->dense_g16, dense_lo - e->off_wd1) : dense_grad_src;
    DTF_TRY(update_range_src(e, dense_lo, dense_hi, 0, 0, 0, dense_adam_slot, gsrc, (DTF_EXP_DP & 4) ? s : comm, g16));
  }
  if (record_updated) HIP_TRY(hipEventRecord(e->ev_dense_updated, comm));
  // tail == s: the tail collective is issued on the compute stream itself (everything after it depends on
  // it anyway), which needs no events at all; a separate tail stream is ordered behind the conv backward
  if (tail != s) {
    HIP_TRY(hipEventRecord(e->ev_tail_ready, s));
    HIP_TRY(hipStreamWaitEvent(tail, e->ev_tail_ready, 0));
  }
  return 0;
}

int dtf_engine_dp_phase3(const DtfEngineDesc* e, long dense_hi, int tail_adam_slot, long step_inc, int wait_dense,
                         hipStream_t s, hipStream_t tail) {
  if (tail != s) {
    HIP_TRY(hipEventRecord(e->ev_tail_reduced, tail));
    HIP_TRY(hipStreamWaitEvent(s, e->ev_tail_reduced, 0));
  }
  if (dense_hi == e->off_out_w) {
    // update + bf16 shadow + wc2 repack + out_w transpose of the (already all-reduced) tail
    // (head | conv2 | conv1) in one launch -- the fused tail kernel with the gradient arena itself as
    // a single "slab" per segment
    const void* part[3] = {f32at(e->grads, e->off_out_w), f32at(e->grads, e->off_wc2), f32at(e->grads, e->off_wc1)};
    int G[3] = {1, 1, 1}, n[3] = {(int)e->head_stride, (int)e->c2_stride, (int)e->c1_stride};
    long st[3] = {e->head_stride, e->c2_stride, e->c1_stride};
    long off[3] = {e->off_out_w, e->off_wc2, e->off_wc1};
    if (e->optimizer == 0)
      DTF_TRY(dtf_fused_reduce_sgd(3, part, G, n, st, off, e->params, e->grads, shadow_of(e), e->hyper, e->off_wc2,
                                   e->off_wc2, e->wc2T, e->off_wc2, e->c1, e->c2, e->out_wT, e->off_out_w, e->hidden,
                                   e->state, e->B, step_inc, 1, s));
    else
      DTF_TRY(dtf_fused_reduce_adam(3, part, G, n, st, off, e->params, e->grads, e->adam_m, e->adam_v, shadow_of(e),
                                    e->hyper, e->off_wc2, e->off_wc2, e->wc2T, e->off_wc2, e->c1, e->c2, e->out_wT,
                                    e->off_out_w, e->hidden, e->state, tail_adam_slot, e->B, step_inc, 1, s));
  } else {
    // update_range refreshes out_wT on whichever stream updates out_w (here, or the dense range on comm)
    DTF_TRY(dtf_engine_update_range(e, dense_hi, e->n_total, 1, 1, step_inc, tail_adam_slot, s));
    DTF_TRY(dtf_engine_transposes(e, 1, s));
  }
  if (wait_dense) HIP_TRY(hipStreamWaitEvent(s, e->ev_dense_updated, 0));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// The whole data-parallel step through the exchange fabric (csrc/kernels/dp.hip): both messages as peer-to-peer
// stores, every per-step counter in device memory.  Enqueue-only and free of per-step arguments, so Python captures it
// ONCE into a HIP graph and a step costs the host one graph launch (parallel/sync_dp.py).
//   s:     conv fwd, fc1 fwd, head, fc1 dgrad (+ head wgrad), fc1 wgrad -> gradient arena
//   comm:    dense exchange (reduce-scatter -> shard sum -> optimizer arithmetic -> all-gather of the compute copy)
//   s:     conv2 wgrad, conv2 dgrad + conv1 wgrad, slab reduce (+ control flags)        | comm joins here
//   s:     tail exchange + verdict + commit: tail update, step state, this rank's shard of the dense master
// collapse_world1: a one-rank job has nobody to exchange the dense gradient with, and SGD then applies it in the
// epilogue of the wgrad GEMM exactly as the single-GPU step does (bitwise the same kernel); the tail still goes through
// the exchange kernel (slot of its own, flags, verdict), which is what a one-GPU box can measure of the protocol.
// ---------------------------------------------------------------------------------------------
int dtf_engine_dp_step(const DtfEngineDesc* e, const DtfDpFabric* f, int collapse_world1, long step_inc, hipStream_t s,
                       hipStream_t comm) {
  if (!f || f->n_dense != (long)e->flat * e->hidden || f->n_tail != e->n_total - e->off_out_w || e->off_wd1 != 0 ||
      f->ag_bytes != (is_f32(e) ? 4 : 2) || e->dense_g16)
    return -3;
  const int adam = e->optimizer != 0;
  DTF_TRY(forward_conv_impl(e, e->ds_images, nullptr, 1, e->B, s));
  DTF_TRY(dtf_engine_forward_dense(e, e->ds_labels, e->idx, 1, e->B, 1, s));
  DTF_TRY(fc1_dgrad(e, 1, s));              // the last reader of the old fc1 weights; + the head wgrad workgroups
  // a one-rank job with collapse_world1: SGD applies the dense update in the wgrad epilogue; any other optimizer lets
  // the commit workgroups of the tail kernel do it straight from the gradient arena (with_dense == 2).  Either way
  // there is no dense exchange kernel, no branch and no event in the step.
  // (bit 1 of collapse_world1: every rank has a GPU to itself -- see backward_conv_impl's dgrad_first)
  const bool own_gpu = (collapse_world1 & 2) != 0;
  collapse_world1 &= 1;
  const bool collapsed = f->world == 1 && collapse_world1 && !adam;
  const bool local_dense = f->world == 1 && collapse_world1 && adam;
  if (collapsed) {
    if (is_f32(e))
      DTF_TRY(dtf_gemm_f32_tn_sk(e->p2, e->da, f32at(e->params, e->off_wd1), e->flat, e->hidden, e->B, e->flat, e->hidden,
                                 e->hidden, e->hyper, e->sk_ws, nullptr, s));
    else
      DTF_TRY(dtf_gemm_tn_sgd(e->p2, e->da, f32at(e->params, e->off_wd1), bf16at(e->shadow, e->off_wd1), e->hyper, e->flat,
                              e->hidden, e->B, e->flat, e->hidden, e->hidden, s));
  } else if (local_dense) {
    DTF_TRY(backward_dense_impl(e, -1, s));
  } else {
    // the hand-over event rides on the wgrad kernel itself where that is possible (eager launches: a recorded event
    // is a marker packet, a 5-7 us hole in the compute stream); under graph capture it is a plain dependency edge
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    const bool bind = comm != s && cap == hipStreamCaptureStatusNone;
    DTF_TRY(backward_dense_impl(e, -1, s, bind ? e->ev_dense_ready : nullptr));
    if (comm != s) {
      if (!bind) HIP_TRY(hipEventRecord(e->ev_dense_ready, s));
      HIP_TRY(hipStreamWaitEvent(comm, e->ev_dense_ready, 0));
    }
    DTF_TRY(dtf_dp_dense_exchange(f, adam, f32at(e->grads, e->off_wd1), f32at(e->params, e->off_wd1),
                                  adam ? f32at(e->adam_m, e->off_wd1) : nullptr, adam ? f32at(e->adam_v, e->off_wd1) : nullptr,
                                  is_f32(e) ? nullptr : (void*)bf16at(e->shadow, e->off_wd1), e->hyper,
                                  adam ? (const void*)((const int64_t*)e->state + 3) : nullptr, comm));
    if (comm != s) HIP_TRY(hipEventRecord(e->ev_dense_updated, comm));
  }
  DTF_TRY(backward_conv_impl(e, e->ds_images, e->idx, 0, 2, -1, s, own_gpu && !collapsed && !local_dense && comm != s));
  if (!collapsed && !local_dense && comm != s) HIP_TRY(hipStreamWaitEvent(s, e->ev_dense_updated, 0));
  return dtf_dp_tail_commit(f, adam, collapsed ? 0 : (local_dense ? 2 : 1), e->params, e->grads, e->adam_m, e->adam_v, shadow_of(e), e->hyper,
                            e->wc2T, e->off_wc2, e->c1, e->c2, e->out_wT, e->off_out_w, e->hidden, e->state, 3, 4, e->B,
                            step_inc, 1, e->off_out_w, e->off_wd1, s);
}

int dtf_engine_dp_record_updated(const DtfEngineDesc* e, hipStream_t comm) {
  return (int)hipEventRecord(e->ev_dense_updated, comm);
}

// make the deferred dense update of the last step visible to s
int dtf_engine_dp_flush(const DtfEngineDesc* e, hipStream_t s) {
  return (int)hipStreamWaitEvent(s, e->ev_dense_updated, 0);
}

}  // extern "C"
