// ONE switch for every diagnostic compile-time knob of the kernels and the step executor.
//
// The shipped library is built WITHOUT -DDTF_DIAG: every knob below is then forced to its production value, and
// defining any of them without DTF_DIAG is a compile error -- a benchmark can therefore never run on a library with
// an ablated kernel by accident.  Diagnostic builds (tools/conv_ablate.py, tools/conv_timeline.py, A/B experiments)
// pass -DDTF_DIAG=1 together with the knobs they want; such a library reports "diag" from dtf_build_flavour() and
// ops/native.py says so loudly when it loads one.
//   DTF_ABLATE            conv2.hip   bit mask: drop B loads / A reads / MFMAs / staging / stores (timing ablations)
//   DTF_STAGGER(_MODE)    conv2.hip   delay part of the grid at kernel start
//   DTF_TIMING            conv2/gemm/head  in-kernel wall-clock stamps
//   DTF_FWD_CFG / DTF_DGRAD_CFG   conv2.hip   tile configuration sweeps
//   DTF_WGRAD_STAGE_LAST  conv2.hip   old staging order of conv5x5_wgrad
//   DTF_GEMM_MINW         gemm.hip    minimum waves per SIMD the GEMMs are compiled for
//   DTF_EXP_DP            engine.hip  data-parallel schedule experiments that BREAK the step
#pragma once

#ifndef DTF_DIAG
#if defined(DTF_ABLATE) || defined(DTF_STAGGER) || defined(DTF_STAGGER_MODE) || defined(DTF_TIMING) || defined(DTF_FWD_CFG) || \
    defined(DTF_DGRAD_CFG) || defined(DTF_WGRAD_STAGE_LAST) || defined(DTF_GEMM_MINW) || defined(DTF_EXP_DP)
#error "diagnostic knobs (DTF_ABLATE, DTF_TIMING, DTF_EXP_DP, ...) need -DDTF_DIAG=1: see csrc/kernels/diag.hpp"
#endif
#define DTF_DIAG 0
#endif

#ifndef DTF_ABLATE
#define DTF_ABLATE 0
#endif
#ifndef DTF_STAGGER
#define DTF_STAGGER 0
#endif
#ifndef DTF_STAGGER_MODE
#define DTF_STAGGER_MODE 0
#endif
#ifndef DTF_TIMING
#define DTF_TIMING 0
#endif
#ifndef DTF_WGRAD_STAGE_LAST
#define DTF_WGRAD_STAGE_LAST 0
#endif
#ifndef DTF_GEMM_MINW
#define DTF_GEMM_MINW 1
#endif
#ifndef DTF_EXP_DP
#define DTF_EXP_DP 0   // 2 = no dense update, 4 = dense update on the compute stream, 8 = events with the default (system-scope) release, 16 = ev_dense_ready as a marker behind the fc1 wgrad, 32 = no ev_dense_ready at all
#endif
