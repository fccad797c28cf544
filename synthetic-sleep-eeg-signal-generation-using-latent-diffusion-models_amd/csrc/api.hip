// Library plumbing: thread-local error string, context, HIP-event timer.
#include "common.h"
#include <algorithm>
#include <stdlib.h>

static thread_local std::string g_err;
void eegldm_set_error(const std::string& msg) { g_err = msg; }

struct TimerState { hipEvent_t a = nullptr, b = nullptr; };
static thread_local TimerState g_timer;

std::atomic<int> g_eeg_env_epoch{0};
std::atomic<int> g_eeg_live_ctx{0};
extern "C" int eegldm_abi_version(void) { return EEGLDM_ABI_VERSION; }
int eeg_det_buffer(eegldm_ctx* ctx, size_t bytes, float** out) {
  if (ctx->det_buf_bytes < bytes) {
    if (ctx->det_buf) { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(ctx->det_buf)); ctx->det_buf = nullptr; ctx->det_buf_bytes = 0; }
    const size_t want = bytes < ((size_t)32 << 20) ? ((size_t)32 << 20) : bytes;
    HIP_TRY(hipMalloc(&ctx->det_buf, want)); ctx->det_buf_bytes = want;
  }
  *out = ctx->det_buf;
  return 0;
}
extern "C" int eegldm_debug_reload_env(void) { return ++g_eeg_env_epoch; }

extern "C" const char* eegldm_last_error(void) { return g_err.c_str(); }

extern "C" int eegldm_ctx_create(int device, void* stream, int own_stream, eegldm_ctx** out) {
  EEG_CHECK(out != nullptr, "out is null");
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    EEG_FAIL(EEGLDM_ERR_UNSUPPORTED, "libeegldm is built for gfx950 only; device %d is %s", device, prop.gcnArchName);
  eegldm_ctx* c = new eegldm_ctx();
  c->device = device;
  c->num_cu = prop.multiProcessorCount;
  if (own_stream) { HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->owns_stream = true; }
  else { c->stream = (hipStream_t)stream; c->owns_stream = false; }
  c->scratch_bytes = 25u << 20;    // [0, 8 MiB): reduction accumulators (see norm.hip / losses.hip / net.hip); [8, 24 MiB): column-sum partials; [24, 25): GroupNorm slot areas C0 / C1
  HIP_TRY(hipMalloc(&c->scratch, c->scratch_bytes));
  HIP_TRY(hipMemset(c->scratch, 0, c->scratch_bytes));   // reduction scratch is self-cleaning: kernels re-zero what they consume
#ifdef EEG_STAGE_TIMING
  const size_t zp_bytes = 64u << 20;     // debug build: the area behind the zero page receives per-block stage timestamps
#else
  const size_t zp_bytes = 4096;
#endif
  // Second stream for the per-layer weight gradients of res_backward / attn_backward (the round-1/2 flow).  OPT-IN since round 3
  // (EEGLDM_SIDE_STREAM=1; EEGLDM_NO_SIDE_STREAM is still accepted and wins): the UNet backward groups its weight gradients and leaves
  // nothing for it, and the fused 3-tap weight-gradient kernel running BESIDE a narrow-block GroupNorm backward made that kernel's
  // group sums come out wrong (reproduced in isolation, tools/debug/gn_conc2.py / gn_conc4.py; cause not found) --
  // no default path of the library runs two of its own kernels concurrently any more.
  if (getenv("EEGLDM_SIDE_STREAM") && atoi(getenv("EEGLDM_SIDE_STREAM")) != 0 && !getenv("EEGLDM_NO_SIDE_STREAM")) {
    HIP_TRY(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    c->side_on = true;
  }
  HIP_TRY(hipMalloc(&c->zero_page, zp_bytes));
  HIP_TRY(hipMemset(c->zero_page, 0, zp_bytes));
  g_eeg_live_ctx++;
  *out = c;
  return 0;
}
// the HIP stream every call on this context enqueues on (the caller's, or the library's own when own_stream was set): what a caller
// that issues its own collectives / copies must order them against
extern "C" void* eegldm_ctx_stream(const eegldm_ctx* c) { return c ? (void*)c->stream : nullptr; }
int ctx_fork(eegldm_ctx* c) {
  if (!c->side_on || c->prof_on || c->defer_wgrad) return 0;      // grouped weight gradients: nothing is left for the side stream
  HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
  HIP_TRY(hipStreamWaitEvent(c->side, c->ev_fork, 0));
  return 0;
}
int ctx_join(eegldm_ctx* c) {
  if (!c->side_on || c->prof_on || c->defer_wgrad) return 0;
  HIP_TRY(hipEventRecord(c->ev_join, c->side));
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
  return 0;
}
extern "C" int eegldm_ctx_destroy(eegldm_ctx* c) {
  if (!c) return 0;
  hipSetDevice(c->device);
  if (c->scratch) hipFree(c->scratch);
  if (c->zero_page) hipFree(c->zero_page);
  if (c->splitk_ws) hipFree(c->splitk_ws);
  if (c->grp_dev) hipFree(c->grp_dev);
  if (c->gn_slot_arena) hipFree(c->gn_slot_arena);
  if (c->gn_fold_dev) hipFree(c->gn_fold_dev);
  if (c->det_buf) hipFree(c->det_buf);
  if (c->owns_stream) hipStreamDestroy(c->stream);
  if (c->side) { hipStreamDestroy(c->side); hipEventDestroy(c->ev_fork); hipEventDestroy(c->ev_join); }
  delete c;
  if (g_eeg_live_ctx.load() > 0) g_eeg_live_ctx--;
  return 0;
}
extern "C" int eegldm_ctx_sync(eegldm_ctx* c) {
  EEG_CHECK(c, "null ctx");
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int eegldm_timer_start(eegldm_ctx* c) {
  EEG_CHECK(c, "null ctx");
  if (!g_timer.a) { HIP_TRY(hipEventCreate(&g_timer.a)); HIP_TRY(hipEventCreate(&g_timer.b)); }
  HIP_TRY(hipEventRecord(g_timer.a, c->stream));
  return 0;
}
extern "C" int eegldm_timer_stop_ms(eegldm_ctx* c, float* ms) {
  EEG_CHECK(c && ms && g_timer.a, "timer not started");
  HIP_TRY(hipEventRecord(g_timer.b, c->stream));
  HIP_TRY(hipEventSynchronize(g_timer.b));
  HIP_TRY(hipEventElapsedTime(ms, g_timer.a, g_timer.b));
  return 0;
}

// ------------------------------------------------------------------ per-launch GEMM profiling (HIP events on the ctx stream)
extern "C" int eegldm_prof_enable(eegldm_ctx* c, int on) {
  EEG_CHECK(c, "null ctx");
  for (auto& r : c->prof) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  c->prof.clear();
  c->prof_on = on != 0;
  if (c->prof_on) {
    // An event pair around a launch also times the event records themselves (the kernel trace of the same run is 3-5 us per
    // launch shorter).  Calibrate: median elapsed time of 33 empty pairs on this stream, subtracted per launch in the summary.
    HIP_TRY(hipStreamSynchronize(c->stream));
    constexpr int NCAL = 33;
    hipEvent_t ea[NCAL], eb[NCAL];
    for (int i = 0; i < NCAL; i++) { HIP_TRY(hipEventCreate(&ea[i])); HIP_TRY(hipEventCreate(&eb[i])); }
    for (int i = 0; i < NCAL; i++) { HIP_TRY(hipEventRecord(ea[i], c->stream)); HIP_TRY(hipEventRecord(eb[i], c->stream)); }
    HIP_TRY(hipStreamSynchronize(c->stream));
    float el[NCAL];
    for (int i = 0; i < NCAL; i++) { el[i] = 0.f; hipEventElapsedTime(&el[i], ea[i], eb[i]); hipEventDestroy(ea[i]); hipEventDestroy(eb[i]); }
    std::sort(el, el + NCAL);
    c->prof_bracket_ms = el[NCAL / 2] > 0.f ? el[NCAL / 2] : 0.0;
  }
  return 0;
}
extern "C" int eegldm_prof_bracket_overhead_ms(eegldm_ctx* c, double* ms) {
  EEG_CHECK(c && ms, "null argument");
  *ms = c->prof_bracket_ms;
  return 0;
}
extern "C" int eegldm_prof_summary(eegldm_ctx* c, int cls, double* flops, double* ms, int* launches) {
  EEG_CHECK(c && flops && ms && launches, "null argument");
  EEG_CHECK(cls >= 0 && cls < PROF_NCLASS, "class %d out of range", cls);
  HIP_TRY(hipStreamSynchronize(c->stream));
  double f = 0, t = 0; int n = 0;
  for (auto& r : c->prof) {
    if (r.cls != cls) continue;
    float e = 0; HIP_TRY(hipEventElapsedTime(&e, r.a, r.b));
    const double own = (double)e - c->prof_bracket_ms;        // minus the empty-bracket time (see eegldm_prof_enable)
    f += r.flops; t += own > 0.0 ? own : 0.0; n++;
  }
  *flops = f; *ms = t; *launches = n;
  return 0;
}

// developer aid: one CSV line per profiled launch (class,M,N,K,taps,splitk,ms,gflop)
extern "C" int eegldm_prof_dump(eegldm_ctx* c, const char* path_host) {
  EEG_CHECK(c && path_host, "null argument");
  HIP_TRY(hipStreamSynchronize(c->stream));
  FILE* f = fopen(path_host, "w");
  EEG_CHECK(f != nullptr, "cannot open %s", path_host);
  fprintf(f, "class,M,N,K,taps,splitk,ms,gflop\n");
  for (auto& r : c->prof) {
    float e = 0; hipEventElapsedTime(&e, r.a, r.b);
    fprintf(f, "%d,%d,%d,%d,%d,%d,%.5f,%.3f\n", r.cls, r.M, r.N, r.K, r.taps, r.splitk, e, r.flops / 1e9);
  }
  fclose(f);
  return 0;
}

#ifdef EEG_STAGE_TIMING
extern "C" int eegldm_debug_read_tlog(eegldm_ctx* c, unsigned long long* dst_host, long n_words) {
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(dst_host, (char*)c->zero_page + 4096, n_words * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset((char*)c->zero_page + 4096, 0, n_words * 8));
  return 0;
}
#endif
