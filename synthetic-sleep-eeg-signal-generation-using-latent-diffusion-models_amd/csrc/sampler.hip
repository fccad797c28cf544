// DDIM / DDPM sampling as ONE native call: the loop of /root/reference/src/sample_trials.py:149-170
// (noise -> [UNet, scheduler.step] x N -> decode(z / scale_factor)) and of sample_trials_ddpm.py:99-104 /
// util.py:261-285 (pixel-space model, 1000-step ancestral sampler).  Host code only.
//
// The reference samples ONE window per call (sample_trials.py:149-163).  At batch 1 a UNet forward is ~130 dependent launches whose
// own latency (not the host's launch rate: the rocprofv3 trace shows them back to back) sets the time per step, so the work went
// into the kernels of that chain (conv_skinny.hip, few-slab GroupNorm) and into taking launches out of the step: the
// embedding rows of all timesteps are computed once per run (below).  The forward CAN be replayed from a hipGraph captured once per
// (B, L) (activation arena, timestep buffer and latent buffer have fixed addresses) -- opt-in, because ROCm's graph launch measured
// slower than the eager launches at B = 1 and equal at B = 256.
#include <map>
#include <mutex>
#include <tuple>
#include <utility>
#include <vector>

#include "net.h"

namespace {
struct GraphKey { const eegldm_unet* u; int B, L; bool operator<(const GraphKey& o) const { return std::tie(u, B, L) < std::tie(o.u, o.B, o.L); } };
struct SamplerState {
  hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr;
  float *x = nullptr, *out = nullptr, *nz = nullptr; int64_t* tt = nullptr;
  hipStream_t stream = nullptr; hipEvent_t ev_in = nullptr, ev_out = nullptr;
  bool capture_failed = false;
  // embedding rows of all timesteps of a run (eager path): table [emb_cap][etot], scratch of the embedding MLP, timesteps on the device
  float *emb_table = nullptr, *emb_work = nullptr; int64_t* steps_dev = nullptr; int emb_cap = 0;
  float* emb_row = nullptr;      // graph path: the ONE row the captured forward reads; the step's table row is copied here before every replay
  bool exec_reads_row = false;   // how `exec` was captured: reading emb_row (table mode) or recomputing the embedding from s.tt -- a replay in the other mode would use a stale timestep
};
std::map<GraphKey, SamplerState>& states() { static std::map<GraphKey, SamplerState> m; return m; }
// guards the map AND serialises eegldm_sample: a call swaps ctx->stream for its duration, so two concurrent calls on contexts that
// share a UNet -- or any other call on the same context from a second thread -- are not supported (one context = one thread,
// include/eegldm.h); the lock at least keeps the graph cache consistent when independent contexts sample from different threads
std::recursive_mutex& states_mutex() { static std::recursive_mutex m; return m; }

__global__ void fill_i64_kernel(int64_t* p, int n, int64_t v) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
}  // namespace

void sampler_release(const eegldm_unet* u) {
  std::lock_guard<std::recursive_mutex> lock(states_mutex());
  auto& m = states();
  for (auto it = m.begin(); it != m.end();) {
    if (it->first.u != u) { ++it; continue; }
    SamplerState& s = it->second;
    if (s.exec) (void)hipGraphExecDestroy(s.exec);
    if (s.graph) (void)hipGraphDestroy(s.graph);
    if (s.x) (void)hipFree(s.x);
    if (s.out) (void)hipFree(s.out);
    if (s.nz) (void)hipFree(s.nz);
    if (s.tt) (void)hipFree(s.tt);
    if (s.emb_table) (void)hipFree(s.emb_table);
    if (s.emb_work) (void)hipFree(s.emb_work);
    if (s.steps_dev) (void)hipFree(s.steps_dev);
    if (s.emb_row) (void)hipFree(s.emb_row);
    if (s.ev_in) (void)hipEventDestroy(s.ev_in);
    if (s.ev_out) (void)hipEventDestroy(s.ev_out);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    it = m.erase(it);
  }
}

extern "C" int eegldm_sample(eegldm_unet* u, eegldm_aekl* ae, const float* noise, const int64_t* timesteps_host, const float* a_t_host,
                             const float* a_prev_host, const float* beta_t_host, int n_steps, int ancestral, int pred_type, int clip_sample,
                             float inv_scale_factor, uint64_t noise_seed, float* latents_out, float* windows_out, int B, int L, int use_graph,
                             int* graph_used_host) {
  EEG_CHECK(u && noise && timesteps_host && a_t_host && a_prev_host, "null argument");
  EEG_CHECK(!ancestral || beta_t_host, "the ancestral (DDPM) step needs beta_t");
  EEG_CHECK(n_steps >= 1 && B >= 1 && L >= 1, "bad sizes");
  EEG_CHECK(latents_out || windows_out, "nothing to return: pass latents_out and/or windows_out");
  eegldm_ctx* ctx = unet_ctx(u);
  const int C = unet_in_channels(u);
  EEG_CHECK(unet_out_channels(u) == C, "sampling needs in_channels == out_channels");
  EEG_CHECK(!ae || aekl_ctx(ae) == ctx, "the autoencoder and the UNet must share one context");
  const long n = (long)B * C * L;
  std::lock_guard<std::recursive_mutex> lock(states_mutex());
  SamplerState& s = states()[GraphKey{u, B, L}];
  if (!s.x) {
    HIP_TRY(hipMalloc(&s.x, sizeof(float) * n)); HIP_TRY(hipMalloc(&s.out, sizeof(float) * n)); HIP_TRY(hipMalloc(&s.nz, sizeof(float) * n));
    HIP_TRY(hipMalloc(&s.tt, sizeof(int64_t) * B));
    HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming));
  }
  // the whole loop runs on the sampler's own stream (a capture cannot start on the NULL stream the caller may have given the context):
  // it waits for the caller's stream first and the caller's stream waits for it at the end
  hipStream_t caller = ctx->stream;
  struct Restore { eegldm_ctx* c; hipStream_t s; ~Restore() { c->stream = s; } } restore{ctx, caller};
  // Eager launches (the default) stay on the caller's stream: a second stream is a second hardware queue, and alternating queues
  // measured +8 % on the one-window chain (47.5 vs 51.8 ms, the same as GPU_MAX_HW_QUEUES=1 gives with the own stream).
  EEG_ENV_VAR(bool, force_own, getenv("EEGLDM_SAMPLE_OWN_STREAM") != nullptr);
  const bool own = use_graph || force_own;
  const hipStream_t run = own ? s.stream : caller;
  if (own) {
    HIP_TRY(hipEventRecord(s.ev_in, caller));
    HIP_TRY(hipStreamWaitEvent(s.stream, s.ev_in, 0));
  }
  ctx->stream = run;
  HIP_TRY(hipMemcpyAsync(s.x, noise, sizeof(float) * n, hipMemcpyDeviceToDevice, run));

  auto set_t = [&](int64_t t) { hipLaunchKernelGGL(fill_i64_kernel, dim3((B + 255) / 256), dim3(256), 0, run, s.tt, B, t); };
  bool graph_ok = false;
  const int etot = unet_emb_width(u);
  struct ClearEmb { eegldm_unet* u; ~ClearEmb() { unet_set_shared_emb(u, nullptr); } } clear_emb{u};
  EEG_ENV_VAR(bool, no_table, getenv("EEGLDM_SAMPLE_NO_EMB_TABLE") != nullptr);
  if (use_graph && !ctx->prof_on && !s.capture_failed) {
    // Round 5: the captured forward reads its embedding projections from ONE fixed row (s.emb_row) that the loop below refills from the
    // table of all timesteps before every replay -- until round 4 the graph path recomputed the embedding MLP and the 21 projections
    // inside every replay (six launches, ~100 us at B = 1: 5 of the 10.6 ms by which the replayed DDIM-50 trailed the eager one).
    if (!no_table) {
      if (!s.emb_row) HIP_TRY(hipMalloc(&s.emb_row, sizeof(float) * (size_t)etot));
      unet_set_shared_emb(u, s.emb_row);
    }
    if (s.exec && s.exec_reads_row != !no_table) {      // captured in the other embedding mode (the switch was flipped in-process): capture again
      (void)hipGraphExecDestroy(s.exec); s.exec = nullptr;
      if (s.graph) { (void)hipGraphDestroy(s.graph); s.graph = nullptr; }
    }
    if (!s.exec) {
      // eager warm-up: grows the arena / workspaces (hipMalloc is not capturable), then capture the identical launch sequence
      set_t(timesteps_host[0]);
      EEG_TRY(eegldm_unet_forward(u, s.x, s.tt, s.out, B, L, 0));
      HIP_TRY(hipStreamSynchronize(run));
      int rc = 0;
      if (hipStreamBeginCapture(run, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        rc = eegldm_unet_forward(u, s.x, s.tt, s.out, B, L, 0);
        hipError_t e = hipStreamEndCapture(run, &s.graph);
        if (rc == 0 && e == hipSuccess && s.graph && hipGraphInstantiate(&s.exec, s.graph, nullptr, nullptr, 0) == hipSuccess) { graph_ok = true; s.exec_reads_row = !no_table; }
      }
      if (!graph_ok) {
        (void)hipGetLastError();
        if (s.graph) { (void)hipGraphDestroy(s.graph); s.graph = nullptr; }
        s.exec = nullptr; s.capture_failed = true;
        if (rc) return rc;
      }
    } else graph_ok = true;
  }
  if (graph_used_host) *graph_used_host = graph_ok ? 1 : 0;

  // Eager path: the timesteps are known up front and shared by all samples, so the timestep-embedding MLP and the ResBlocks' embedding
  // projections run ONCE for all n_steps (one batch of n_steps rows) instead of six launches (~100 us at B = 1) inside every step;
  // each forward then reads its step's row with row stride 0.  EEGLDM_SAMPLE_NO_EMB_TABLE=1 restores the per-step computation.
  const bool table = !no_table;
  if (!graph_ok) unet_set_shared_emb(u, nullptr);      // (a failed capture falls back to the eager path below)
  if (table) {
    if (s.emb_cap < n_steps) {
      HIP_TRY(hipStreamSynchronize(run));
      if (s.emb_table) (void)hipFree(s.emb_table);
      if (s.emb_work) (void)hipFree(s.emb_work);
      if (s.steps_dev) (void)hipFree(s.steps_dev);
      s.emb_table = nullptr; s.emb_work = nullptr; s.steps_dev = nullptr; s.emb_cap = 0;
      HIP_TRY(hipMalloc(&s.emb_table, sizeof(float) * (size_t)n_steps * etot));
      HIP_TRY(hipMalloc(&s.emb_work, sizeof(float) * (size_t)n_steps * unet_embed_work_floats(u)));
      HIP_TRY(hipMalloc(&s.steps_dev, sizeof(int64_t) * n_steps));
      s.emb_cap = n_steps;
    }
    HIP_TRY(hipMemcpyAsync(s.steps_dev, timesteps_host, sizeof(int64_t) * n_steps, hipMemcpyHostToDevice, run));
    EEG_TRY(unet_embed_table(u, s.steps_dev, n_steps, s.emb_table, s.emb_work));
    set_t(timesteps_host[0]);                       // s.tt is not read on this path; keep it defined
  }

  for (int i = 0; i < n_steps; i++) {
    if (table && graph_ok) HIP_TRY(hipMemcpyAsync(s.emb_row, s.emb_table + (size_t)i * etot, sizeof(float) * (size_t)etot, hipMemcpyDeviceToDevice, run));
    else if (table) unet_set_shared_emb(u, s.emb_table + (size_t)i * etot);
    else set_t(timesteps_host[i]);
    if (graph_ok) HIP_TRY(hipGraphLaunch(s.exec, run));
    else EEG_TRY(eegldm_unet_forward(u, s.x, s.tt, s.out, B, L, 0));
    if (ancestral) {
      const bool last = a_prev_host[i] >= 1.0f;
      if (!last) EEG_TRY(eegldm_randn(ctx, s.nz, n, noise_seed, (uint64_t)i * (uint64_t)((n + 3) / 4)));
      EEG_TRY(eegldm_ddpm_step(ctx, s.out, s.x, last ? nullptr : s.nz, a_t_host[i], a_prev_host[i], beta_t_host[i], pred_type, clip_sample, s.x, nullptr, n));
    } else {
      EEG_TRY(eegldm_ddim_step(ctx, s.out, s.x, a_t_host[i], a_prev_host[i], pred_type, clip_sample, s.x, nullptr, n));
    }
  }
  if (latents_out) HIP_TRY(hipMemcpyAsync(latents_out, s.x, sizeof(float) * n, hipMemcpyDeviceToDevice, run));
  if (windows_out) {
    if (ae) {
      if (inv_scale_factor != 1.0f) EEG_TRY(eegldm_axpy(ctx, s.x, s.x, inv_scale_factor - 1.0f, n));     // z / scale_factor (sample_trials.py:166)
      EEG_TRY(eegldm_aekl_decode(ae, s.x, windows_out, B, L));
    } else {
      HIP_TRY(hipMemcpyAsync(windows_out, s.x, sizeof(float) * n, hipMemcpyDeviceToDevice, run));    // pixel-space model: x IS the window
    }
  }
  if (own) {
    HIP_TRY(hipEventRecord(s.ev_out, run));
    HIP_TRY(hipStreamWaitEvent(caller, s.ev_out, 0));
  }
  return 0;
}
