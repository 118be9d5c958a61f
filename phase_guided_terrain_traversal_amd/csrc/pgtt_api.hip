// pgtt_api.hip — host side of libpgtt.so: the C ABI declared in include/pgtt.h.
// There is NO CPU fallback: every entry point that computes needs a HIP device and fails with
// PGTT_E_NODEVICE / PGTT_E_HIP otherwise.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "pgtt_kernels.hip.h"

// the physics_kernel instantiations (3 lane layouts x step/forward x DR x terrain) live in their own translation units
// (pgtt_physics_inst.hip, compiled in parallel); this file only sees their host launchers
// weak: a side build may compile a subset of the variants (csrc/Makefile `fastdiv`); launching a missing one is a PGTT_E_STATE
#define PG_DECL(S, M, D, T) __attribute__((weak)) void pgtt_launch_physics_s##S##_##M##_##D##_##T(int nblocks, hipStream_t st, const pgtt::KArgs& a, const float* action);
#define PG_DECL8(S) PG_DECL(S, 0, 0, 0) PG_DECL(S, 0, 0, 1) PG_DECL(S, 0, 1, 0) PG_DECL(S, 0, 1, 1) PG_DECL(S, 1, 0, 0) PG_DECL(S, 1, 0, 1) PG_DECL(S, 1, 1, 0) PG_DECL(S, 1, 1, 1)
PG_DECL8(1) PG_DECL8(2) PG_DECL8(4)
#undef PG_DECL8
#undef PG_DECL

namespace {

thread_local std::string g_err;


int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return fail(PGTT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

}  // namespace

struct pgtt_env {
  int device = 0;
  int N = 0;
  PgttConfig cfg{};
  PgttModel model{};
  PgttConfig* d_cfg = nullptr;
  PgttModel* d_model = nullptr;
  pgtt::TerrainBox* d_terrain = nullptr;
  float4* d_cull = nullptr;       // [T][B] (centre x, y, world-AABB half extents x, y): what the scan's cull reads, 16 of a record's 80 bytes, contiguous
  uint4* d_grid = nullptr; float grid_E = 1.f, grid_inv = 1.f;     // terrain grid of the collision pass (pgtt_physics_quad.hip.h::collide)
  int T = 0, B = 0;
  PgttBuffers buf{};
  bool bound = false;
  unsigned long long seed = 0;
  long long env_off = 0;
  float test_rng_fix = NAN; int test_scan_preset = 0;   // pgtt_set_test_overrides
  float* d_handover = nullptr;    // [N][kHandover]: physics -> observe hand-over of one pgtt_step (pgtt_kernels.hip.h, KArgs)
  int* d_flag = nullptr; int* h_flag = nullptr;      // pgtt_reset's range check of the caller's terrain-variant labels (device word, pinned host word)
  bool labels_unchecked = true;   // set by pgtt_bind / pgtt_set_terrain: the next pgtt_reset of any kind checks the labels; afterwards only whole-batch resets do
  bool timing = false;
  int timing_period = 1, timing_tick = 0; bool timing_now = false;   // time every timing_period-th step (event records cost ~3 us of GPU idle each)
  bool split_observe = false;     // observe = observe_kernel<OBS_STEP_OBS> + task_kernel (PgttConfig.observe_form)
  int layout = 0;                 // lane layout of physics_kernel (PGTT_LAYOUT_*): 1 quad (16 envs per wave), 2 oct (8), 4 hex (4), 0 auto
  // kernel timing: a ring of event quadruples (physics begin / end, observe begin / end), one per step; a slot is
  // read back when it comes up for re-use (its step finished long ago: no stall) or by pgtt_kernel_ms_mean()
  static constexpr int kRing = 64;
  hipEvent_t ev[kRing][4] = {};
  bool ev_used[kRing] = {};
  bool ev_chain[kRing] = {};      // the observe launch of the slot started where its physics launch ended (pgtt_step): event 1 is its start, event 2 is not recorded
  int ev_slot = -1;               // slot of the most recent step
  double sum_phys = 0.0, sum_obs = 0.0; long n_timed = 0;
  bool ev_valid = false;
};

#if defined(PGTT_TRACE) || defined(PGTT_TIME)
static float* g_trace = nullptr;
static int g_trace_launch = 0;      // every physics launch records into its own 65536-float segment (mod 4)
static float* pgtt_trace_buffer() {
  if (!g_trace) { hipMalloc(&g_trace, 262144 * sizeof(float)); hipMemset(g_trace, 0, 262144 * sizeof(float)); }
  return g_trace;
}
extern "C" int pgtt_trace_read(float* host, int n) {
  hipDeviceSynchronize();
  return (int)hipMemcpy(host, pgtt_trace_buffer(), n * sizeof(float), hipMemcpyDeviceToHost);
}
extern "C" void pgtt_trace_clear() { hipDeviceSynchronize(); hipMemset(pgtt_trace_buffer(), 0, 262144 * sizeof(float)); g_trace_launch = 0; }
#endif

namespace {

pgtt::KArgs make_args(pgtt_env* h, const unsigned char* mask, float yaw_override) {
  pgtt::KArgs a;
  a.model = h->d_model; a.cfg = h->d_cfg; a.terrain = h->d_terrain; a.cull = h->d_cull; a.T = h->T; a.B = h->B; a.grid = h->d_grid; a.grid_E = h->grid_E; a.grid_inv = h->grid_inv;
  a.buf = h->buf; a.N = h->N; a.seed = h->seed; a.env_off = h->env_off; a.mask = mask; a.yaw_override = yaw_override; a.write_qpos = 0;
  a.rng_fix = h->test_rng_fix; a.scan_preset = h->test_scan_preset;
  a.handover_w = h->d_handover; a.handover_r = nullptr;
#if defined(PGTT_TRACE) || defined(PGTT_TIME)
  a.trace = pgtt_trace_buffer();
#endif
  return a;
}

// add the kernel times of ring slot r (if it holds a finished or pending step) to the running sums and free the slot
int harvest(pgtt_env* h, int r) {
  if (!h->ev_used[r]) return PGTT_OK;
  float p = 0.f, o = 0.f;
  HIP_TRY(hipEventSynchronize(h->ev[r][3]));
  HIP_TRY(hipEventElapsedTime(&p, h->ev[r][0], h->ev[r][1]));
  HIP_TRY(hipEventElapsedTime(&o, h->ev[r][h->ev_chain[r] ? 1 : 2], h->ev[r][3]));
  h->sum_phys += p; h->sum_obs += o; h->n_timed++; h->ev_used[r] = false;
  return PGTT_OK;
}

// number of labels outside [0, T): the step kernels clamp such a label (no out-of-bounds read), pgtt_reset reports it
__global__ void variant_range_kernel(const int32_t* __restrict__ variant, int N, int T, int* __restrict__ bad) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const bool out = e < N && (variant[e] < 0 || variant[e] >= T);
  const unsigned long long b = __ballot(out);
  if (b != 0ull && (threadIdx.x & 63) == 0) atomicAdd(bad, __popcll(b));
}

template <int MODE>
int launch_physics(pgtt_env* h, const pgtt::KArgs& a_in, const float* action, hipStream_t st) {
  pgtt::KArgs a = a_in;
#if defined(PGTT_TRACE) || defined(PGTT_TIME)
  a.trace = pgtt_trace_buffer() + 65536 * (g_trace_launch++ & 3);
#endif
  // quad layout: 16 envs per 64-thread block; hex layout: 4 envs per block (pgtt_physics_quad.hip.h)
  const bool dr = h->buf.params != nullptr, terr = h->T > 0;
  typedef void (*launcher)(int, hipStream_t, const pgtt::KArgs&, const float*);
  static const launcher table[3][2][2][2] = {
      {{{pgtt_launch_physics_s1_0_0_0, pgtt_launch_physics_s1_0_0_1}, {pgtt_launch_physics_s1_0_1_0, pgtt_launch_physics_s1_0_1_1}},
       {{pgtt_launch_physics_s1_1_0_0, pgtt_launch_physics_s1_1_0_1}, {pgtt_launch_physics_s1_1_1_0, pgtt_launch_physics_s1_1_1_1}}},
      {{{pgtt_launch_physics_s4_0_0_0, pgtt_launch_physics_s4_0_0_1}, {pgtt_launch_physics_s4_0_1_0, pgtt_launch_physics_s4_0_1_1}},
       {{pgtt_launch_physics_s4_1_0_0, pgtt_launch_physics_s4_1_0_1}, {pgtt_launch_physics_s4_1_1_0, pgtt_launch_physics_s4_1_1_1}}},
      {{{pgtt_launch_physics_s2_0_0_0, pgtt_launch_physics_s2_0_0_1}, {pgtt_launch_physics_s2_0_1_0, pgtt_launch_physics_s2_0_1_1}},
       {{pgtt_launch_physics_s2_1_0_0, pgtt_launch_physics_s2_1_0_1}, {pgtt_launch_physics_s2_1_1_0, pgtt_launch_physics_s2_1_1_1}}}};
  // auto: a launch lasts as long as one wave's instruction stream while all its waves run concurrently, one per SIMD (<= 1024
  // waves), and the stream is the shorter the more lanes share an env: hex (4 envs per wave: line-search rows, Cholesky columns
  // and - on box terrain - the collision passes and contact slots split over four sub-lanes) up to 4096 envs (level4 0.16 ms
  // against 0.31 ms in the quad layout); oct (8 envs per wave) up to 8192 envs = still one round of waves (0.26 ms against 0.31 ms
  // for quad's 512 waves); beyond that QUAD (16 envs per wave): 16384 envs are ONE round of 1024 waves (0.34 ms) where oct needs
  // two (0.47 ms).  On box terrain that holds since round 4: the quad kernels used to stage the env's box centres / extents in LDS
  // (68 KB per workgroup = two per CU, i.e. two rounds at 16384 envs); they now read them from the L2-resident table (30 KB of
  // LDS = one workgroup per SIMD): level4 at 16384 / 32768 envs 41.8 / 47.6 M env-steps/s against 31.7 / 33.9 M in the oct layout.
  const int subs = h->layout != 0 ? h->layout : (h->N <= 4096 ? 4 : (h->N <= 8192 ? 2 : 1));
  const int per = 16 / subs;
  const launcher fn = table[subs == 1 ? 0 : (subs == 4 ? 1 : 2)][MODE][dr ? 1 : 0][terr ? 1 : 0];
  if (!fn) return fail(PGTT_E_STATE, "this build of the library does not contain the physics_kernel variant the call needs (lane layout / DR / terrain); use libpgtt.so");
  if (a_in.N > 0) fn((h->N + per - 1) / per, st, a, action);      // N = 0 in the arguments: availability probe only (pgtt_reset, before it writes anything)
  return PGTT_OK;
}

template <int OMODE>
void launch_observe(pgtt_env* h, const pgtt::KArgs& a, const float* action, hipStream_t st) {
  dim3 grid(h->N), block(64);
  if (h->T > 0) hipLaunchKernelGGL((pgtt::observe_kernel<OMODE, true>), grid, block, 0, st, a, action);
  else hipLaunchKernelGGL((pgtt::observe_kernel<OMODE, false>), grid, block, 0, st, a, action);
}

int check_ready(pgtt_env* h) {
  if (!h) return fail(PGTT_E_ARG, "null handle");
  if (!h->bound) return fail(PGTT_E_STATE, "pgtt_bind must be called before reset/step");
  return PGTT_OK;
}

}  // namespace

extern "C" {

const char* pgtt_last_error(void) { return g_err.c_str(); }
const char* pgtt_version(void) { return "pgtt-mi355x 0.5 (gfx950)"; }
int pgtt_obs_dims(const PgttConfig* cfg, int* state_dim, int* priv_dim) {
  if (!cfg || !state_dim || !priv_dim) return fail(PGTT_E_ARG, "pgtt_obs_dims: null argument");
  if (cfg->method != PGTT_METHOD_PGTT && cfg->method != PGTT_METHOD_BASELINE) return fail(PGTT_E_ARG, "pgtt_obs_dims: unknown method");
  *state_dim = cfg->method == PGTT_METHOD_BASELINE ? PGTT_OBS_BASELINE : PGTT_OBS;
  *priv_dim = cfg->method == PGTT_METHOD_BASELINE ? PGTT_PRIV_BASELINE : PGTT_PRIV;
  return PGTT_OK;
}

int pgtt_sizeof_model(void) { return (int)sizeof(PgttModel); }
int pgtt_sizeof_config(void) { return (int)sizeof(PgttConfig); }
int pgtt_sizeof_buffers(void) { return (int)sizeof(PgttBuffers); }

int pgtt_create(const PgttConfig* cfg, const PgttModel* model, int device, int num_envs, pgtt_handle* out) {
  if (!cfg || !model || !out) return fail(PGTT_E_ARG, "pgtt_create: null argument");
  if (num_envs <= 0) return fail(PGTT_E_ARG, "pgtt_create: num_envs must be positive");
  if (num_envs > (1 << 22)) return fail(PGTT_E_ARG, "pgtt_create: at most 4 194 304 envs per handle (the step kernel addresses its rows through 32-bit byte offsets)");
  if (cfg->n_substeps < 1 || cfg->n_substeps > 64) return fail(PGTT_E_ARG, "pgtt_create: n_substeps out of range");
  if (cfg->method != PGTT_METHOD_PGTT && cfg->method != PGTT_METHOD_BASELINE) return fail(PGTT_E_ARG, "pgtt_create: unknown method");
  if (cfg->lane_layout != PGTT_LAYOUT_AUTO && cfg->lane_layout != PGTT_LAYOUT_QUAD && cfg->lane_layout != PGTT_LAYOUT_OCT && cfg->lane_layout != PGTT_LAYOUT_HEX)
    return fail(PGTT_E_ARG, "pgtt_create: lane_layout must be PGTT_LAYOUT_AUTO, _QUAD, _OCT or _HEX");
  if (cfg->observe_form != PGTT_OBSERVE_FUSED && cfg->observe_form != PGTT_OBSERVE_SPLIT) return fail(PGTT_E_ARG, "pgtt_create: unknown observe_form");
  // values the kernels have as compile-time shapes, not as run-time parameters: anything else is refused here rather than computed wrongly
  if (cfg->history_len != 2) return fail(PGTT_E_ARG, "pgtt_create: history_len must be 2 (the history rows hold two samples of 12 joints, go2/configs.py)");
  if (cfg->history_update_steps < 1) return fail(PGTT_E_ARG, "pgtt_create: history_update_steps must be >= 1");
  if (cfg->episode_length < 1) return fail(PGTT_E_ARG, "pgtt_create: episode_length must be >= 1");
  if (!(cfg->sim_dt > 0.f) || !(cfg->ctrl_dt > 0.f) || cfg->n_substeps != (int)std::lround(cfg->ctrl_dt / cfg->sim_dt))
    return fail(PGTT_E_ARG, "pgtt_create: n_substeps must equal round(ctrl_dt / sim_dt) (mjx_env.MjxEnv.n_substeps)");
  if (std::fabs(model->timestep - cfg->sim_dt) > 1e-9f) return fail(PGTT_E_ARG, "pgtt_create: model.timestep must equal config.sim_dt (go2/base.py:55)");
  if (model->max_contact_points < 1 || model->max_contact_points > 4)
    return fail(PGTT_E_ARG, "pgtt_create: max_contact_points must be 1 .. 4 (a foot holds at most four box contacts in the kernels; the reference uses 4)");
  if (model->iterations < 1 || model->iterations > 64 || model->ls_iterations < 1 || model->ls_iterations > 64) return fail(PGTT_E_ARG, "pgtt_create: solver iteration counts out of range");
  {
    // mixed condim = max of the pair (mjx collision_driver): the constraint rows of the kernels are the 4-row pyramid of condim 3
    const int cf = model->foot_condim > model->floor_condim ? model->foot_condim : model->floor_condim;
    const int cb = model->foot_condim > model->box_condim ? model->foot_condim : model->box_condim;
    if (cf != 3 || cb != 3) return fail(PGTT_E_ARG, "pgtt_create: foot-floor and foot-box contacts must mix to condim 3 (pyramidal, 4 rows per contact)");
  }
  {
    // box contacts: the collision stage keeps PENETRATING (foot, box) pairs only, which is every pair MJX can activate as long as the mixed margin
    // max(margins) - max(gaps) of the pair is <= 0 (the reference: foot margin -0.001, box margin 0).  The plane contact has no such limit.
    const float mg = model->foot_margin > model->box_margin ? model->foot_margin : model->box_margin;
    const float gp = model->foot_gap > model->box_gap ? model->foot_gap : model->box_gap;
    if (mg - gp > 0.f) return fail(PGTT_E_ARG, "pgtt_create: foot-box contacts need max(foot_margin, box_margin) - max(foot_gap, box_gap) <= 0 (rows of non-penetrating box pairs are not formed)");
  }
  static const int expect_dof[12] = {9, 10, 11, 6, 7, 8, 15, 16, 17, 12, 13, 14};
  for (int a = 0; a < 12; a++)
    if (model->act_dof[a] != expect_dof[a]) return fail(PGTT_E_ARG, "pgtt_create: actuators must be declared FR,FL,RR,RL on joints FL,FR,RL,RR");
  for (int b = 1; b < 13; b++)
    if (model->body_quat[b][0] != 1.f || model->body_quat[b][1] != 0.f || model->body_quat[b][2] != 0.f || model->body_quat[b][3] != 0.f)
      return fail(PGTT_E_ARG, "pgtt_create: link frames must be unrotated (body_quat = 1 0 0 0)");
  if (model->foot_friction[0] != model->foot_friction[0]) return fail(PGTT_E_ARG, "pgtt_create: NaN in model");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PGTT_E_NODEVICE, "pgtt_create: no HIP device (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(PGTT_E_ARG, "pgtt_create: device index out of range");
  HIP_TRY(hipSetDevice(device));
  pgtt_env* h = new pgtt_env();
  h->device = device; h->N = num_envs; h->cfg = *cfg; h->model = *model;
  // lane layout and observe form come through the ABI (PgttConfig.lane_layout / observe_form), never from the environment:
  // AUTO picks by batch size (launch_physics); the fused observe kernel (four waves per SIMD since its reward terms are evaluated
  // lane-parallel) is faster than the split form at every batch size measured (16384 envs: 0.088 against 0.135 ms)
  h->layout = cfg->lane_layout;
  h->split_observe = cfg->observe_form == PGTT_OBSERVE_SPLIT;
  // a failure half way frees what the call allocated so far (pgtt_destroy walks the same fields)
  auto built = [&]() -> int {
    HIP_TRY(hipMalloc(&h->d_cfg, sizeof(PgttConfig)));
    HIP_TRY(hipMalloc(&h->d_model, sizeof(PgttModel)));
    HIP_TRY(hipMalloc(&h->d_handover, (size_t)num_envs * pgtt::kHandover * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_flag, sizeof(int)));
    HIP_TRY(hipHostMalloc(&h->h_flag, sizeof(int)));
    HIP_TRY(hipMemcpy(h->d_cfg, cfg, sizeof(PgttConfig), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_model, model, sizeof(PgttModel), hipMemcpyHostToDevice));
    for (int r = 0; r < pgtt_env::kRing; r++) for (int i = 0; i < 4; i++) HIP_TRY(hipEventCreate(&h->ev[r][i]));
    return PGTT_OK;
  };
  if (int rc = built()) { pgtt_destroy(h); return rc; }
  *out = h;
  return PGTT_OK;
}

int pgtt_destroy(pgtt_handle h) {
  if (!h) return PGTT_OK;
  hipSetDevice(h->device);
  if (h->d_cfg) hipFree(h->d_cfg);
  if (h->d_model) hipFree(h->d_model);
  if (h->d_handover) hipFree(h->d_handover);
  if (h->d_flag) hipFree(h->d_flag);
  if (h->h_flag) hipHostFree(h->h_flag);
  if (h->d_terrain) hipFree(h->d_terrain);
  if (h->d_cull) hipFree(h->d_cull);
  if (h->d_grid) hipFree(h->d_grid);
  for (int r = 0; r < pgtt_env::kRing; r++) for (int i = 0; i < 4; i++) if (h->ev[r][i]) hipEventDestroy(h->ev[r][i]);
  delete h;
  return PGTT_OK;
}

int pgtt_set_terrain(pgtt_handle h, const float* boxes, int T, int B) {
  if (!h) return fail(PGTT_E_ARG, "null handle");
  if (T < 0 || B < 0 || B > PGTT_MAX_BOX) return fail(PGTT_E_ARG, "pgtt_set_terrain: need 0 <= B <= 100, T >= 0");
  if (T > 0 && (!boxes || B == 0)) return fail(PGTT_E_ARG, "pgtt_set_terrain: null table");
  // the quad / oct step kernels address both tables through 32-bit BYTE offsets from their bases (PG_ADDR32, pgtt_physics.hip.h)
  if ((unsigned long long)T * (unsigned long long)B * sizeof(pgtt::TerrainBox) >= (1ull << 32) || (unsigned long long)T * pgtt::kGridG * pgtt::kGridG * sizeof(uint4) >= (1ull << 32))
    return fail(PGTT_E_ARG, "pgtt_set_terrain: terrain table too large (T * B * 80 bytes and T * 4096 bytes must stay below 4 GiB)");
  HIP_TRY(hipSetDevice(h->device));
  if (h->d_terrain) { HIP_TRY(hipFree(h->d_terrain)); h->d_terrain = nullptr; }
  if (h->d_cull) { HIP_TRY(hipFree(h->d_cull)); h->d_cull = nullptr; }
  if (h->d_grid) { HIP_TRY(hipFree(h->d_grid)); h->d_grid = nullptr; }
  h->T = 0; h->B = 0;
  if (T == 0) return PGTT_OK;
  std::vector<pgtt::TerrainBox> tab((size_t)T * B);
  for (size_t i = 0; i < tab.size(); i++) {
    const float* r = boxes + 10 * i;
    pgtt::TerrainBox& t = tab[i];
    t.px = r[0]; t.py = r[1]; t.pz = r[2];
    const float w = r[3], x = r[4], y = r[5], z = r[6];
    // rotation matrix exactly as the simulator's quat_to_mat (no normalisation), plain fp32 ops
    volatile float q00 = w * w, q01 = w * x, q02 = w * y, q03 = w * z, q11 = x * x, q12 = x * y, q13 = x * z;
    volatile float q22 = y * y, q23 = y * z, q33 = z * z;
    volatile float a0 = q00 + q11; volatile float a1 = a0 - q22; t.m00 = a1 - q33;
    volatile float d0 = q12 - q03; t.m01 = 2.f * d0;
    volatile float d1 = q13 + q02; t.m02 = 2.f * d1;
    volatile float d2 = q12 + q03; t.m10 = 2.f * d2;
    volatile float b0 = q00 - q11; volatile float b1 = b0 + q22; t.m11 = b1 - q33;
    volatile float d3 = q23 - q01; t.m12 = 2.f * d3;
    volatile float d4 = q13 - q02; t.m20 = 2.f * d4;
    volatile float d5 = q23 + q01; t.m21 = 2.f * d5;
    volatile float c1 = b0 - q22; t.m22 = c1 + q33;
    t.sx = r[7]; t.sy = r[8]; t.sz = r[9];
    t.rb = std::sqrt(r[7] * r[7] + r[8] * r[8] + r[9] * r[9]) * 1.000001f;
    t.hx = (std::fabs(t.m00) * r[7] + std::fabs(t.m01) * r[8] + std::fabs(t.m02) * r[9]) * 1.00001f + 1e-6f;
    t.hy = (std::fabs(t.m10) * r[7] + std::fabs(t.m11) * r[8] + std::fabs(t.m12) * r[9]) * 1.00001f + 1e-6f;
    t.hz = (std::fabs(t.m20) * r[7] + std::fabs(t.m21) * r[8] + std::fabs(t.m22) * r[9]) * 1.00001f + 1e-6f;
    t.pad = 0.f;
  }
  {
    // Terrain grid: kGridG x kGridG cells over [-E, E]^2, E = the reach of the boxes that are actually placed (the unused
    // placeholders of a variant are parked at ~(100 + k) m, terrain_scene_mjx.xml / terrain/generator.py:368-391).  Cell (ix, iy)
    // holds the set of boxes whose world AABB, grown by the largest foot radius (+ slack for the rounding of the cell index),
    // touches it; indices are clamped, so border cells reach to infinity and a parked box lands in a corner cell: the set of a
    // foot's cell is a superset of the boxes whose grown AABB contains the foot centre, wherever the foot is.
    const int G = pgtt::kGridG;
    float rmax = 0.f;
    for (int l = 0; l < PGTT_NLEG; l++) rmax = std::fmax(rmax, h->model.foot_radius[l]);
    const float grow = rmax + 1e-5f + 1e-4f;
    float E = 1.0f;
    for (const auto& t : tab)
      if (std::fabs(t.px) < 50.f && std::fabs(t.py) < 50.f) E = std::fmax(E, std::fmax(std::fabs(t.px) + t.hx, std::fabs(t.py) + t.hy) + grow);
    const float inv = (float)G / (2.f * E);
    std::vector<uint32_t> grid((size_t)T * G * G * 4, 0u);
    auto cell_of = [&](float x) { int c = (int)std::floor((x + E) * inv); return c < 0 ? 0 : (c > G - 1 ? G - 1 : c); };
    for (int v = 0; v < T; v++)
      for (int b = 0; b < B; b++) {
        const pgtt::TerrainBox& t = tab[(size_t)v * B + b];
        const int x0 = cell_of(t.px - t.hx - grow), x1 = cell_of(t.px + t.hx + grow), y0 = cell_of(t.py - t.hy - grow), y1 = cell_of(t.py + t.hy + grow);
        for (int iy = y0; iy <= y1; iy++)
          for (int ix = x0; ix <= x1; ix++) grid[(((size_t)v * G + iy) * G + ix) * 4 + (b >> 5)] |= 1u << (b & 31);
      }
    HIP_TRY(hipMalloc(&h->d_grid, grid.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(h->d_grid, grid.data(), grid.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    h->grid_E = E; h->grid_inv = inv;
  }
  HIP_TRY(hipMalloc(&h->d_terrain, tab.size() * sizeof(pgtt::TerrainBox)));
  HIP_TRY(hipMemcpy(h->d_terrain, tab.data(), tab.size() * sizeof(pgtt::TerrainBox), hipMemcpyHostToDevice));
  {
    std::vector<float4> cull(tab.size());
    for (size_t i = 0; i < tab.size(); i++) cull[i] = make_float4(tab[i].px, tab[i].py, tab[i].hx, tab[i].hy);
    HIP_TRY(hipMalloc(&h->d_cull, cull.size() * sizeof(float4)));
    HIP_TRY(hipMemcpy(h->d_cull, cull.data(), cull.size() * sizeof(float4), hipMemcpyHostToDevice));
  }
  h->T = T; h->B = B;
  h->labels_unchecked = true;
  return PGTT_OK;
}

int pgtt_bind(pgtt_handle h, const PgttBuffers* b) {
  if (!h || !b) return fail(PGTT_E_ARG, "pgtt_bind: null argument");
  if (!b->state || !b->istate || !b->frame || !b->scan_z || !b->obs_state || !b->obs_priv || !b->reward || !b->done || !b->metrics)
    return fail(PGTT_E_ARG, "pgtt_bind: state, istate, frame, scan_z, obs_state, obs_priv, reward, done, metrics are required");
  if (h->cfg.autoreset && (!b->first_state || !b->first_obs))
    return fail(PGTT_E_ARG, "pgtt_bind: autoreset needs first_state and first_obs");
  h->buf = *b;
  h->bound = true;
  h->labels_unchecked = true;
  return PGTT_OK;
}

int pgtt_reset(pgtt_handle h, uint64_t seed, int64_t env_id_offset, const uint8_t* mask, void* stream) {
  if (int rc = check_ready(h)) return rc;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((h->N + 63) / 64), block(64);
  if (h->T > 0 && h->buf.variant && (mask == nullptr || h->labels_unchecked)) {
    // every env's terrain-variant label must name one of the T variants of pgtt_set_terrain.  One launch + a 4-byte read-back BEFORE anything is
    // written; this is the one place where the library waits for the stream.  Whole-batch resets (mask NULL: off the steady-state path, AutoReset
    // lives in the step) always check; a MASKED reset - the form a caller may put into its loop - checks only when pgtt_bind / pgtt_set_terrain
    // ran since the last check and is otherwise asynchronous like every other entry point.  Not while a capture is under way (a capture cannot
    // wait; the query itself fails with hipErrorStreamCaptureImplicit on the null stream while ANOTHER stream captures in global mode: treated as
    // "capturing") - the kernels clamp the label either way.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }
    if (cap == hipStreamCaptureStatusNone) {
      HIP_TRY(hipMemsetAsync(h->d_flag, 0, sizeof(int), st));
      hipLaunchKernelGGL(variant_range_kernel, grid, block, 0, st, h->buf.variant, h->N, h->T, h->d_flag);
      HIP_TRY(hipMemcpyAsync(h->h_flag, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (*h->h_flag != 0)
        return fail(PGTT_E_ARG, "pgtt_reset: " + std::to_string(*h->h_flag) + " terrain variant label(s) outside [0, " + std::to_string(h->T) + ") in PgttBuffers.variant");
      h->labels_unchecked = false;
    }
  }
  h->seed = seed; h->env_off = env_id_offset;
  pgtt::KArgs a = make_args(h, mask, 0.f);
  { pgtt::KArgs probe = a; probe.N = 0; if (int rc = launch_physics<pgtt::MODE_FORWARD>(h, probe, nullptr, st)) return rc; }
  hipLaunchKernelGGL(pgtt::reset_pose_kernel<0>, grid, block, 0, st, a);
  if (int rc = launch_physics<pgtt::MODE_FORWARD>(h, a, nullptr, st)) return rc;          // mjx_env.init -> forward
  launch_observe<pgtt::OBS_SCAN_LIFT>(h, a, nullptr, st);         // lift by the max terrain height under the footprint
  a.write_qpos = 1;
  if (int rc = launch_physics<pgtt::MODE_FORWARD>(h, a, nullptr, st)) return rc;          // mjx.forward on the lifted pose
  launch_observe<pgtt::OBS_RESET>(h, a, nullptr, st);             // info, obs, first-state capture
  HIP_TRY(hipGetLastError());
  return PGTT_OK;
}

int pgtt_physics(pgtt_handle h, const float* action, void* stream) {
  if (int rc = check_ready(h)) return rc;
  if (!action) return fail(PGTT_E_ARG, "pgtt_physics: null action");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  pgtt::KArgs a = make_args(h, nullptr, 0.f);
  h->timing_now = h->timing && (h->timing_tick++ % h->timing_period) == h->timing_period / 2;     // not the first step after a synchronisation: its launch latency is not the kernel's
  if (h->timing_now) {
    h->ev_slot = (h->ev_slot + 1) % pgtt_env::kRing;
    if (int rc = harvest(h, h->ev_slot)) return rc;
    HIP_TRY(hipEventRecord(h->ev[h->ev_slot][0], st));
  }
  if (int rc = launch_physics<pgtt::MODE_STEP>(h, a, action, st)) return rc;
  if (h->timing_now) HIP_TRY(hipEventRecord(h->ev[h->ev_slot][1], st));
  HIP_TRY(hipGetLastError());
  return PGTT_OK;
}

// same_step: called by pgtt_step right behind this step's physics launch on the same stream - the observe kernel may then take the physics
// outputs from the hand-over record instead of the caller-visible rows (which it must use when the caller had a chance to edit them)
static int observe_impl(pgtt_handle h, const float* action, void* stream, bool same_step) {
  if (int rc = check_ready(h)) return rc;
  if (!action) return fail(PGTT_E_ARG, "pgtt_observe: null action");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  pgtt::KArgs a = make_args(h, nullptr, 0.f);
  if (same_step) a.handover_r = h->d_handover;
  const bool timed = h->timing_now && h->ev_slot >= 0;
  if (timed) { h->ev_chain[h->ev_slot] = same_step; if (!same_step) HIP_TRY(hipEventRecord(h->ev[h->ev_slot][2], st)); }      // an event record idles the device for ~3.5 us
  if (h->split_observe) {
    // scan + observation rows (one env per wave), then rewards / bookkeeping / wrapper (one env per lane)
    launch_observe<pgtt::OBS_STEP_OBS>(h, a, action, st);
    hipLaunchKernelGGL(pgtt::task_kernel<0>, dim3((h->N + 63) / 64), dim3(64), 0, st, a, action);
  } else {
    launch_observe<pgtt::OBS_STEP>(h, a, action, st);
  }
  if (timed) { HIP_TRY(hipEventRecord(h->ev[h->ev_slot][3], st)); h->ev_used[h->ev_slot] = true; h->ev_valid = true; }
  HIP_TRY(hipGetLastError());
  return PGTT_OK;
}

int pgtt_observe(pgtt_handle h, const float* action, void* stream) { return observe_impl(h, action, stream, false); }

int pgtt_step(pgtt_handle h, const float* action, void* stream) {
  if (int rc = pgtt_physics(h, action, stream)) return rc;
  return observe_impl(h, action, stream, true);
}

int pgtt_scan(pgtt_handle h, float yaw_override_or_nan, void* stream) {
  if (int rc = check_ready(h)) return rc;
  HIP_TRY(hipSetDevice(h->device));
  pgtt::KArgs a = make_args(h, nullptr, yaw_override_or_nan);
  launch_observe<pgtt::OBS_SCAN_ONLY>(h, a, nullptr, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return PGTT_OK;
}

int pgtt_interval_reduce(pgtt_handle h, float* acc_dev, float env_steps, int accumulate, void* stream) {
  if (int rc = check_ready(h)) return rc;
  if (!acc_dev) return fail(PGTT_E_ARG, "pgtt_interval_reduce: null output");
  if (!h->buf.interval_sums) return fail(PGTT_E_STATE, "pgtt_interval_reduce: no interval_sums buffer is bound");
  HIP_TRY(hipSetDevice(h->device));
  hipLaunchKernelGGL(pgtt::interval_reduce_kernel<0>, dim3(PGTT_NMETRIC + 3), dim3(256), 0, (hipStream_t)stream, h->buf.interval_sums, h->N, PGTT_NMETRIC + 2, acc_dev, env_steps, accumulate);
  HIP_TRY(hipGetLastError());
  return PGTT_OK;
}

int pgtt_set_test_overrides(pgtt_handle h, float rng_value_or_nan, int scan_preset) {
  if (!h) return fail(PGTT_E_ARG, "null handle");
  if (!h->cfg.test_hooks) return fail(PGTT_E_STATE, "pgtt_set_test_overrides: the handle was not created with PgttConfig.test_hooks");
  h->test_rng_fix = rng_value_or_nan;
#ifdef PGTT_OBS_STOP
  h->test_scan_preset = scan_preset;      // 100 + k: phase boundary at which the step's observe waves leave (tools/gpu_observe_instr.py)
#else
  h->test_scan_preset = scan_preset != 0;
#endif
  return PGTT_OK;
}

int pgtt_enable_timing(pgtt_handle h, int enable) {
  if (!h) return fail(PGTT_E_ARG, "null handle");
  h->timing = enable != 0; h->ev_valid = false; h->ev_slot = -1;
  h->timing_period = enable > 1 ? enable : 1; h->timing_tick = 0; h->timing_now = false;
  for (int r = 0; r < pgtt_env::kRing; r++) h->ev_used[r] = false;
  h->sum_phys = 0.0; h->sum_obs = 0.0; h->n_timed = 0;
  return PGTT_OK;
}

int pgtt_last_kernel_ms(pgtt_handle h, float* physics_ms, float* observe_ms) {
  if (!h || !physics_ms || !observe_ms) return fail(PGTT_E_ARG, "pgtt_last_kernel_ms: null argument");
  if (!h->timing || !h->ev_valid || h->ev_slot < 0 || !h->ev_used[h->ev_slot]) return fail(PGTT_E_STATE, "pgtt_last_kernel_ms: timing not enabled or no step recorded");
  hipEvent_t* e = h->ev[h->ev_slot];
  HIP_TRY(hipEventSynchronize(e[3]));
  HIP_TRY(hipEventElapsedTime(physics_ms, e[0], e[1]));
  HIP_TRY(hipEventElapsedTime(observe_ms, e[h->ev_chain[h->ev_slot] ? 1 : 2], e[3]));
  return PGTT_OK;
}

int pgtt_kernel_ms_mean(pgtt_handle h, float* physics_ms, float* observe_ms, int* steps) {
  if (!h || !physics_ms || !observe_ms || !steps) return fail(PGTT_E_ARG, "pgtt_kernel_ms_mean: null argument");
  if (!h->timing) return fail(PGTT_E_STATE, "pgtt_kernel_ms_mean: timing not enabled");
  for (int r = 0; r < pgtt_env::kRing; r++) if (int rc = harvest(h, r)) return rc;
  if (h->n_timed == 0) return fail(PGTT_E_STATE, "pgtt_kernel_ms_mean: no step recorded");
  *physics_ms = (float)(h->sum_phys / h->n_timed); *observe_ms = (float)(h->sum_obs / h->n_timed); *steps = (int)h->n_timed;
  return PGTT_OK;
}

}  // extern "C"
