// capi.hip — the C ABI (include/gymnet_amd.h) over the HIP kernels.  Host code only; compiled by hipcc.
//
// There is NO CPU fallback in this library: without a usable AMD GPU every compute entry point fails
// with GYMNET_ERR_NO_DEVICE / GYMNET_ERR_HIP.  The CPU restatement under oracle/ is test infrastructure
// and is never linked here.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "handle.hpp"

using namespace gymnet;

namespace {
thread_local std::string g_last_error;
constexpr int64_t kSmallHostPath = 4096;
constexpr float FMAX = 3.4028234663852886e38f;
constexpr float PI_F = 3.14159265358979323846f;
}  // namespace

namespace gymnet {

// Observation bounds: CartPoleEnv.cs:46-48 (high = [x_thr*2, float.MaxValue, theta_thr*2, float.MaxValue]).
// Algorithmic bytes per env-step: SURVEY.md §8(a)/(d).
const EnvDesc kEnvs[4] = {
    {"CartPole-v1", 4, 4, true, false, true, 2, 0.f, 0.f,
     {-4.8000002f, -FMAX, -0.41887903f, -FMAX}, {4.8000002f, FMAX, 0.41887903f, FMAX}, 0.f, 1.f, 41, 41, {-1, -1, -1, -1}},
    {"Pendulum-v1", 2, 3, false, true, false, 0, -2.f, 2.f,
     {-1.f, -1.f, -8.f}, {1.f, 1.f, 8.f}, -16.2736044f, 0.f, 37, 33, {-1, 2, -1, -1}},
    {"MountainCar-v0", 2, 2, true, false, false, 3, 0.f, 0.f,
     {-1.2f, -0.07f}, {0.6f, 0.07f}, -1.f, -1.f, 25, 25, {-1, -1, -1, -1}},
    {"Acrobot-v1", 4, 6, false, false, false, 3, 0.f, 0.f,
     {-1.f, -1.f, -1.f, -1.f, -4.f * PI_F, -9.f * PI_F}, {1.f, 1.f, 1.f, 1.f, 4.f * PI_F, 9.f * PI_F}, -1.f, 0.f, 65, 57, {-1, -1, 4, 5}},
};

void set_last_error(const char *msg) { g_last_error = msg ? msg : ""; }

int fail(gymnet_vecenv *h, int status, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (h) h->err = buf;
    // HIP keeps the last error of the calling thread until somebody reads it, and the launchers end with `return
    // hipGetLastError()`: a HIP failure reported here must not make the NEXT (valid) launch on this thread fail too.
    // A pure argument-validation failure made no HIP call and does not touch the runtime.
    if (status == GYMNET_ERR_HIP || status == GYMNET_ERR_OOM || status == GYMNET_ERR_RCCL) (void)hipGetLastError();
    return status;
}

}  // namespace gymnet

namespace {

template <class T>
int dalloc(gymnet_vecenv *h, T **p, size_t count) {
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T) + 16);
    if (e != hipSuccess) return fail(h, GYMNET_ERR_OOM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    h->owned.push_back(q);
    *p = static_cast<T *>(q);
    return GYMNET_OK;
}

// Host-boundary staging (actions in, row-major observations out, reset mask) is allocated on first use: a
// device-resident consumer (bench.py, a GPU policy) never pays for it — at 2^27 CartPole lanes it is 2.6 GiB.
int ensure_staging(gymnet_vecenv *h, bool actions, bool pack, bool mask) {
    if (actions && !h->d_actions) ST_TRY(dalloc(h, (int32_t **)&h->d_actions, (size_t)h->padded));
    if (pack && !h->d_pack) ST_TRY(dalloc(h, (char **)&h->d_pack, (size_t)h->padded * h->desc->obs_dim * h->esz));
    if (mask && !h->d_mask) ST_TRY(dalloc(h, &h->d_mask, (size_t)h->padded));
    return GYMNET_OK;
}

void destroy_graph(GraphEntry &g) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
    g.exec = nullptr; g.graph = nullptr;
}

void drop_graphs(gymnet_vecenv *h) {
    for (auto &g : h->graphs) destroy_graph(g);
    h->graphs.clear();
}

int apply_policy(gymnet_vecenv *h, const gymnet_launch_policy &p, bool strict);
template <class R>
StepArgsT<R> make_step_args(gymnet_vecenv *h, const void *d_actions) {
    StepArgsT<R> a{};
    const bool alias = h->desc->alias;
    a.state = static_cast<R *>(h->d_state);
    a.state_out = static_cast<R *>((h->double_buffer && alias) ? h->d_state_alt : h->d_state);
    a.obs = static_cast<R *>((h->double_buffer && !alias) ? h->d_obs_alt : h->d_obs);
    a.obs_in = static_cast<const R *>(h->d_obs);
    a.action = d_actions;
    a.reward = h->d_reward;
    a.done = h->d_done;
    a.sbd = h->d_sbd;
    a.tick2 = h->d_tick2;
    // the dense "last finished episode per lane" arrays are maintained by the step kernel itself unless the caller opted for
    // compact records only (GYMNET_FLAG_COMPACT_RECORDS_ONLY with DONE_LIST): then the getters apply the records on demand
    const bool dense = !(h->compact_only && h->d_done_list);
    a.final_obs = dense ? static_cast<R *>(h->d_final_obs) : nullptr;
    a.done_list = h->d_done_list;
    a.done_count2 = h->d_done_count2;
    a.done_cap = h->done_cap;
    a.ep_ret = h->d_ep_ret; a.ep_len = h->d_ep_len; a.fin_ret = dense ? h->d_fin_ret : nullptr; a.fin_len = dense ? h->d_fin_len : nullptr;
    a.rec_ret = h->d_rec_ret; a.rec_len = h->d_rec_len; a.rec_obs = static_cast<R *>(h->d_rec_obs);
    a.lane_seed = h->d_lane_seed;
    a.after_done = h->d_after_done;
    a.n = h->n; a.state_stride = h->sstride; a.obs_stride = h->ostride;
    a.lane_offset = (uint64_t)h->cfg.lane_offset;
    a.seed = h->seed;
    a.parity = (int32_t)h->tslot;
    a.cparity = (int32_t)(h->step_launches & 1u);
    a.max_episode_steps = h->cfg.max_episode_steps;
    return a;
}

// after a step launch with GYMNET_FLAG_DOUBLE_BUFFER: what was written becomes current
void swap_buffers(gymnet_vecenv *h) {
    if (!h->double_buffer) return;
    if (h->desc->alias) { std::swap(h->d_state, h->d_state_alt); h->d_obs = h->d_state; h->d_obs_alt = h->d_state_alt; }
    else std::swap(h->d_obs, h->d_obs_alt);
    h->cur ^= 1;
}

// Gathers the sharded done list of the most recent step — and the records written beside it — into compact arrays
// (stream-ordered, non-blocking); with `dense`, also applies the records to the dense per-lane arrays.  NULL outputs are skipped.
template <class R>
int compact_done_typed(gymnet_vecenv *h, int32_t *d_list, float *d_ret, int32_t *d_len, void *d_obs, int64_t capacity, uint32_t *d_count, bool dense) {
    CompactArgsT<R> c{};
    c.counts = h->d_done_count2 + (size_t)h->last_cparity * kShards * kCountStride;
    c.list = h->d_done_list; c.cap = h->done_cap;
    c.rec_ret = h->d_rec_ret; c.rec_len = h->d_rec_len; c.rec_obs = static_cast<const R *>(h->d_rec_obs); c.obs_dim = h->desc->obs_dim;
    c.out_list = d_list; c.out_ret = d_ret; c.out_len = d_len; c.out_obs = static_cast<R *>(d_obs); c.out_capacity = capacity; c.out_count = d_count;
    if (dense) { c.dense_ret = h->d_rec_ret ? h->d_fin_ret : nullptr; c.dense_len = h->d_fin_len; c.dense_obs = h->d_rec_obs ? static_cast<R *>(h->d_final_obs) : nullptr; }
    c.n = h->n;
    HIP_TRY(h, launch_compact_done(c, h->stream));
    return GYMNET_OK;
}
// d_obs: terminal observations of the handle's state scalar (float, or double for a GYMNET_FLAG_F64 handle)
int compact_done(gymnet_vecenv *h, int32_t *d_list, float *d_ret, int32_t *d_len, void *d_obs, int64_t capacity, uint32_t *d_count,
                 bool dense = false) {
    return h->f64 ? compact_done_typed<double>(h, d_list, d_ret, d_len, d_obs, capacity, d_count, dense)
                  : compact_done_typed<float>(h, d_list, d_ret, d_len, d_obs, capacity, d_count, dense);
}

// device address of state row k of the CURRENT state (its own row of d_state, or the observation row that holds it)
void *state_row(gymnet_vecenv *h, int k) {
    const int m = h->desc->alias ? -1 : h->desc->state_row_in_obs[k];
    return m < 0 ? row_at(h->d_state, k, h->sstride, h->esz) : row_at(h->d_obs, m, h->ostride, h->esz);
}

// Applies the fields of `p` that are not -1.  strict: a value the handle cannot run — or that would not take effect for it (a
// multi-lane form on a batch that is not whole groups, a reset form the env has no use for) — is an error and nothing changes;
// else it is ignored (probe builds).  gymnet_vecenv_kernel_name stays the authority on what the next launch runs.
int apply_policy(gymnet_vecenv *h, const gymnet_launch_policy &p, bool strict) {
    LaunchCfg c = h->lcfg;
    int graph_mode = h->graph_mode;
    const bool acrobot = h->cfg.env_id == GYMNET_ENV_ACROBOT;
    const bool wide2 = acrobot || h->f64;            // the env's wide form is two lanes per thread (else four)
    auto bad = [&](const char *what, int v) -> int {
        return strict ? fail(h, GYMNET_ERR_INVALID_ARG, "launch policy: %s = %d is not available for this handle", what, v) : GYMNET_OK;
    };
    if (p.vec != -1) {
        const bool ok = p.vec == 1 || (p.vec == 4 && h->can_vec4 && !wide2) || (p.vec == 2 && h->can_vec2 && wide2);
        if (ok) c.vec = p.vec; else ST_TRY(bad("vec", p.vec));
    }
    if (p.block != -1) { if (p.block == 64 || p.block == 128 || p.block == 256) c.block = p.block; else ST_TRY(bad("block", p.block)); }
    if (p.nt != -1) { if (p.nt == 0 || p.nt == 12 || p.nt == 15) c.nt = p.nt; else ST_TRY(bad("nt", p.nt)); }
    if (p.sequential_lanes != -1) {
        if (p.sequential_lanes >= 1 && p.sequential_lanes <= (h->f64 ? 4 : 5) && (acrobot || h->f64 || p.sequential_lanes == 1)) c.items = p.sequential_lanes;
        else ST_TRY(bad("sequential_lanes", p.sequential_lanes));
    }
    if (p.reset_form != -1) {
        const bool has_form1 = h->desc->alias;       // the wave-compacted reset hands back the state only (CartPole, MountainCar)
        if (p.reset_form == 0 || (p.reset_form == 1 && (has_form1 || !strict))) c.reset_form = p.reset_form; else ST_TRY(bad("reset_form", p.reset_form));
    }
    if (p.lds_pipe != -1) {
        if (p.lds_pipe == 0 || (p.lds_pipe == 1 && h->lds_ok)) c.lds_pipe = p.lds_pipe; else ST_TRY(bad("lds_pipe", p.lds_pipe));
    }
    if (p.occupancy_lds_bytes != -1) {
        if (p.occupancy_lds_bytes >= 0 && p.occupancy_lds_bytes <= 160 * 1024) c.lds_bytes = p.occupancy_lds_bytes;
        else ST_TRY(bad("occupancy_lds_bytes", p.occupancy_lds_bytes));
    }
    if (p.graph != -1) { if (p.graph == 0 || p.graph == 1) graph_mode = p.graph; else if (p.graph == -2) graph_mode = -1; else ST_TRY(bad("graph", p.graph)); }
    if (strict && p.sequential_lanes > 1) {
        // an explicitly requested multi-lane form must be the one the launcher resolves to: lean variant, the lane
        // width it is defined for, whole 2 * items * 256-lane groups for the pair form
        int rvec = 1, rseq = 1;
        resolved_step_shape(h->cfg.env_id, h->f64, h->autoreset, h->extras, c, h->n, &rvec, &rseq);
        if (rseq != p.sequential_lanes)
            return fail(h, GYMNET_ERR_INVALID_ARG, "launch policy: sequential_lanes = %d would not take effect for this handle (vec %d, %s variant, "
                        "%lld lanes: the launcher resolves to %d)", p.sequential_lanes, c.vec, h->extras ? "bookkeeping" : "lean", (long long)h->n, rseq);
    }
    h->graph_mode = graph_mode;
    if (std::memcmp(&c, &h->lcfg, sizeof c) != 0) {
        h->lcfg = c;
        drop_graphs(h);          // captured launches froze the old configuration
    }
    return GYMNET_OK;
}

// The launch configuration a handle starts with (gymnet_vecenv_set_launch_policy changes it afterwards).
void default_policy(gymnet_vecenv *h) {
    const EnvDesc &d = *h->desc;
    const gymnet_config *cfg = &h->cfg;
    if (h->f64) {
        // float64 CartPole: two lanes per thread (one dwordx4 per state row and direction) where the rows are 16-byte aligned
        // (external observation buffers may not be); stream policy by the bytes a vector step moves, with the thresholds of the
        // float32 path below
        const size_t step_bytes = (size_t)h->n * bytes_per_step(h);
        const bool can2 = aligned16(h->d_state) && (h->sstride % 2 == 0) && (!h->d_state_alt || aligned16(h->d_state_alt));
        h->can_vec4 = false; h->can_vec2 = can2; h->lds_ok = false;
        // measured at 2^20 lanes (73 MiB per step; us per step, gpurun_out r4): every stream non-temporal 14.3, state cacheable 14.8,
        // nothing non-temporal 16.2, one lane per thread 15.6
        // reset_form 1 (two lanes per reset in the wave-compacted form, step_kernels.hpp) for the one-shot kernel and the fused rollout:
        // one-shot 13.7 -> 13.4 us at 2^20 lanes, bookkeeping rollout 7.5 -> 6.9 us per step, lean rollout 5.9 -> 6.0
        // (profiles/rollout_reset_forms_r05.txt); the multi-pair kernel below always draws once per thread-group of pairs
        h->lcfg = LaunchCfg{can2 ? 2 : 1, 256, 15, 0, 1, can2 ? 1 : 0, 0, h->simds};
        // Stream policy by the bytes a vector step moves (round 6, profiles/f64_sizes_r06.txt; us per 2^20 lanes, one-shot kernel,
        // nothing non-temporal | state cacheable (12) | every stream non-temporal (15)):
        //   2^21 lanes 12.9 | 14.8 | 13.7     3 * 2^20  11.8 | 13.4 | 12.6     2^22  12.1 | 13.8 | 14.1     2^23  13.1 | 13.5 | 12.6
        // While everything a step touches fits the 256 MiB Infinity Cache (<= 300 MiB moved: the state is rewritten in place), plain
        // cacheable accesses win by 6-13 % — the next launch finds its inputs on the die; beyond it nothing can stay and every stream
        // is marked non-temporal (rounds 4-5 kept the state cacheable between 96 and 768 MiB: 4-7 % slower at every size measured).
        if (step_bytes > ((size_t)96 << 20) && step_bytes <= ((size_t)300 << 20)) h->lcfg.nt = 0;
        // The multi-pair kernel (step_kernel_pipe2<CartPole64, k>: a thread owns k lane pairs, all loads first, then advance pair after
        // pair, ONE wave-compacted reset for all of them, then the state rows) wins where it runs as ONE resident generation that fills
        // the chip: four pairs hold 185 VGPRs = two waves per SIMD (2048 waves of 512 lanes), two pairs three (3072 waves of 256 lanes).
        // us per step, one-shot | 2 pairs | 4 pairs (profiles/f64_sizes_r05.txt, one box):
        //   2^19 lanes 6.27 | 6.42 | 7.57      3 * 2^18  10.94 | 8.46 | 10.19      2^20  13.13 | 12.17 | 11.19
        //   5 * 2^18   16.64 | 16.82 | 17.40   6 * 2^18  21.17 | 19.78 | 19.58     2^21  27.72 | 28.56 | 29.64   (larger: one-shot)
        // (round 4, 271 VALU per env-step, per-pair drain-loop reset: one-shot 14.4, 2 pairs 13.1, 4 pairs 14.4 at 2^20; round 5 before
        // the deferred reset: 13.4-13.6 | 13.1-13.4 | 12.8-13.2.)  Lean variant only; without auto-reset whole 2 * k * 256-lane groups
        // only (the launcher falls back otherwise).  A wave count just under a whole number per SIMD is as good (10^6 lanes: 11.7 us against 13.0).
        // Beyond one generation the kernel is launched slice by slice (step_kernels.hpp pipe2_chunks) and is no faster than the one-shot
        // kernel — 2^21 lanes: 27.9 us in two slices, 35.2 as a looping grid, 27.4 one-shot with the same mask, 25.8 one-shot
        // cacheable (profiles/f64_sizes_r06.txt): what makes 2^20 lanes fast is that the launch's written lines fit the L2s and leave
        // after the kernel, not the launch shape — so it is not selected there.  (Round 5's window test also let FOUR pairs through at
        // three waves per SIMD, which they cannot hold: 3 * 2^19 lanes ran at 16.6 us per 2^20 lanes against 13.3.  Fixed.)
        if (can2 && h->n >= ((int64_t)3 << 18)) {            // (below: ramp-bound, fewer and fatter waves lose)
            for (int items : {4, 2}) {
                if (!h->autoreset && h->n % ((int64_t)512 * items) != 0) continue;     // (the auto-reset form takes any batch size)
                const double waves_per_simd = (double)((h->n + 128 * items - 1) / ((int64_t)128 * items)) / (double)h->simds;   // MI355X: 256 CUs x 4 SIMDs
                // NOT one wave more: the kernel holds 191 VGPRs, two waves per SIMD are resident, and a 2049th wave starts when an
                // earlier one has finished — 2^20 + 2 lanes: 19.2 us against 11.1 (profiles/f64_sizes_r05.txt, second table)
                const bool fills = items == 4 ? (waves_per_simd >= 1.9 && waves_per_simd <= 2.0) : (waves_per_simd >= 2.85 && waves_per_simd <= 3.0);
                if (fills) { h->lcfg.items = items; h->lcfg.nt = 15; break; }
            }
        }
        return;
    }
    // dwordx4 streams need 16-byte aligned component arrays; external buffers may not be
    const bool can_vec4 = aligned16(h->d_state) && aligned16(h->d_obs) && (h->sstride % 4 == 0) && (h->ostride % 4 == 0) &&
                          (!h->d_obs_alt || aligned16(h->d_obs_alt));
    // Launch policy, measured on MI355X with tools/probe_step.hip and bench.py (profiles/probe_r01.txt),
    // keyed on the bytes one vector step moves (lanes x algorithmic bytes per env-step):
    //  - <= 2^19 lanes (round 1-4: "<= 24 MiB per vector step"): scalar lanes — 4x the waves hide latency better than dwordx4 on a small grid;
    //  - <= 48 MiB (2^20 CartPole lanes): dwordx4, every stream non-temporal;
    //  - <= 300 MiB (the step's whole footprint fits the 256 MiB Infinity Cache): dwordx4, nothing non-temporal (round 6);
    //  - <= 768 MiB (the state fits it): dwordx4, state cacheable, action / reward / done streamed
    //    past it, so the next launch re-reads the state from the cache;
    //  - larger: nothing can stay resident — scalar lanes, every stream non-temporal;
    //  - Acrobot (RK4, ALU-bound: ~650 VALU per env-step) always takes scalar lanes: 21.7 vs 27.7 us at 2^20.
    const size_t step_bytes = (size_t)h->n * (size_t)d.algorithmic_bytes;
    const bool alu_bound = cfg->env_id == GYMNET_ENV_ACROBOT;
    h->lcfg.simds = h->simds;
    // Round 5: the small-batch rule is a LANE count, not a byte count.  Scalar lanes win while the launch is ONE resident generation
    // of waves — n <= 8 waves x 1024 SIMDs x 64 lanes = 2^19 — and lose beyond it (a second generation of 64-lane waves starts when the
    // first retires); the byte threshold had been fitted on CartPole alone (2^19 lanes = 21.5 MB) and kept MountainCar (25 B per lane)
    // on scalar lanes up to 10^6 lanes: 5.53 us against 4.53 with 16-byte lanes; 3 * 2^18 lanes 4.35 against 3.93
    // (profiles/small_batches_r05.txt; CartPole and Pendulum cross over at the same lane count).
    const bool one_generation = h->n <= ((int64_t)8 * h->simds * 64);
    if (alu_bound || one_generation) { h->lcfg.vec = 1; h->lcfg.nt = 15; }
    // Round 6 (profiles/trough_r06.txt): while everything a step touches fits the 256 MiB Infinity Cache — up to ~300 MiB
    // moved, the state being rewritten in place — and the launch is more than ~1.25 generations of waves, NO stream is marked
    // non-temporal.  us per 2^20 lanes, mask 0 against the former choice (15 up to 48 MiB, 12 beyond), two boxes:
    //   CartPole     5 * 2^18 lanes 6.26 / 6.56   2^21 7.03 / 7.33   3 * 2^20 6.86 / 7.54   2^22 6.72 / 7.24   6 * 2^20 6.40 / 6.78
    //                2^23 6.80 / 6.85 (the crossover)   2^24 7.96 / 6.52 (mask 12 stays)
    //   MountainCar  3 * 2^19 4.31 / 5.10   2^21 3.83 / 4.48   2^22 4.14 / 4.33   2^23 3.92 / 4.10   2^24 3.85 / 3.87
    //   Pendulum     3 * 2^19 5.93 / 5.42 and 2^21 6.05 / 5.30 (loses: two of its four written rows are never read back),
    //                2^22 5.51 / 6.27   2^23 5.68 / 5.68 — hence its later start
    else if (h->n >= (cfg->env_id == GYMNET_ENV_PENDULUM ? (int64_t)3 << 20 : (int64_t)5 << 18) && step_bytes <= ((size_t)300 << 20)) { h->lcfg.vec = 4; h->lcfg.nt = 0; }
    else if (step_bytes <= ((size_t)48 << 20)) { h->lcfg.vec = 4; h->lcfg.nt = 15; }
    else if (step_bytes <= ((size_t)768 << 20)) { h->lcfg.vec = 4; h->lcfg.nt = 12; }
    else { h->lcfg.vec = 1; h->lcfg.nt = 15; }
    if (!can_vec4) h->lcfg.vec = 1;
    // Acrobot's wide form is TWO lanes per thread whose arithmetic rides the packed FP32 instructions (envs.hpp).  Opt-in
    // (gymnet_launch_policy.vec = 2): bit-identical, 287 instead of 454 VALU per env-step, and slower — 15.0 vs 14.2 us at 2^20
    // lanes, 13.6 vs 12.3 us per 2^20 lanes at 2^23 (profiles/acrobot_probes_r02.txt, docs/ledger.md §4a).
    const bool can_vec2 = aligned_to(h->d_state, 8) && aligned_to(h->d_obs, 8) && (h->sstride % 2 == 0) && (h->ostride % 2 == 0) &&
                          (!h->d_obs_alt || aligned_to(h->d_obs_alt, 8));
    // Acrobot's multi-lane kernel (step_kernel_pipe: all loads first, then compute / store lane after lane) wins where the
    // one-shot kernel would run as ~2 lock-step wave generations: it turns the launch into ONE generation of 4096 waves whose
    // stores drain under the next lane's arithmetic.  Measured (profiles/acrobot_probes_r02.txt, us per 2^20 lanes, one-shot
    // vs k = ceil(n / 2^18) lanes per thread): n = 2^19 15.1 vs 14.5, 3*2^18 13.9 vs 13.9, 2^20 14.9 vs 13.1, 5*2^18 13.7 vs
    // 13.1; beyond that the one-shot kernel's generations overlap by themselves (6*2^18: 13.7 vs 13.8; 2^21: 13.4 vs 13.3).
    // k is n / 2^18 ROUNDED, not rounded up: one lane more than 2^20 used to select k = 5 — 17.5 us per step against 12.5 at 2^20 lanes
    // (profiles/ragged_r05.txt); with k = 4 the 4097th workgroup is one more wave on a chip that holds eight per SIMD.
    if (alu_bound && h->n >= ((int64_t)1 << 19) && h->n < ((int64_t)11 << 17)) h->lcfg.items = (int)((h->n + ((int64_t)1 << 17)) >> 18);
    // Wave-compacted fused reset (kernels.hip: reset_pending_wave) wherever the dwordx4 lean kernel of an env whose observation IS
    // its state runs: the wave's finished sub-lanes are drawn in ONE Philox pass by its first lanes instead of 1.6 mostly idle
    // passes.  CartPole at 2^20 lanes: 6.91 -> 6.52 us per launch, bit-identical (profiles/forms_probe_r03.txt).
    h->lcfg.reset_form = (d.alias && h->lcfg.vec == 4) ? 1 : 0;
    // producer / consumer form of the multi-lane kernel (opt-in, gymnet_launch_policy.lds_pipe = 1): whole 512-lane tiles and
    // 16-byte aligned rows only.  Bit-identical and SLOWER at 2^20 lanes — 13.5-14.9 vs 11.7-12.7 us (profiles/acrobot_lds_r03.txt):
    // the computing waves' waits on memory instructions drop from 34 % to 21 % of their cycles, the s_barrier per tile adds more.
    const bool lds_ok = alu_bound && can_vec4 && (h->n % 512) == 0;
    // 64-thread workgroups for the dwordx4 kernels of the smallest payloads (< 40 MiB per vector step at dwordx4: MountainCar and
    // Pendulum at 2^20 lanes): such a launch is ramp / drain bound, and one-wave workgroups ramp and retire faster — MountainCar 4.92-5.05
    // -> 4.55-4.73 us, Pendulum 5.97 -> 5.88 us; CartPole (41 MiB) does not gain (tools/gpu_nt_ab_r03.sh, profiles/block_nt_probe_r03.txt)
    if (h->lcfg.vec == 4 && step_bytes < ((size_t)40 << 20)) h->lcfg.block = 64;
    h->can_vec4 = can_vec4; h->can_vec2 = can_vec2; h->lds_ok = lds_ok;
}

void recompute_extras(gymnet_vecenv *h) {
    h->extras = (h->cfg.flags & (GYMNET_FLAG_DONE_LIST | GYMNET_FLAG_EPISODE_STATS | GYMNET_FLAG_FINAL_OBS)) != 0 ||
                h->d_lane_seed != nullptr;
}

}  // namespace

namespace gymnet {

// Env.Seed(int) on a handle the caller has ENTERed
int seed_handle(gymnet_vecenv *h, uint64_t seed) {
    h->seed = seed;
    h->tick = 0;
    drop_graphs(h);   // the seed is a (frozen) kernel argument of captured launches
    h->d_lane_seed = nullptr;   // back to one key for all lanes (d_lane_seed_buf is kept for the next Seed(int[]))
    recompute_extras(h);
    return write_tick(h);
}

// one vector step = one kernel launch; bumps the host mirrors of the device-side counters
int launch_one_step(gymnet_vecenv *h, const void *d_actions) {
    int cparity;
    if (h->f64) {
        const StepArgsT<double> a = make_step_args<double>(h, d_actions);
        cparity = a.cparity;
        HIP_TRY(h, launch_step(h->cfg.env_id, h->autoreset, h->extras, a, h->lcfg, h->stream));
    } else {
        const StepArgsT<float> a = make_step_args<float>(h, d_actions);
        cparity = a.cparity;
        HIP_TRY(h, launch_step(h->cfg.env_id, h->autoreset, h->extras, a, h->lcfg, h->stream));
    }
    swap_buffers(h);
    h->last_cparity = cparity;
    h->tick += 1;
    h->tslot ^= 1;
    h->step_launches += 1;
    h->lane_steps += (uint64_t)h->n;
    return GYMNET_OK;
}

template <class R>
static int reset_lanes_typed(gymnet_vecenv *h, const uint8_t *d_mask) {
    ResetArgsT<R> r{};
    r.state = static_cast<R *>(h->d_state); r.obs = static_cast<R *>(h->d_obs); r.sbd = h->d_sbd; r.done = h->d_done;
    r.mask = d_mask;
    r.tick2 = h->d_tick2; r.lane_seed = h->d_lane_seed;
    r.ep_ret = h->d_ep_ret; r.ep_len = h->d_ep_len;
    r.n = h->n; r.state_stride = h->sstride; r.obs_stride = h->ostride;
    r.lane_offset = (uint64_t)h->cfg.lane_offset; r.seed = h->seed;
    r.parity = (int32_t)h->tslot;
    HIP_TRY(h, launch_reset(h->cfg.env_id, r, h->stream));
    h->tick += 1;
    h->tslot ^= 1;
    return GYMNET_OK;
}

int launch_reset_lanes(gymnet_vecenv *h, const uint8_t *d_mask) {
    return h->f64 ? reset_lanes_typed<double>(h, d_mask) : reset_lanes_typed<float>(h, d_mask);
}

int write_tick(gymnet_vecenv *h) {
    uint64_t both[2] = {h->tick, h->tick};
    HIP_TRY(h, hipMemcpyAsync(h->d_tick2, both, sizeof both, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));   // `both` is a stack temporary
    return GYMNET_OK;
}

// row-major packing / host export of the CURRENT observations, typed by the handle's state scalar
static hipError_t pack_current_obs(gymnet_vecenv *h, const void *src_obs, int64_t stride, void *dst) {
    return h->f64 ? launch_pack_obs(h->desc->obs_dim, static_cast<const double *>(src_obs), stride, static_cast<double *>(dst), h->n, h->stream)
                  : launch_pack_obs(h->desc->obs_dim, static_cast<const float *>(src_obs), stride, static_cast<float *>(dst), h->n, h->stream);
}
static hipError_t export_current(gymnet_vecenv *h, bool host_form, void *obs_out, float *reward_out, uint8_t *done_out) {
    const int O = h->desc->obs_dim;
    if (h->f64) {
        const double *o = static_cast<const double *>(h->d_obs);
        return host_form ? launch_export_host(O, o, h->ostride, h->d_reward, h->d_done, static_cast<double *>(obs_out), reward_out, done_out, h->n, h->stream)
                         : launch_export_small(O, o, h->ostride, h->d_reward, h->d_done, static_cast<double *>(obs_out), reward_out, done_out, h->n, h->stream);
    }
    const float *o = static_cast<const float *>(h->d_obs);
    return host_form ? launch_export_host(O, o, h->ostride, h->d_reward, h->d_done, static_cast<float *>(obs_out), reward_out, done_out, h->n, h->stream)
                     : launch_export_small(O, o, h->ostride, h->d_reward, h->d_done, static_cast<float *>(obs_out), reward_out, done_out, h->n, h->stream);
}

// queues the copies of the current results to host buffers (any may be NULL) without the closing synchronize;
// only for handles without the host-mapped small-batch path
int queue_copy_out(gymnet_vecenv *h, void *obs_out, float *reward_out, uint8_t *done_out) {
    const EnvDesc &d = *h->desc;
    if (obs_out) {
        ST_TRY(ensure_staging(h, false, true, false));
        HIP_TRY(h, pack_current_obs(h, h->d_obs, h->ostride, h->d_pack));
        HIP_TRY(h, hipMemcpyAsync(obs_out, h->d_pack, (size_t)h->n * d.obs_dim * h->esz, hipMemcpyDeviceToHost, h->stream));
    }
    if (reward_out) HIP_TRY(h, hipMemcpyAsync(reward_out, h->d_reward, (size_t)h->n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (done_out) HIP_TRY(h, hipMemcpyAsync(done_out, h->d_done, (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
    return GYMNET_OK;
}

// copy the current results to host buffers (any may be NULL); blocks
int copy_out(gymnet_vecenv *h, void *obs_out, float *reward_out, uint8_t *done_out) {
    const EnvDesc &d = *h->desc;
    const size_t esz = h->esz;      // element size of an observation at the boundary
    if (h->pin_block && (obs_out || reward_out || done_out) && (!obs_out || obs_out == h->pin_obs) &&
        (!reward_out || reward_out == h->pin_reward) && (!done_out || done_out == h->pin_done)) {
        // the caller reads the library's pinned buffers: ONE kernel writes the results across PCIe, no staging, no memcpy calls
        HIP_TRY(h, export_current(h, /*host_form=*/true, obs_out, reward_out, done_out));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return GYMNET_OK;
    }
    if (h->hm_block) {   // latency path for small batches: one export kernel into host-mapped memory, one sync, host memcpy
        if (obs_out || reward_out || done_out)
            HIP_TRY(h, export_current(h, /*host_form=*/false, h->hm_obs, reward_out ? h->hm_reward : nullptr, done_out ? h->hm_done : nullptr));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (obs_out) std::memcpy(obs_out, h->hm_obs, (size_t)h->n * d.obs_dim * esz);
        if (reward_out) std::memcpy(reward_out, h->hm_reward, (size_t)h->n * sizeof(float));
        if (done_out) std::memcpy(done_out, h->hm_done, (size_t)h->n);
        return GYMNET_OK;
    }
    ST_TRY(queue_copy_out(h, obs_out, reward_out, done_out));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return GYMNET_OK;
}

// Discrete.Contains over the staged batch (only with GYMNET_FLAG_VALIDATE_ACTIONS); blocks
int validate_staged_actions(gymnet_vecenv *h, const void *d_actions) {
    const EnvDesc &d = *h->desc;
    if (!(h->cfg.flags & GYMNET_FLAG_VALIDATE_ACTIONS) || d.box_action) return GYMNET_OK;
    HIP_TRY(h, hipMemsetAsync(h->d_bad, 0, sizeof(uint32_t), h->stream));
    HIP_TRY(h, launch_validate_discrete(static_cast<const int32_t *>(d_actions), h->n, d.action_n, h->d_bad, h->stream));
    uint32_t bad = 0;
    HIP_TRY(h, hipMemcpyAsync(&bad, h->d_bad, sizeof bad, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (bad)
        return fail(h, GYMNET_ERR_INVALID_ACTION, "Action is outside of the configured action space. (%u of %lld lanes, Discrete(%d))",
                    bad, (long long)h->n, d.action_n);
    return GYMNET_OK;
}

// stages host actions; *d_use receives the device-visible pointer the step kernel should read
int stage_host_actions(gymnet_vecenv *h, const void *actions, const void **d_use, bool validate_now) {
    if (!actions) return fail(h, GYMNET_ERR_INVALID_ARG, "actions is null");
    if (h->hm_block) {   // the previous step's kernel has completed (every host-boundary call ends with a sync)
        std::memcpy(h->hm_actions, actions, (size_t)h->n * 4);
        *d_use = h->hm_actions;
    } else {
        ST_TRY(ensure_staging(h, true, false, false));
        HIP_TRY(h, hipMemcpyAsync(h->d_actions, actions, (size_t)h->n * 4, hipMemcpyHostToDevice, h->stream));
        *d_use = h->d_actions;
    }
    return validate_now ? validate_staged_actions(h, *d_use) : GYMNET_OK;
}

}  // namespace gymnet

namespace gymnet {

// `steps` vector steps, one kernel launch each (caller has ENTERed the handle)
int rollout_steps(gymnet_vecenv *h, const void *d_actions, int64_t steps, int64_t action_stride, int64_t ring, int graph_mode) {
    if (!d_actions) return fail(h, GYMNET_ERR_INVALID_ARG, "d_actions is null");
    if (steps < 0 || ring < 1 || action_stride < 0) return fail(h, GYMNET_ERR_INVALID_ARG, "bad steps/ring/action_stride");
    if (h->lcfg.vec > 1 && (!aligned_to(d_actions, 4 * h->lcfg.vec) || (action_stride % h->lcfg.vec) != 0))
        return fail(h, GYMNET_ERR_INVALID_ARG, "d_actions and action_stride must be %d-byte aligned", 4 * h->lcfg.vec);
    const char *base = static_cast<const char *>(d_actions);
    auto slice = [&](int64_t t) -> const void * { return base + (size_t)((t % ring) * action_stride) * 4; };
    if ((h->cfg.flags & GYMNET_FLAG_VALIDATE_ACTIONS) && !h->desc->box_action && steps > 0) {
        // Discrete.Contains over every slice the rollout will read, before any state changes (one readback)
        HIP_TRY(h, hipMemsetAsync(h->d_bad, 0, sizeof(uint32_t), h->stream));
        for (int64_t k = 0; k < (steps < ring ? steps : ring); ++k)
            HIP_TRY(h, launch_validate_discrete(static_cast<const int32_t *>(slice(k)), h->n, h->desc->action_n, h->d_bad, h->stream));
        uint32_t bad = 0;
        HIP_TRY(h, hipMemcpyAsync(&bad, h->d_bad, sizeof bad, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (bad) return fail(h, GYMNET_ERR_INVALID_ACTION, "Action is outside of the configured action space. (%u lane-steps, Discrete(%d))", bad, h->desc->action_n);
    }

    // graph length: a multiple of `ring` (so every replay starts at slice 0) and even (so the double-buffered
    // device tick, done-count halves and observation buffers are the same at every replay)
    int64_t glen = (ring % 2 == 0) ? ring : 2 * ring;
    int64_t t = 0;
    // Graph replay only pays while the host launch path (~3.5 us per launch) is the bottleneck: measured on
    // MI355X, replay beats eager launches up to ~2^18 CartPole lanes (2.6 vs 5.7 us/step at 2^16), ties at 2^19
    // and loses at 2^20 (8.08 vs 7.85 us/step: a kernel node costs more than a back-to-back stream launch).
    const bool launch_bound = (size_t)h->n * bytes_per_step(h) < ((size_t)24 << 20);
    if (graph_mode < 0) graph_mode = h->graph_mode;          // gymnet_vecenv_set_launch_policy(.graph)
    const bool use_graph = graph_mode >= 0 ? graph_mode != 0 : launch_bound;
    if (use_graph && glen <= 4096 && steps >= glen) {
        const int parity = h->tslot, cparity = (int)(h->step_launches & 1u), cur = h->cur;
        GraphEntry *ge = nullptr;
        for (auto &g : h->graphs)
            if (g.actions == d_actions && g.len == glen && g.stride == action_stride && g.ring == ring && g.parity == parity &&
                g.cparity == cparity && g.cur == cur)
                ge = &g;
        if (!ge) {
            const uint64_t tick0 = h->tick, sl0 = h->step_launches, ls0 = h->lane_steps;
            const int slot0 = h->tslot, cur0 = h->cur, lcp0 = h->last_cparity;
            HIP_TRY(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            int st = GYMNET_OK;
            for (int64_t k = 0; k < glen && st == GYMNET_OK; ++k) st = launch_one_step(h, slice(k));
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamEndCapture(h->stream, &graph);
            // capturing launched nothing: rewind the host mirrors (glen is even, so the buffer pair is back where it was)
            h->tick = tick0; h->step_launches = sl0; h->lane_steps = ls0; h->tslot = slot0; h->last_cparity = lcp0;
            if (h->cur != cur0) swap_buffers(h);
            if (st != GYMNET_OK) { if (graph) (void)hipGraphDestroy(graph); return st; }
            if (e != hipSuccess) return fail(h, GYMNET_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
            hipGraphExec_t exec = nullptr;
            e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            if (e != hipSuccess) { (void)hipGraphDestroy(graph); return fail(h, GYMNET_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
            if (h->graphs.size() >= kMaxGraphs) {   // a trainer that keeps reallocating its action buffer must not leak graphs
                size_t lru = 0;
                for (size_t i = 1; i < h->graphs.size(); ++i) if (h->graphs[i].last_use < h->graphs[lru].last_use) lru = i;
                HIP_TRY(h, hipStreamSynchronize(h->stream));          // the evicted exec may still be running
                destroy_graph(h->graphs[lru]);
                h->graphs.erase(h->graphs.begin() + (long)lru);
            }
            h->graphs.push_back(GraphEntry{d_actions, glen, action_stride, ring, parity, cparity, cur, graph, exec, 0});
            ge = &h->graphs.back();
        }
        ge->last_use = ++h->graph_clock;
        for (; t + glen <= steps; t += glen) {
            HIP_TRY(h, hipGraphLaunch(ge->exec, h->stream));
            h->tick += (uint64_t)glen;
            h->step_launches += (uint64_t)glen;
            h->lane_steps += (uint64_t)glen * (uint64_t)h->n;
            h->last_cparity = (int)((h->step_launches - 1) & 1u);
        }
    }
    for (; t < steps; ++t) ST_TRY(launch_one_step(h, slice(t)));
    return GYMNET_OK;
}

}  // namespace gymnet

namespace gymnet {

// ~1.5 us per poll over PCIe: the resident kernel leaves after ~4-5 ms without a command (rounds 4-5: 40000 polls = 50-75 ms).  While
// it spins it occupies the handle's stream and one wave, and anything that waits for the WHOLE device — hipDeviceSynchronize,
// torch.cuda.synchronize(), a hipFree out of a caching allocator — or that serialises dispatch (rocprofv3 --pmc, AMD_SERIALIZE_KERNEL)
// waits for this timeout: it bounds what a host that mixes an env loop with other GPU work in one process can lose per
// such call.  A loop that steps back to back never sees it; a step after a longer pause pays one relaunch (~20 us, the launch path's
// cost).  GYMNET_FLAG_RESIDENT is opt-in in every facade for the same reason.
constexpr uint64_t kResidentIdlePolls = 3000;

// a polite busy-wait: the host thread spins on a cache line the GPU writes over PCIe
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#endif
}

template <class R>
static int resident_start_typed(gymnet_vecenv *h) {
    StepArgsT<R> a = make_step_args<R>(h, h->mb_dev->actions);       // the kernel reads its actions straight from the mailbox
    ResetArgsT<R> r{};
    r.state = static_cast<R *>(h->d_state); r.obs = static_cast<R *>(h->d_obs); r.sbd = h->d_sbd; r.done = h->d_done; r.mask = nullptr;
    r.tick2 = h->d_tick2; r.lane_seed = h->d_lane_seed; r.ep_ret = h->d_ep_ret; r.ep_len = h->d_ep_len;
    r.n = h->n; r.state_stride = h->sstride; r.obs_stride = h->ostride; r.lane_offset = (uint64_t)h->cfg.lane_offset; r.seed = h->seed;
    r.parity = (int32_t)h->tslot;
    HIP_TRY(h, launch_resident(h->cfg.env_id, h->autoreset, h->extras, a, r, h->mb_dev, kResidentIdlePolls, h->stream));
    return GYMNET_OK;
}

static int resident_start(gymnet_vecenv *h) {
    // the kernel picks the sequence number it continues from out of the mailbox: nothing is pending at this point
    __atomic_store_n(&h->mb->exited, 0u, __ATOMIC_RELAXED);
    __atomic_store_n(&h->mb->done_seq, h->mb_seq, __ATOMIC_RELAXED);
    __atomic_store_n(&h->mb->cmd_seq, h->mb_seq, __ATOMIC_RELEASE);
    ST_TRY(h->f64 ? resident_start_typed<double>(h) : resident_start_typed<float>(h));
    h->resident_running = true;
    return GYMNET_OK;
}

// Tells a running resident kernel to leave and waits until it has; the engine tick it kept in a register comes back through
// the mailbox (and through both halves of d_tick2 for the next launch).
int resident_stop(gymnet_vecenv *h) {
    if (!h->resident_running) return GYMNET_OK;
    if (!__atomic_load_n(&h->mb->exited, __ATOMIC_ACQUIRE)) {
        h->mb->cmd = kMailboxExit;
        __atomic_store_n(&h->mb->cmd_seq, ++h->mb_seq, __ATOMIC_RELEASE);
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->resident_running = false;
    h->tick = h->mb->tick;
    return GYMNET_OK;
}

// One command through the mailbox: post it, wait for its results, copy them out.  actions: host, or NULL (resets).
static int resident_command(gymnet_vecenv *h, uint32_t cmd, const void *actions, void *obs_out, float *reward_out, uint8_t *done_out) {
    const EnvDesc &d = *h->desc;
    if (h->resident_running && __atomic_load_n(&h->mb->exited, __ATOMIC_ACQUIRE)) ST_TRY(resident_stop(h));   // it timed out since the last call
    if (!h->resident_running) ST_TRY(resident_start(h));
    if (actions) std::memcpy(h->mb->actions, actions, (size_t)h->n * 4);
    h->mb->cmd = cmd;
    const uint64_t seq = ++h->mb_seq;
    __atomic_store_n(&h->mb->cmd_seq, seq, __ATOMIC_RELEASE);
    uint64_t spins = 0;
    const auto t_start = std::chrono::steady_clock::now();
    while (__atomic_load_n(&h->mb->done_seq, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0x3FFu) == 0) {
            if (__atomic_load_n(&h->mb->exited, __ATOMIC_ACQUIRE)) {
                // the kernel left (idle timeout) in the instant this command was posted, without serving it: it touches
                // nothing any more — wait for it to be gone, start a new one, which finds the command pending
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                h->resident_running = false;
                h->tick = h->mb->tick;
                if (__atomic_load_n(&h->mb->done_seq, __ATOMIC_ACQUIRE) == seq) break;
                __atomic_store_n(&h->mb->exited, 0u, __ATOMIC_RELAXED);
                ST_TRY(h->f64 ? resident_start_typed<double>(h) : resident_start_typed<float>(h));
                h->resident_running = true;
            } else if (hipStreamQuery(h->stream) == hipSuccess) {
                // the stream is empty although nobody answered: the kernel is gone.  Nothing will ever read the posted command; resync
                // the tick from what the kernel last published so that the next call starts a fresh kernel on consistent state
                h->resident_running = false;
                h->tick = h->mb->tick;
                __atomic_store_n(&h->mb->done_seq, seq, __ATOMIC_RELAXED);      // the command is void: a restarted kernel must not run it late
                return fail(h, GYMNET_ERR_HIP, "the resident kernel ended without answering command %llu", (unsigned long long)seq);
            }
            if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(10)) {
                //  do not leave the command in the mailbox for a late-waking kernel to apply to NEWER actions: overwrite it
                // with EXIT, wait until the kernel has left (it publishes its tick on the way out), void the sequence number
                h->mb->cmd = kMailboxExit;
                __atomic_store_n(&h->mb->cmd_seq, ++h->mb_seq, __ATOMIC_RELEASE);
                (void)hipStreamSynchronize(h->stream);
                h->resident_running = false;
                h->tick = h->mb->tick;
                return fail(h, GYMNET_ERR_HIP, "the resident kernel did not answer command %llu within 10 s; it has been stopped and the next call starts a new one",
                            (unsigned long long)seq);
            }
        }
        cpu_relax();
    }
    const char *mb_obs = reinterpret_cast<const char *>(h->mb) + kMailboxObsOffset;
    if (obs_out) std::memcpy(obs_out, mb_obs, (size_t)h->n * d.obs_dim * h->esz);
    if (reward_out) std::memcpy(reward_out, h->mb->reward, (size_t)h->n * 4);
    if (done_out) std::memcpy(done_out, h->mb->done, (size_t)h->n);
    h->tick = h->mb->tick;
    if (cmd == kMailboxStep) h->lane_steps += (uint64_t)h->n;
    return GYMNET_OK;
}

// the resident path serves this handle's host-boundary steps right now (everything that changes what the kernel's arguments
// describe — seeds, per-lane keys, state, policy — goes through ENTER, which makes the kernel leave first; it restarts with fresh arguments)
static bool resident_serves(const gymnet_vecenv *h) { return h->resident && !h->async_pending; }

}  // namespace gymnet

extern "C" {

int gymnet_abi_version(void) { return GYMNET_ABI_VERSION; }

const char *gymnet_status_string(int status) {
    switch (status) {
        case GYMNET_OK: return "ok";
        case GYMNET_ERR_INVALID_ARG: return "invalid argument";
        case GYMNET_ERR_INVALID_ACTION: return "Action is outside of the configured action space.";
        case GYMNET_ERR_HIP: return "HIP runtime error";
        case GYMNET_ERR_OOM: return "out of memory";
        case GYMNET_ERR_NO_DEVICE: return "no AMD GPU available (this engine has no CPU fallback)";
        case GYMNET_ERR_ALREADY_STEPPING: return "already running an async step";
        case GYMNET_ERR_NOT_STEPPING: return "not running an async step";
        case GYMNET_ERR_UNSUPPORTED: return "unsupported for this environment / configuration";
        case GYMNET_ERR_RCCL: return "RCCL error";
        default: return "unknown status";
    }
}

const char *gymnet_last_error(void) { return g_last_error.c_str(); }

int gymnet_device_count(int *count) {
    return guarded([&]() -> int {
    if (!count) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "count is null");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0) {
        *count = 0;
        (void)hipGetLastError();
        return fail(nullptr, GYMNET_ERR_NO_DEVICE, "hipGetDeviceCount: %s", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    }
    *count = c;
    return GYMNET_OK;
    });
}

int gymnet_env_describe(int env_id, gymnet_env_info *out) {
    return guarded([&]() -> int {
    if (!out) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "out is null");
    if (env_id < 0 || env_id > 3) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "unknown env_id %d", env_id);
    const EnvDesc &d = kEnvs[env_id];
    std::memset(out, 0, sizeof *out);
    out->struct_size = sizeof *out;
    out->env_id = env_id;
    std::snprintf(out->name, sizeof out->name, "%s", d.name);
    out->state_dim = d.state_dim; out->obs_dim = d.obs_dim; out->obs_aliases_state = d.alias;
    out->action_is_box = d.box_action; out->action_n = d.action_n;
    out->action_low = d.action_low; out->action_high = d.action_high;
    for (int k = 0; k < 8; ++k) { out->obs_low[k] = d.obs_low[k]; out->obs_high[k] = d.obs_high[k]; }
    out->reward_low = d.reward_low; out->reward_high = d.reward_high;
    out->algorithmic_bytes_per_step = d.algorithmic_bytes;
    out->traffic_bytes_per_step = d.traffic_bytes;
    for (int k = 0; k < 8; ++k) out->state_row_in_obs[k] = (k < d.state_dim && !d.alias) ? d.state_row_in_obs[k] : -1;
    return GYMNET_OK;
    });
}

int gymnet_vecenv_destroy(gymnet_vecenv *h) {
    return guarded([&]() -> int {
    if (!h) return GYMNET_OK;
    DeviceScope dev_scope;
    (void)hipSetDevice(h->device);
    if (h->resident_running) (void)resident_stop(h);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->mb) (void)hipHostFree(h->mb);
    drop_graphs(h);
    for (void *p : h->owned) (void)hipFree(p);
    if (h->d_ep_seg) (void)hipFree(h->d_ep_seg);
    if (h->hm_block) (void)hipHostFree(h->hm_block);
    if (h->pin_block) (void)hipHostFree(h->pin_block);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return GYMNET_OK;
    });
}

int gymnet_vecenv_create(const gymnet_config *cfg, gymnet_vecenv **out) {
    return guarded([&]() -> int {
    if (!out) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (!cfg) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "cfg is null");
    if (cfg->struct_size != sizeof(gymnet_config))
        return fail(nullptr, GYMNET_ERR_INVALID_ARG, "cfg.struct_size %u != %zu (ABI mismatch)", cfg->struct_size, sizeof(gymnet_config));
    if (cfg->env_id < 0 || cfg->env_id > 3) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "unknown env_id %d", cfg->env_id);
    if (cfg->num_envs <= 0 || cfg->num_envs > (int64_t)1 << 31)
        return fail(nullptr, GYMNET_ERR_INVALID_ARG, "num_envs %lld out of range [1, 2^31]", (long long)cfg->num_envs);
    if (cfg->lane_offset < 0) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "lane_offset < 0");
    constexpr uint32_t kKnownFlags = GYMNET_FLAG_AUTORESET | GYMNET_FLAG_VALIDATE_ACTIONS | GYMNET_FLAG_DONE_LIST | GYMNET_FLAG_EPISODE_STATS |
                                     GYMNET_FLAG_FINAL_OBS | GYMNET_FLAG_DOUBLE_BUFFER | GYMNET_FLAG_F64 | GYMNET_FLAG_COMPACT_RECORDS_ONLY |
                                     GYMNET_FLAG_RESIDENT;
    if (cfg->flags & ~kKnownFlags)      // a flag from a newer header must not be silently ignored by an older library
        return fail(nullptr, GYMNET_ERR_INVALID_ARG, "unknown flag bits 0x%x (this library is ABI %d)", cfg->flags & ~kKnownFlags, GYMNET_ABI_VERSION);
    if (cfg->max_episode_steps < 0 || (cfg->max_episode_steps > 0 && !(cfg->flags & GYMNET_FLAG_EPISODE_STATS)))
        return fail(nullptr, GYMNET_ERR_INVALID_ARG, "max_episode_steps needs GYMNET_FLAG_EPISODE_STATS");
    if ((cfg->flags & GYMNET_FLAG_FINAL_OBS) && !(cfg->flags & GYMNET_FLAG_AUTORESET))
        return fail(nullptr, GYMNET_ERR_INVALID_ARG, "GYMNET_FLAG_FINAL_OBS needs GYMNET_FLAG_AUTORESET");
    if (cfg->d_ext_obs && cfg->ext_obs_stride < cfg->num_envs)
        return fail(nullptr, GYMNET_ERR_INVALID_ARG, "ext_obs_stride %lld < num_envs", (long long)cfg->ext_obs_stride);
    if (cfg->d_ext_obs_alt && (!cfg->d_ext_obs || !(cfg->flags & GYMNET_FLAG_DOUBLE_BUFFER)))
        return fail(nullptr, GYMNET_ERR_INVALID_ARG, "d_ext_obs_alt needs d_ext_obs and GYMNET_FLAG_DOUBLE_BUFFER");
    if (cfg->d_ext_obs_alt && cfg->d_ext_obs_alt == cfg->d_ext_obs)
        return fail(nullptr, GYMNET_ERR_INVALID_ARG, "d_ext_obs_alt must be a different buffer than d_ext_obs");
    if ((cfg->flags & GYMNET_FLAG_F64) && cfg->env_id != GYMNET_ENV_CARTPOLE)
        return fail(nullptr, GYMNET_ERR_UNSUPPORTED, "GYMNET_FLAG_F64 exists for CartPole only (the one env whose float64 arithmetic the reference defines)");
    const size_t esz_cfg = (cfg->flags & GYMNET_FLAG_F64) ? 8 : 4;
    if ((cfg->d_ext_obs && !aligned_to(cfg->d_ext_obs, (int)esz_cfg)) || (cfg->d_ext_obs_alt && !aligned_to(cfg->d_ext_obs_alt, (int)esz_cfg)))
        return fail(nullptr, GYMNET_ERR_INVALID_ARG, "d_ext_obs / d_ext_obs_alt must be aligned to the observation element (%zu bytes)", esz_cfg);
    if (cfg->flags & GYMNET_FLAG_RESIDENT) {
        if (cfg->num_envs > kMailboxLanes) return fail(nullptr, GYMNET_ERR_UNSUPPORTED, "GYMNET_FLAG_RESIDENT serves up to %d lanes (one wave)", kMailboxLanes);
        if ((cfg->flags & (GYMNET_FLAG_DONE_LIST | GYMNET_FLAG_FINAL_OBS | GYMNET_FLAG_DOUBLE_BUFFER)) || cfg->d_ext_obs || cfg->stream)
            return fail(nullptr, GYMNET_ERR_UNSUPPORTED, "GYMNET_FLAG_RESIDENT cannot be combined with DONE_LIST / FINAL_OBS / DOUBLE_BUFFER / d_ext_obs / a caller's stream");
    }
    if ((cfg->flags & GYMNET_FLAG_COMPACT_RECORDS_ONLY) && !(cfg->flags & GYMNET_FLAG_DONE_LIST))
        return fail(nullptr, GYMNET_ERR_INVALID_ARG, "GYMNET_FLAG_COMPACT_RECORDS_ONLY needs GYMNET_FLAG_DONE_LIST");

    int ndev = 0;
    int s = gymnet_device_count(&ndev);
    if (s != GYMNET_OK) return s;
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "device %d not in [0, %d)", cfg->device, ndev);

    gymnet_vecenv *h = new (std::nothrow) gymnet_vecenv();
    if (!h) return fail(nullptr, GYMNET_ERR_OOM, "host allocation failed");
    h->cfg = *cfg;
    h->desc = &kEnvs[cfg->env_id];
    h->device = cfg->device;
    h->n = cfg->num_envs;
    h->seed = cfg->seed;
    h->autoreset = (cfg->flags & GYMNET_FLAG_AUTORESET) != 0;
    h->double_buffer = (cfg->flags & GYMNET_FLAG_DOUBLE_BUFFER) != 0;
    h->f64 = (cfg->flags & GYMNET_FLAG_F64) != 0;
    h->esz = h->f64 ? 8 : 4;
    h->compact_only = (cfg->flags & GYMNET_FLAG_COMPACT_RECORDS_ONLY) != 0;
    const EnvDesc &d = *h->desc;
    DeviceScope dev_scope;

#define CREATE_TRY(expr)                      \
    do {                                      \
        int s_ = (expr);                      \
        if (s_ != GYMNET_OK) { gymnet_vecenv_destroy(h); return s_; } \
    } while (0)
#define CREATE_HIP(expr)                                                                    \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            int s_ = fail(nullptr, GYMNET_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            gymnet_vecenv_destroy(h);                                                       \
            return s_;                                                                      \
        }                                                                                   \
    } while (0)

    CREATE_HIP(hipSetDevice(h->device));
    {   // SIMD units of this device: the launch policy's waves-per-SIMD arithmetic (default_policy) and the looped multi-pair kernel's
        // resident generation (pipe2_shape) are sized from the device the handle lives on, not from an MI355X literal
        int cus = 0;
        CREATE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
        h->simds = cus > 0 ? cus * 4 : 1024;
    }
    if (cfg->stream) { h->stream = static_cast<hipStream_t>(cfg->stream); }
    else { CREATE_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }

    const int64_t padded = (h->n + 63) / 64 * 64;   // every component array starts 256-byte aligned
    h->padded = padded;
    const size_t esz = h->esz;      // bytes per state / observation element: 4, or 8 for a GYMNET_FLAG_F64 handle
    if (cfg->d_ext_obs) {
        if (d.alias) { h->d_state = cfg->d_ext_obs; h->sstride = cfg->ext_obs_stride; h->d_obs = h->d_state; h->ostride = h->sstride; }
        else {
            h->d_obs = cfg->d_ext_obs; h->ostride = cfg->ext_obs_stride;
            CREATE_TRY(dalloc(h, (char **)&h->d_state, (size_t)padded * d.state_dim * esz)); h->sstride = padded;
        }
    } else {
        CREATE_TRY(dalloc(h, (char **)&h->d_state, (size_t)padded * d.state_dim * esz)); h->sstride = padded;
        if (d.alias) { h->d_obs = h->d_state; h->ostride = h->sstride; }
        else { CREATE_TRY(dalloc(h, (char **)&h->d_obs, (size_t)padded * d.obs_dim * esz)); h->ostride = padded; }
    }
    if (h->double_buffer) {   // the second observation buffer (for aliasing envs: the second STATE buffer), same stride
        void *alt = cfg->d_ext_obs_alt;
        if (!alt) CREATE_TRY(dalloc(h, (char **)&alt, (size_t)h->ostride * d.obs_dim * esz));
        if (d.alias) { h->d_state_alt = alt; h->d_obs_alt = alt; } else { h->d_obs_alt = alt; }
    }
    CREATE_TRY(dalloc(h, &h->d_reward, (size_t)padded));
    CREATE_TRY(dalloc(h, &h->d_done, (size_t)padded));
    CREATE_TRY(dalloc(h, &h->d_tick2, 2));
    // d_actions / d_pack / d_mask (host-boundary staging) are allocated on first use: ensure_staging()
    CREATE_TRY(dalloc(h, &h->d_after_done, (size_t)kShards * kAfterStride));
    CREATE_TRY(dalloc(h, &h->d_bad, 1));
    if (h->n <= kSmallHostPath) {
        const size_t a_bytes = (size_t)padded * 4, o_bytes = (size_t)padded * d.obs_dim * esz, r_bytes = (size_t)padded * 4, d_bytes = (size_t)padded;
        if (hipHostMalloc(&h->hm_block, a_bytes + o_bytes + r_bytes + d_bytes, hipHostMallocMapped) == hipSuccess) {
            char *b = static_cast<char *>(h->hm_block);
            h->hm_actions = b; h->hm_obs = b + a_bytes;
            h->hm_reward = reinterpret_cast<float *>(b + a_bytes + o_bytes); h->hm_done = reinterpret_cast<uint8_t *>(b + a_bytes + o_bytes + r_bytes);
        } else {
            (void)hipGetLastError();
            h->hm_block = nullptr;             // fall back to the memcpy path
        }
    }
    if (cfg->flags & GYMNET_FLAG_RESIDENT) {
        void *blk = nullptr;
        CREATE_HIP(hipHostMalloc(&blk, kMailboxBytes, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(blk, 0, kMailboxBytes);
        h->mb = static_cast<Mailbox *>(blk);
        void *dev = nullptr;
        CREATE_HIP(hipHostGetDevicePointer(&dev, blk, 0));
        h->mb_dev = static_cast<Mailbox *>(dev);
        h->resident = true;
    }
    if (d.has_sbd && !h->autoreset) CREATE_TRY(dalloc(h, &h->d_sbd, (size_t)padded));
    if (cfg->flags & GYMNET_FLAG_FINAL_OBS) CREATE_TRY(dalloc(h, (char **)&h->d_final_obs, (size_t)h->n * d.obs_dim * esz));
    if (cfg->flags & GYMNET_FLAG_DONE_LIST) {
        // segment capacity: the most lanes the waves of one shard can own, for any lane width (4 / 2 / 1 lanes per thread)
        const int64_t w4 = (h->n + 255) / 256, w2 = (h->n + 127) / 128, w1 = (h->n + 63) / 64;
        const int64_t cap4 = (w4 + kShards - 1) / kShards * 256, cap2 = (w2 + kShards - 1) / kShards * 128, cap1 = (w1 + kShards - 1) / kShards * 64;
        h->done_cap = cap4 > cap1 ? cap4 : cap1;
        if (cap2 > h->done_cap) h->done_cap = cap2;
        CREATE_TRY(dalloc(h, &h->d_done_list, (size_t)kShards * (size_t)h->done_cap));
        CREATE_TRY(dalloc(h, &h->d_done_compact, (size_t)padded));
        CREATE_TRY(dalloc(h, &h->d_done_count2, (size_t)2 * kShards * kCountStride));
        CREATE_TRY(dalloc(h, &h->d_done_total, 1));
        CREATE_HIP(hipMemsetAsync(h->d_done_count2, 0, (size_t)2 * kShards * kCountStride * sizeof(uint32_t), h->stream));
        CREATE_HIP(hipMemsetAsync(h->d_done_total, 0, sizeof(uint32_t), h->stream));
    }
    if ((cfg->flags & GYMNET_FLAG_DONE_LIST) && (cfg->flags & GYMNET_FLAG_EPISODE_STATS)) {
        CREATE_TRY(dalloc(h, &h->d_rec_ret, (size_t)kShards * (size_t)h->done_cap));
        CREATE_TRY(dalloc(h, &h->d_rec_len, (size_t)kShards * (size_t)h->done_cap));
    }
    if ((cfg->flags & GYMNET_FLAG_DONE_LIST) && (cfg->flags & GYMNET_FLAG_FINAL_OBS))
        CREATE_TRY(dalloc(h, (char **)&h->d_rec_obs, (size_t)kShards * (size_t)h->done_cap * d.obs_dim * esz));
    if (cfg->flags & GYMNET_FLAG_EPISODE_STATS) {
        CREATE_TRY(dalloc(h, &h->d_ep_ret, (size_t)padded)); CREATE_TRY(dalloc(h, &h->d_ep_len, (size_t)padded));
        CREATE_TRY(dalloc(h, &h->d_fin_ret, (size_t)padded)); CREATE_TRY(dalloc(h, &h->d_fin_len, (size_t)padded));
        CREATE_HIP(hipMemsetAsync(h->d_ep_ret, 0, (size_t)padded * 4, h->stream));
        CREATE_HIP(hipMemsetAsync(h->d_ep_len, 0, (size_t)padded * 4, h->stream));
        CREATE_HIP(hipMemsetAsync(h->d_fin_ret, 0, (size_t)padded * 4, h->stream));
        CREATE_HIP(hipMemsetAsync(h->d_fin_len, 0, (size_t)padded * 4, h->stream));
    }
    recompute_extras(h);
    // defined start: zero state, reward, done; sbd = -1
    CREATE_HIP(hipMemsetAsync(h->d_state, 0, ((size_t)h->sstride * (d.state_dim - 1) + (size_t)h->n) * esz, h->stream));
    if (!d.alias) CREATE_HIP(hipMemsetAsync(h->d_obs, 0, ((size_t)h->ostride * (d.obs_dim - 1) + (size_t)h->n) * esz, h->stream));
    if (h->d_obs_alt) CREATE_HIP(hipMemsetAsync(h->d_obs_alt, 0, ((size_t)h->ostride * (d.obs_dim - 1) + (size_t)h->n) * esz, h->stream));
    CREATE_HIP(hipMemsetAsync(h->d_reward, 0, (size_t)padded * 4, h->stream));
    CREATE_HIP(hipMemsetAsync(h->d_done, 0, (size_t)padded, h->stream));
    CREATE_HIP(hipMemsetAsync(h->d_after_done, 0, (size_t)kShards * kAfterStride * sizeof(unsigned long long), h->stream));
    if (h->d_final_obs) CREATE_HIP(hipMemsetAsync(h->d_final_obs, 0, (size_t)h->n * d.obs_dim * esz, h->stream));
    if (h->d_sbd) CREATE_HIP(launch_fill_i32(h->d_sbd, -1, h->n, h->stream));
    CREATE_TRY(write_tick(h));

    default_policy(h);
#undef CREATE_TRY
#undef CREATE_HIP
    *out = h;
    return GYMNET_OK;
    });
}

int gymnet_vecenv_seed(gymnet_vecenv *h, uint64_t seed) {
    return guarded([&]() -> int {
    ENTER(h);
    return seed_handle(h, seed);
    });
}

int gymnet_vecenv_seed_lanes(gymnet_vecenv *h, const uint64_t *seeds, int64_t count) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!seeds) return fail(h, GYMNET_ERR_INVALID_ARG, "seeds is null");
    if (count != h->n)   // VecEnv.cs:49
        return fail(h, GYMNET_ERR_INVALID_ARG, "Number of seeds passed should be equals to number of environments (%lld != %lld)",
                    (long long)count, (long long)h->n);
    // VecEnv.Seed(int) reaches a VecEnv-typed caller's lanes as N EQUAL seeds (VecEnv.cs:44-46 -> Environments[i].Seed(seed)).
    // With one key for every lane the per-lane-key kernel variant would compute exactly what the lean one does (the Philox
    // counter carries the lane id either way) — only slower, and without the fused rollout.  So an all-equal vector IS Seed(int).
    bool all_equal = true;
    for (int64_t i = 1; i < count && all_equal; ++i) all_equal = seeds[i] == seeds[0];
    if (all_equal) return seed_handle(h, seeds[0]);
    // one allocation, reused by every later Seed(int[]) (it used to grow by N*8 bytes per call); the stream is drained
    // first so no launch still in flight reads the keys being overwritten
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (!h->d_lane_seed_buf) ST_TRY(dalloc(h, &h->d_lane_seed_buf, (size_t)h->n));
    HIP_TRY(h, hipMemcpyAsync(h->d_lane_seed_buf, seeds, (size_t)h->n * 8, hipMemcpyHostToDevice, h->stream));
    h->d_lane_seed = h->d_lane_seed_buf;
    recompute_extras(h);
    h->tick = 0;
    drop_graphs(h);
    return write_tick(h);   // synchronizes: `seeds` is the caller's again on return
    });
}

int gymnet_vecenv_reset_device(gymnet_vecenv *h) {
    return guarded([&]() -> int {
    ENTER(h);
    return launch_reset_lanes(h, nullptr);
    });
}

int gymnet_vecenv_reset_where_device(gymnet_vecenv *h, const uint8_t *d_mask) {
    return guarded([&]() -> int {
    ENTER(h);
    // with AUTORESET the step already re-drew every finished lane (its done flag stays set as the step's RESULT):
    // resetting "the lanes whose done flag is set" again would discard the observation the step returned
    if (!d_mask && h->autoreset) return GYMNET_OK;
    // own done flags: the reset kernel may read its mask from the very array it clears (each lane's flag is read
    // before the same thread clears it), so no snapshot copy is needed
    return launch_reset_lanes(h, d_mask ? d_mask : h->d_done);
    });
}

int gymnet_vecenv_reset(gymnet_vecenv *h, void *obs_out) {
    return guarded([&]() -> int {
    ENTER_KEEP_RESIDENT(h);
    if (resident_serves(h)) return resident_command(h, kMailboxResetAll, nullptr, obs_out, nullptr, nullptr);
    if (h->resident_running) ST_TRY(resident_stop(h));
    ST_TRY(launch_reset_lanes(h, nullptr));
    return copy_out(h, obs_out, nullptr, nullptr);
    });
}

int gymnet_vecenv_reset_where(gymnet_vecenv *h, const uint8_t *mask, void *obs_out) {
    return guarded([&]() -> int {
    ENTER_KEEP_RESIDENT(h);
    if (!mask && !h->autoreset && resident_serves(h)) return resident_command(h, kMailboxResetDone, nullptr, obs_out, nullptr, nullptr);
    if (h->resident_running) ST_TRY(resident_stop(h));
    if (!mask && h->autoreset) return copy_out(h, obs_out, nullptr, nullptr);   // no-op, see gymnet_vecenv_reset_where_device
    if (mask) {
        ST_TRY(ensure_staging(h, false, false, true));
        HIP_TRY(h, hipMemcpyAsync(h->d_mask, mask, (size_t)h->n, hipMemcpyHostToDevice, h->stream));
    }
    ST_TRY(launch_reset_lanes(h, mask ? h->d_mask : h->d_done));
    return copy_out(h, obs_out, nullptr, nullptr);
    });
}

int gymnet_vecenv_host_buffers(gymnet_vecenv *h, void **actions, void **obs, float **reward, uint8_t **done) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!h->pin_block) {
        const EnvDesc &d = *h->desc;
        auto up = [](size_t b) { return (b + 4095) & ~(size_t)4095; };          // every buffer on its own page
        const size_t a_b = up((size_t)h->n * 4), o_b = up((size_t)h->n * d.obs_dim * h->esz), r_b = up((size_t)h->n * 4), d_b = up((size_t)h->n);
        void *blk = nullptr;
        hipError_t e = hipHostMalloc(&blk, a_b + o_b + r_b + d_b, hipHostMallocMapped | hipHostMallocPortable);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(h, GYMNET_ERR_OOM, "hipHostMalloc(%zu bytes, mapped) failed: %s", a_b + o_b + r_b + d_b, hipGetErrorString(e)); }
        std::memset(blk, 0, a_b + o_b + r_b + d_b);
        char *b = static_cast<char *>(blk);
        h->pin_block = blk; h->pin_actions = b; h->pin_obs = b + a_b;
        h->pin_reward = reinterpret_cast<float *>(b + a_b + o_b); h->pin_done = reinterpret_cast<uint8_t *>(b + a_b + o_b + r_b);
    }
    if (actions) *actions = h->pin_actions;
    if (obs) *obs = h->pin_obs;
    if (reward) *reward = h->pin_reward;
    if (done) *done = h->pin_done;
    return GYMNET_OK;
    });
}

// Discrete.Contains over a HOST batch (the resident path validates before it posts: nothing has been staged on the device)
static int validate_host_actions(gymnet_vecenv *h, const void *actions) {
    const EnvDesc &d = *h->desc;
    if (!(h->cfg.flags & GYMNET_FLAG_VALIDATE_ACTIONS) || d.box_action) return GYMNET_OK;
    const int32_t *a = static_cast<const int32_t *>(actions);
    uint32_t bad = 0;
    for (int64_t i = 0; i < h->n; ++i) bad += (a[i] < 0 || a[i] >= d.action_n) ? 1u : 0u;
    if (bad) return fail(h, GYMNET_ERR_INVALID_ACTION, "Action is outside of the configured action space. (%u of %lld lanes, Discrete(%d))", bad, (long long)h->n, d.action_n);
    return GYMNET_OK;
}

int gymnet_vecenv_step(gymnet_vecenv *h, const void *actions, void *obs_out, float *reward_out, uint8_t *done_out) {
    return guarded([&]() -> int {
    ENTER_KEEP_RESIDENT(h);
    if (h->async_pending) return fail(h, GYMNET_ERR_ALREADY_STEPPING, "already running an async step");
    if (resident_serves(h)) {
        if (!actions) return fail(h, GYMNET_ERR_INVALID_ARG, "actions is null");
        ST_TRY(validate_host_actions(h, actions));
        return resident_command(h, kMailboxStep, actions, obs_out, reward_out, done_out);
    }
    const void *d_act = nullptr;
    ST_TRY(stage_host_actions(h, actions, &d_act, true));
    ST_TRY(launch_one_step(h, d_act));
    return copy_out(h, obs_out, reward_out, done_out);
    });
}

int gymnet_vecenv_step_broadcast(gymnet_vecenv *h, int32_t action, void *obs_out, float *reward_out, uint8_t *done_out) {
    return guarded([&]() -> int {
    ENTER_KEEP_RESIDENT(h);
    if (h->async_pending) return fail(h, GYMNET_ERR_ALREADY_STEPPING, "already running an async step");
    const EnvDesc &d = *h->desc;
    if (resident_serves(h)) {
        int32_t bits[kMailboxLanes];
        float f = (float)action;                         // IVecEnv.Step(int) on a Box space: the int is the (scalar) torque
        int32_t word = action;
        if (d.box_action) std::memcpy(&word, &f, 4);
        for (int64_t i = 0; i < h->n; ++i) bits[i] = word;
        ST_TRY(validate_host_actions(h, bits));
        return resident_command(h, kMailboxStep, bits, obs_out, reward_out, done_out);
    }
    ST_TRY(ensure_staging(h, true, false, false));
    if (d.box_action) {   // IVecEnv.Step(int) on a Box space: the int is the (scalar) torque
        float f = (float)action;
        int32_t bits;
        std::memcpy(&bits, &f, 4);
        HIP_TRY(h, launch_fill_i32(static_cast<int32_t *>(h->d_actions), bits, h->n, h->stream));
    } else {
        if ((h->cfg.flags & GYMNET_FLAG_VALIDATE_ACTIONS) && (action < 0 || action >= d.action_n))
            return fail(h, GYMNET_ERR_INVALID_ACTION, "Action is outside of the configured action space. (%d, Discrete(%d))", action, d.action_n);
        HIP_TRY(h, launch_fill_i32(static_cast<int32_t *>(h->d_actions), action, h->n, h->stream));
    }
    ST_TRY(launch_one_step(h, h->d_actions));
    return copy_out(h, obs_out, reward_out, done_out);
    });
}

int gymnet_vecenv_step_async(gymnet_vecenv *h, const void *actions) {
    return guarded([&]() -> int {
    ENTER(h);
    if (h->async_pending) return fail(h, GYMNET_ERR_ALREADY_STEPPING, "already running an async step");
    const void *d_act = nullptr;
    ST_TRY(stage_host_actions(h, actions, &d_act, true));
    ST_TRY(launch_one_step(h, d_act));
    h->async_pending = true;
    return GYMNET_OK;
    });
}

int gymnet_vecenv_step_wait(gymnet_vecenv *h, void *obs_out, float *reward_out, uint8_t *done_out) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!h->async_pending) return fail(h, GYMNET_ERR_NOT_STEPPING, "not running an async step");
    h->async_pending = false;
    return copy_out(h, obs_out, reward_out, done_out);
    });
}

int gymnet_vecenv_read(gymnet_vecenv *h, void *obs_out, float *reward_out, uint8_t *done_out) {
    return guarded([&]() -> int {
    ENTER(h);
    return copy_out(h, obs_out, reward_out, done_out);
    });
}

int gymnet_vecenv_step_device(gymnet_vecenv *h, const void *d_actions) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!d_actions) return fail(h, GYMNET_ERR_INVALID_ARG, "d_actions is null");
    if (h->lcfg.vec > 1 && !aligned_to(d_actions, 4 * h->lcfg.vec))
        return fail(h, GYMNET_ERR_INVALID_ARG, "d_actions must be %d-byte aligned", 4 * h->lcfg.vec);
    ST_TRY(validate_staged_actions(h, d_actions));
    return launch_one_step(h, d_actions);
    });
}

int gymnet_vecenv_rollout_device(gymnet_vecenv *h, const void *d_actions, int64_t steps, int64_t action_stride, int64_t ring) {
    return guarded([&]() -> int {
    ENTER(h);
    return rollout_steps(h, d_actions, steps, action_stride, ring);
    });
}

int gymnet_vecenv_rollout_fused_device(gymnet_vecenv *h, const void *d_actions, int64_t steps, int64_t action_stride,
                                       int64_t ring, const gymnet_rollout_buffers *rec) {
    gymnet_rollout_spec spec{};
    spec.struct_size = sizeof spec;
    spec.action_source = GYMNET_ACTIONS_RING;
    spec.d_actions = d_actions; spec.steps = steps; spec.action_stride = action_stride; spec.ring = ring;
    if (rec) { spec.d_rec_obs = rec->d_obs; spec.d_rec_reward = rec->d_reward; spec.d_rec_done = rec->d_done; }
    return gymnet_vecenv_rollout_fused_ex_device(h, &spec);
}

}  // extern "C"

namespace {

// segment buffers of the fused rollout's episode records: kShards segments of `cap` records each (t, lane, return, length) +
// the shard counters.  Random lanes do not fill the shards evenly, so a segment gets twice its share of the caller's capacity —
// and what a shard still cannot hold (lanes that finish very unevenly) goes to ONE shared overflow segment of `capacity` records
// (round 6): records are dropped only when more episodes end than the caller's arrays hold.
int ensure_episode_segments(gymnet_vecenv *h, int64_t capacity) {
    const int64_t cap = 2 * ((capacity + kShards - 1) / kShards) + 64;
    if (h->d_ep_seg && cap <= h->ep_seg_cap && capacity <= h->ep_ov_cap) return GYMNET_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));          // a previous rollout's gather may still read the old segments
    if (h->d_ep_seg) (void)hipFree(h->d_ep_seg);
    h->d_ep_seg = nullptr; h->ep_seg_cap = 0; h->ep_ov_cap = 0;
    void *q = nullptr;
    // four arrays of [kShards segments of cap | one overflow segment of capacity] records, then kShards + 1 counters (a cache line each)
    const size_t bytes = ((size_t)kShards * (size_t)cap + (size_t)capacity) * 16 + (size_t)(kShards + 1) * kCountStride * 4 + 8;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(h, GYMNET_ERR_OOM, "hipMalloc(%zu bytes) for the rollout's episode records failed: %s", bytes, hipGetErrorString(e));
    h->d_ep_seg = q; h->ep_seg_cap = cap; h->ep_ov_cap = capacity;
    return GYMNET_OK;
}

template <class R>
int rollout_fused_typed(gymnet_vecenv *h, const gymnet_rollout_spec &sp, LaunchCfg cfg, bool episodes) {
    const StepArgsT<R> a = make_step_args<R>(h, sp.d_actions);
    RolloutArgsT<R> r{};
    r.steps = sp.steps; r.action_stride = sp.action_stride; r.ring = sp.ring;
    r.rec_obs = static_cast<R *>(sp.d_rec_obs); r.rec_reward = sp.d_rec_reward; r.rec_done = sp.d_rec_done;
    r.rec_action = sp.d_rec_actions;
    r.action_source = sp.action_source; r.epsilon = sp.epsilon; r.action_seed = sp.action_seed; r.action_tick0 = sp.action_tick0;
    if (episodes) {
        char *seg = static_cast<char *>(h->d_ep_seg);
        const size_t one = ((size_t)kShards * (size_t)h->ep_seg_cap + (size_t)h->ep_ov_cap) * 4;
        r.ep_t = reinterpret_cast<int32_t *>(seg); r.ep_lane = reinterpret_cast<int32_t *>(seg + one);
        r.ep_ret = h->d_ep_ret ? reinterpret_cast<float *>(seg + 2 * one) : nullptr;
        r.ep_len = h->d_ep_ret ? reinterpret_cast<int32_t *>(seg + 3 * one) : nullptr;
        r.ep_count = reinterpret_cast<uint32_t *>(seg + 4 * one);
        r.ep_cap = h->ep_seg_cap;
        r.ov_cap = sp.ep_capacity < h->ep_ov_cap ? sp.ep_capacity : h->ep_ov_cap;
        r.records_no_overflow = (sp.record_flags & GYMNET_RECORDS_NO_OVERFLOW) ? 1 : 0;
        // the kShards + 1 counters, zeroed on the stream by a kernel of our own
        HIP_TRY(h, launch_fill_i32(reinterpret_cast<int32_t *>(r.ep_count), 0, (int64_t)(kShards + 1) * kCountStride, h->stream));
    }
    HIP_TRY(h, launch_rollout_fused(h->cfg.env_id, h->autoreset, h->extras, a, r, cfg, h->stream));
    if (episodes) {
        EpisodeGatherArgs g{};
        g.counts = r.ep_count; g.cap = r.ep_cap;
        g.ep_t = r.ep_t; g.ep_lane = r.ep_lane; g.ep_ret = r.ep_ret; g.ep_len = r.ep_len;
        g.ov_cap = r.ov_cap;
        g.out_t = sp.d_ep_step; g.out_lane = sp.d_ep_lane; g.out_ret = sp.d_ep_return; g.out_len = sp.d_ep_length;
        g.out_capacity = sp.ep_capacity; g.out_count = sp.d_ep_count;
        HIP_TRY(h, launch_gather_episodes(g, h->stream));
    }
    h->last_cparity = a.cparity;          // the done list (if any) describes the rollout's last step
    return GYMNET_OK;
}

}  // namespace

extern "C" {

int gymnet_vecenv_rollout_fused_ex_device(gymnet_vecenv *h, const gymnet_rollout_spec *spec) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!spec) return fail(h, GYMNET_ERR_INVALID_ARG, "spec is null");
    if (spec->struct_size != sizeof(gymnet_rollout_spec))
        return fail(h, GYMNET_ERR_INVALID_ARG, "spec.struct_size %u != %zu (ABI mismatch)", spec->struct_size, sizeof(gymnet_rollout_spec));
    const gymnet_rollout_spec &sp = *spec;
    const EnvDesc &d = *h->desc;
    if (sp.action_source < GYMNET_ACTIONS_RING || sp.action_source > GYMNET_ACTIONS_EPSILON_GREEDY) return fail(h, GYMNET_ERR_INVALID_ARG, "bad action_source %d", sp.action_source);
    const bool ring_read = sp.action_source != GYMNET_ACTIONS_SAMPLE;
    if (ring_read && !sp.d_actions) return fail(h, GYMNET_ERR_INVALID_ARG, "d_actions is null");
    if (sp.steps < 0 || (ring_read && (sp.ring < 1 || sp.action_stride < 0))) return fail(h, GYMNET_ERR_INVALID_ARG, "bad steps/ring/action_stride");
    if (sp.steps > INT32_MAX) return fail(h, GYMNET_ERR_INVALID_ARG, "steps must fit an int32 (episode records carry the step index)");
    if (sp.action_source == GYMNET_ACTIONS_EPSILON_GREEDY) {
        if (d.box_action) return fail(h, GYMNET_ERR_UNSUPPORTED, "epsilon-greedy composition is defined for Discrete action spaces");
        if (!(sp.epsilon >= 0.0f && sp.epsilon <= 1.0f)) return fail(h, GYMNET_ERR_INVALID_ARG, "epsilon must be in [0, 1]");
    }
    if ((h->cfg.flags & GYMNET_FLAG_VALIDATE_ACTIONS) && ring_read)
        return fail(h, GYMNET_ERR_UNSUPPORTED, "VALIDATE_ACTIONS is per step; use gymnet_vecenv_rollout_device (sampled actions are valid by construction)");
    const bool episodes = sp.d_ep_lane || sp.d_ep_step || sp.d_ep_return || sp.d_ep_length || sp.d_ep_count;
    if (episodes) {
        if (!h->extras) return fail(h, GYMNET_ERR_UNSUPPORTED, "episode records need a bookkeeping handle (GYMNET_FLAG_EPISODE_STATS / DONE_LIST / FINAL_OBS)");
        if ((sp.d_ep_return || sp.d_ep_length) && !h->d_ep_ret) return fail(h, GYMNET_ERR_UNSUPPORTED, "episode return / length records need GYMNET_FLAG_EPISODE_STATS");
        if (sp.ep_capacity < 0 || !sp.d_ep_count) return fail(h, GYMNET_ERR_INVALID_ARG, "episode records need ep_capacity >= 0 and d_ep_count");
    }
    if (sp.d_rec_actions && !h->extras && sp.action_source == GYMNET_ACTIONS_RING)
        return fail(h, GYMNET_ERR_UNSUPPORTED, "d_rec_actions: the actions of a plain ring rollout ARE the ring (recorded for sampled / epsilon-greedy actions and on bookkeeping handles)");
    if (sp.steps == 0) {
        if (episodes) HIP_TRY(h, hipMemsetAsync(sp.d_ep_count, 0, 2 * sizeof(uint32_t), h->stream));
        return GYMNET_OK;
    }
    LaunchCfg cfg = h->lcfg;
    // wide accesses on the streams a rollout touches: 16-byte rows of observations (4 floats / 2 doubles per thread), vec rewards /
    // actions, vec done bytes; anything less aligned runs one lane per thread (same bits)
    if (cfg.vec > 1) {
        auto streams_fit = [&](int w) {
            bool ok = (h->n % w) == 0;
            if (ring_read) ok = ok && aligned_to(sp.d_actions, 4 * w) && (sp.action_stride % w) == 0;
            return ok && (!sp.d_rec_obs || aligned16(sp.d_rec_obs)) && (!sp.d_rec_reward || aligned_to(sp.d_rec_reward, 4 * w)) &&
                   (!sp.d_rec_done || aligned_to(sp.d_rec_done, w)) && (!sp.d_rec_actions || aligned_to(sp.d_rec_actions, 4 * w));
        };
        const int w = h->f64 ? 2 : cfg.vec;               // the fused rollout's wide form: 2 doubles / 4 floats (2: Acrobot) per thread
        // the FAT form where every stream allows it — float64: FOUR lanes per thread (the state sits in registers for the whole rollout, so
        // this is about how many lanes share a wave's per-step overheads, not about access width; launch_rollout_env) — else two, else one;
        // (float32 has no fat form: eight floats per thread were measured slower, step_kernels.hpp rollout_fat_lanes)
        if (h->f64 && h->desc->alias && (h->sstride % 4) == 0 && streams_fit(4)) cfg.vec = 4;
        else if (!streams_fit(w)) cfg.vec = 1;
    }
    if (episodes) ST_TRY(ensure_episode_segments(h, sp.ep_capacity));
    ST_TRY(h->f64 ? rollout_fused_typed<double>(h, sp, cfg, episodes) : rollout_fused_typed<float>(h, sp, cfg, episodes));
    swap_buffers(h);                     // DOUBLE_BUFFER: the launch read one buffer and wrote the other, once
    h->tick += (uint64_t)sp.steps;
    h->tslot ^= 1;                       // one launch: it read one half of d_tick2 and wrote the other
    h->step_launches += 1;
    h->lane_steps += (uint64_t)sp.steps * (uint64_t)h->n;
    return GYMNET_OK;
    });
}

int gymnet_vecenv_pack_obs_device(gymnet_vecenv *h, void *d_obs_rowmajor) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!d_obs_rowmajor) return fail(h, GYMNET_ERR_INVALID_ARG, "d_obs_rowmajor is null");
    if (!aligned16(d_obs_rowmajor)) return fail(h, GYMNET_ERR_INVALID_ARG, "d_obs_rowmajor must be 16-byte aligned");
    HIP_TRY(h, pack_current_obs(h, h->d_obs, h->ostride, d_obs_rowmajor));
    return GYMNET_OK;
    });
}

int gymnet_vecenv_sync(gymnet_vecenv *h) {
    return guarded([&]() -> int {
    ENTER(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return GYMNET_OK;
    });
}

int gymnet_vecenv_device_view(gymnet_vecenv *h, gymnet_device_view *out) {
    return guarded([&]() -> int {
    if (!h || !out) return fail(h, GYMNET_ERR_INVALID_ARG, "null argument");
    std::memset(out, 0, sizeof *out);
    out->struct_size = sizeof *out;
    out->state_dim = h->desc->state_dim; out->obs_dim = h->desc->obs_dim; out->obs_aliases_state = h->desc->alias;
    out->num_envs = h->n; out->state_stride = h->sstride; out->obs_stride = h->ostride;
    out->d_state = h->d_state; out->d_obs = h->d_obs;
    out->state_dtype = h->f64 ? GYMNET_DTYPE_F64 : GYMNET_DTYPE_F32;
    out->d_reward = h->d_reward; out->d_done = h->d_done;
    out->d_steps_beyond_done = h->d_sbd; out->d_final_obs = h->d_final_obs; out->d_done_list = h->d_done_compact;
    out->d_episode_return = h->d_ep_ret; out->d_episode_length = h->d_ep_len;
    out->d_finished_return = h->d_fin_ret; out->d_finished_length = h->d_fin_len;
    out->stream = h->stream;
    out->obs_buffer = h->cur;
    out->d_obs_alt = h->d_obs_alt;
    return GYMNET_OK;
    });
}

int gymnet_vecenv_launch_policy(gymnet_vecenv *h, int32_t *vec, int32_t *block, int32_t *nt, int32_t *sequential_lanes) {
    return guarded([&]() -> int {
    if (!h) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "null handle");
    if (vec) *vec = h->lcfg.vec;
    if (block) *block = h->lcfg.block;
    if (nt) *nt = h->lcfg.nt;
    if (sequential_lanes) {      // what the launcher resolves to (lanes, or lane pairs, a thread works through one after another)
        int rvec = 1, rseq = 1;
        resolved_step_shape(h->cfg.env_id, h->f64, h->autoreset, h->extras, h->lcfg, h->n, &rvec, &rseq);
        *sequential_lanes = rseq;
    }
    return GYMNET_OK;
    });
}

int gymnet_vecenv_kernel_name(gymnet_vecenv *h, char *buf, int32_t capacity) {
    return guarded([&]() -> int {
    if (!h || !buf || capacity < 1) return fail(h, GYMNET_ERR_INVALID_ARG, "null handle / buffer");
    if (describe_step_kernel(h->cfg.env_id, h->f64, h->autoreset, h->extras, h->lcfg, h->n, buf, (size_t)capacity) < 0)
        return fail(h, GYMNET_ERR_INVALID_ARG, "unknown env");
    return GYMNET_OK;
    });
}

int gymnet_vecenv_get_state(gymnet_vecenv *h, void *state_soa_v) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!state_soa_v) return fail(h, GYMNET_ERR_INVALID_ARG, "state_soa is null");
    const size_t row = (size_t)h->n * h->esz;
    for (int k = 0; k < h->desc->state_dim; ++k)      // row by row: a row the observation repeats lives in the observation array
        HIP_TRY(h, hipMemcpyAsync(static_cast<char *>(state_soa_v) + (size_t)k * row, state_row(h, k), row, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return GYMNET_OK;
    });
}

int gymnet_vecenv_set_state(gymnet_vecenv *h, const void *state_soa_v) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!state_soa_v) return fail(h, GYMNET_ERR_INVALID_ARG, "state_soa is null");
    const size_t row = (size_t)h->n * h->esz;
    for (int k = 0; k < h->desc->state_dim; ++k)
        HIP_TRY(h, hipMemcpyAsync(state_row(h, k), static_cast<const char *>(state_soa_v) + (size_t)k * row, row, hipMemcpyHostToDevice, h->stream));
    if (!h->desc->alias)      // (float32 envs only: the float64 handle's observation IS its state)
        HIP_TRY(h, launch_observe(h->cfg.env_id, static_cast<const float *>(h->d_state), h->sstride, static_cast<float *>(h->d_obs), h->ostride, h->n, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return GYMNET_OK;
    });
}

int gymnet_vecenv_get_steps_beyond_done(gymnet_vecenv *h, int32_t *out) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!h->d_sbd) return fail(h, GYMNET_ERR_UNSUPPORTED, "steps_beyond_done exists only for CartPole without GYMNET_FLAG_AUTORESET");
    if (!out) return fail(h, GYMNET_ERR_INVALID_ARG, "out is null");
    HIP_TRY(h, hipMemcpyAsync(out, h->d_sbd, (size_t)h->n * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return GYMNET_OK;
    });
}

int gymnet_vecenv_set_steps_beyond_done(gymnet_vecenv *h, const int32_t *in) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!h->d_sbd) return fail(h, GYMNET_ERR_UNSUPPORTED, "steps_beyond_done exists only for CartPole without GYMNET_FLAG_AUTORESET");
    if (!in) return fail(h, GYMNET_ERR_INVALID_ARG, "in is null");
    HIP_TRY(h, hipMemcpyAsync(h->d_sbd, in, (size_t)h->n * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return GYMNET_OK;
    });
}

int gymnet_vecenv_get_tick(gymnet_vecenv *h, uint64_t *tick) {
    return guarded([&]() -> int {
    if (!h || !tick) return fail(h, GYMNET_ERR_INVALID_ARG, "null argument");
    *tick = h->tick;
    return GYMNET_OK;
    });
}

int gymnet_vecenv_set_tick(gymnet_vecenv *h, uint64_t tick) {
    return guarded([&]() -> int {
    ENTER(h);
    h->tick = tick;
    return write_tick(h);
    });
}

int gymnet_vecenv_counters(gymnet_vecenv *h, gymnet_counters *out) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!out) return fail(h, GYMNET_ERR_INVALID_ARG, "out is null");
    std::memset(out, 0, sizeof *out);
    out->struct_size = sizeof *out;
    std::vector<unsigned long long> adv((size_t)kShards * kAfterStride);
    HIP_TRY(h, hipMemcpyAsync(adv.data(), h->d_after_done, adv.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    std::vector<uint32_t> cntv;
    if (h->d_done_count2 && h->last_cparity >= 0) {
        cntv.resize((size_t)kShards * kCountStride);
        HIP_TRY(h, hipMemcpyAsync(cntv.data(), h->d_done_count2 + (size_t)h->last_cparity * kShards * kCountStride,
                                  cntv.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    unsigned long long ad = 0;
    for (int s = 0; s < kShards; ++s) ad += adv[(size_t)s * kAfterStride];
    int64_t last_done = -1;
    if (!cntv.empty()) { last_done = 0; for (int s = 0; s < kShards; ++s) last_done += cntv[(size_t)s * kCountStride]; }
    uint64_t dtick[2] = {0, 0};
    HIP_TRY(h, hipMemcpy(dtick, h->d_tick2, sizeof dtick, hipMemcpyDeviceToHost));
    out->tick = dtick[h->tslot];   // the device's own count (== host mirror h->tick)
    out->lane_steps = h->lane_steps;
    out->stepped_after_done = ad;
    out->last_done_count = last_done;
    return GYMNET_OK;
    });
}

int gymnet_vecenv_done_lanes(gymnet_vecenv *h, int32_t *lanes_out, int64_t capacity, int64_t *count) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!h->d_done_list) return fail(h, GYMNET_ERR_UNSUPPORTED, "needs GYMNET_FLAG_DONE_LIST");
    if (!count || capacity < 0 || (capacity > 0 && !lanes_out)) return fail(h, GYMNET_ERR_INVALID_ARG, "bad count/capacity/lanes_out");
    if (h->last_cparity < 0) { *count = 0; return GYMNET_OK; }
    ST_TRY(compact_done(h, h->d_done_compact, nullptr, nullptr, nullptr, h->padded, h->d_done_total));
    uint32_t c32 = 0;
    HIP_TRY(h, hipMemcpyAsync(&c32, h->d_done_total, sizeof c32, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int64_t c = (int64_t)c32;
    *count = c;
    const int64_t m = c < capacity ? c : capacity;
    if (m > 0) HIP_TRY(h, hipMemcpy(lanes_out, h->d_done_compact, (size_t)m * 4, hipMemcpyDeviceToHost));
    return GYMNET_OK;
    });
}

int gymnet_vecenv_done_lanes_device(gymnet_vecenv *h, int32_t *d_lanes_out, uint32_t *d_count_out) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!h->d_done_list) return fail(h, GYMNET_ERR_UNSUPPORTED, "needs GYMNET_FLAG_DONE_LIST");
    if (!d_count_out) return fail(h, GYMNET_ERR_INVALID_ARG, "d_count_out is null");
    if (h->last_cparity < 0) { HIP_TRY(h, hipMemsetAsync(d_count_out, 0, sizeof(uint32_t), h->stream)); return GYMNET_OK; }
    return compact_done(h, d_lanes_out ? d_lanes_out : h->d_done_compact, nullptr, nullptr, nullptr, h->padded, d_count_out);
    });
}

int gymnet_vecenv_done_records_device(gymnet_vecenv *h, int32_t *d_lanes, float *d_return, int32_t *d_length, void *d_final_obs,
                                      int64_t capacity, uint32_t *d_count) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!h->d_done_list) return fail(h, GYMNET_ERR_UNSUPPORTED, "needs GYMNET_FLAG_DONE_LIST");
    if (capacity < 0) return fail(h, GYMNET_ERR_INVALID_ARG, "capacity < 0");
    if ((d_return || d_length) && !h->d_rec_ret) return fail(h, GYMNET_ERR_UNSUPPORTED, "episode records need GYMNET_FLAG_EPISODE_STATS");
    if (d_final_obs && !h->d_rec_obs) return fail(h, GYMNET_ERR_UNSUPPORTED, "terminal observations need GYMNET_FLAG_FINAL_OBS");
    if (h->last_cparity < 0) { if (d_count) HIP_TRY(h, hipMemsetAsync(d_count, 0, sizeof(uint32_t), h->stream)); return GYMNET_OK; }
    return compact_done(h, d_lanes, d_return, d_length, d_final_obs, capacity, d_count);
    });
}

int gymnet_vecenv_done_records(gymnet_vecenv *h, int32_t *lanes_out, float *return_out, int32_t *length_out, void *final_obs_out,
                               int64_t capacity, int64_t *count) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!h->d_done_list) return fail(h, GYMNET_ERR_UNSUPPORTED, "needs GYMNET_FLAG_DONE_LIST");
    if (!count || capacity < 0) return fail(h, GYMNET_ERR_INVALID_ARG, "bad count/capacity");
    if ((return_out || length_out) && !h->d_rec_ret) return fail(h, GYMNET_ERR_UNSUPPORTED, "episode records need GYMNET_FLAG_EPISODE_STATS");
    if (final_obs_out && !h->d_rec_obs) return fail(h, GYMNET_ERR_UNSUPPORTED, "terminal observations need GYMNET_FLAG_FINAL_OBS");
    if (h->last_cparity < 0) { *count = 0; return GYMNET_OK; }
    const int O = h->desc->obs_dim;
    if ((return_out || length_out) && !h->d_rec_ret_c) { ST_TRY(dalloc(h, &h->d_rec_ret_c, (size_t)h->padded)); ST_TRY(dalloc(h, &h->d_rec_len_c, (size_t)h->padded)); }
    if (final_obs_out && !h->d_rec_obs_c) ST_TRY(dalloc(h, (char **)&h->d_rec_obs_c, (size_t)h->padded * O * h->esz));
    ST_TRY(compact_done(h, h->d_done_compact, (return_out || length_out) ? h->d_rec_ret_c : nullptr, (return_out || length_out) ? h->d_rec_len_c : nullptr,
                        final_obs_out ? h->d_rec_obs_c : nullptr, h->padded, h->d_done_total));
    uint32_t c32 = 0;
    HIP_TRY(h, hipMemcpyAsync(&c32, h->d_done_total, sizeof c32, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *count = (int64_t)c32;
    const size_t m = (size_t)((int64_t)c32 < capacity ? (int64_t)c32 : capacity);
    if (m > 0) {
        if (lanes_out) HIP_TRY(h, hipMemcpyAsync(lanes_out, h->d_done_compact, m * 4, hipMemcpyDeviceToHost, h->stream));
        if (return_out) HIP_TRY(h, hipMemcpyAsync(return_out, h->d_rec_ret_c, m * 4, hipMemcpyDeviceToHost, h->stream));
        if (length_out) HIP_TRY(h, hipMemcpyAsync(length_out, h->d_rec_len_c, m * 4, hipMemcpyDeviceToHost, h->stream));
        if (final_obs_out) HIP_TRY(h, hipMemcpyAsync(final_obs_out, h->d_rec_obs_c, m * O * h->esz, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return GYMNET_OK;
    });
}

int gymnet_vecenv_episode_stats(gymnet_vecenv *h, float *finished_return, int32_t *finished_length) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!h->d_fin_ret) return fail(h, GYMNET_ERR_UNSUPPORTED, "needs GYMNET_FLAG_EPISODE_STATS");
    // GYMNET_FLAG_COMPACT_RECORDS_ONLY: the step wrote compact records only — bring the dense view up to date for the most recent step
    if (h->compact_only && h->d_rec_ret && h->last_cparity >= 0) ST_TRY(compact_done(h, nullptr, nullptr, nullptr, nullptr, 0, nullptr, true));
    if (finished_return) HIP_TRY(h, hipMemcpyAsync(finished_return, h->d_fin_ret, (size_t)h->n * 4, hipMemcpyDeviceToHost, h->stream));
    if (finished_length) HIP_TRY(h, hipMemcpyAsync(finished_length, h->d_fin_len, (size_t)h->n * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return GYMNET_OK;
    });
}

int gymnet_vecenv_final_obs(gymnet_vecenv *h, void *final_obs_out) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!h->d_final_obs) return fail(h, GYMNET_ERR_UNSUPPORTED, "needs GYMNET_FLAG_FINAL_OBS");
    if (!final_obs_out) return fail(h, GYMNET_ERR_INVALID_ARG, "final_obs_out is null");
    if (h->compact_only && h->d_rec_obs && h->last_cparity >= 0) ST_TRY(compact_done(h, nullptr, nullptr, nullptr, nullptr, 0, nullptr, true));
    ST_TRY(ensure_staging(h, false, true, false));
    HIP_TRY(h, pack_current_obs(h, h->d_final_obs, h->n, h->d_pack));       // dense terminal observations [O][n] -> row-major
    HIP_TRY(h, hipMemcpyAsync(final_obs_out, h->d_pack, (size_t)h->n * h->desc->obs_dim * h->esz, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return GYMNET_OK;
    });
}

int gymnet_sample_discrete_device(int device, void *stream, int32_t *d_out, int64_t count, int32_t n, int32_t start,
                                  uint64_t seed, uint64_t lane_offset, uint64_t tick) {
    return guarded([&]() -> int {
    if (!d_out || count < 0 || n <= 0) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "bad d_out/count/n");
    DeviceScope dev_scope;
    HIP_TRY(nullptr, hipSetDevice(device));
    HIP_TRY(nullptr, launch_sample_discrete(d_out, count, n, start, seed, lane_offset, tick, static_cast<hipStream_t>(stream)));
    return GYMNET_OK;
    });
}

int gymnet_sample_discrete_masked_device(int device, void *stream, int32_t *d_out, int64_t count, int32_t n, int32_t start,
                                         const uint8_t *d_mask, int64_t mask_stride, uint64_t seed, uint64_t lane_offset, uint64_t tick) {
    return guarded([&]() -> int {
    if (!d_out || count < 0 || n <= 0) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "bad d_out/count/n");
    if (!d_mask) return gymnet_sample_discrete_device(device, stream, d_out, count, n, start, seed, lane_offset, tick);   // Discrete.cs:27
    if (mask_stride != 0 && mask_stride < n) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "mask_stride must be 0 (shared row) or >= n");
    DeviceScope dev_scope;
    HIP_TRY(nullptr, hipSetDevice(device));
    HIP_TRY(nullptr, launch_sample_discrete_masked(d_out, count, n, start, d_mask, mask_stride, seed, lane_offset, tick, static_cast<hipStream_t>(stream)));
    return GYMNET_OK;
    });
}

int gymnet_vecenv_sample_actions_masked_device(gymnet_vecenv *h, int32_t *d_actions, const uint8_t *d_mask, int64_t mask_stride,
                                               uint64_t seed, uint64_t tick) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!d_actions) return fail(h, GYMNET_ERR_INVALID_ARG, "d_actions is null");
    const EnvDesc &d = *h->desc;
    if (d.box_action) return fail(h, GYMNET_ERR_UNSUPPORTED, "Box.sample cannot be provided a mask.");   // Box.cs:70
    if (d_mask && mask_stride != 0 && mask_stride < d.action_n) return fail(h, GYMNET_ERR_INVALID_ARG, "mask_stride must be 0 (shared row) or >= n");
    if (!d_mask)
        HIP_TRY(h, launch_sample_discrete(d_actions, h->n, d.action_n, 0, seed, (uint64_t)h->cfg.lane_offset, tick, h->stream));
    else
        HIP_TRY(h, launch_sample_discrete_masked(d_actions, h->n, d.action_n, 0, d_mask, mask_stride, seed,
                                                 (uint64_t)h->cfg.lane_offset, tick, h->stream));
    return GYMNET_OK;
    });
}

int gymnet_sample_box_device(int device, void *stream, float *d_out, int64_t count, float low, float high,
                             uint64_t seed, uint64_t lane_offset, uint64_t tick) {
    return guarded([&]() -> int {
    if (!d_out || count < 0 || !(low <= high)) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "bad d_out/count/bounds");
    DeviceScope dev_scope;
    HIP_TRY(nullptr, hipSetDevice(device));
    HIP_TRY(nullptr, launch_sample_box(d_out, count, low, high, seed, lane_offset, tick, static_cast<hipStream_t>(stream)));
    return GYMNET_OK;
    });
}

int gymnet_sample_box_elementwise_device(int device, void *stream, float *d_out, int64_t count, int32_t dim, const float *d_low,
                                         const float *d_high, uint64_t seed, uint64_t lane_offset, uint64_t tick) {
    return guarded([&]() -> int {
    if (!d_out || !d_low || !d_high || count < 0 || dim <= 0) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "bad d_out/d_low/d_high/count/dim");
    DeviceScope dev_scope;
    HIP_TRY(nullptr, hipSetDevice(device));
    HIP_TRY(nullptr, launch_sample_box_elementwise(d_out, count, dim, d_low, d_high, seed, lane_offset, tick, static_cast<hipStream_t>(stream)));
    return GYMNET_OK;
    });
}

int gymnet_vecenv_sample_actions_device(gymnet_vecenv *h, void *d_actions, uint64_t seed, uint64_t tick) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!d_actions) return fail(h, GYMNET_ERR_INVALID_ARG, "d_actions is null");
    const EnvDesc &d = *h->desc;
    if (d.box_action)
        HIP_TRY(h, launch_sample_box(static_cast<float *>(d_actions), h->n, d.action_low, d.action_high, seed,
                                     (uint64_t)h->cfg.lane_offset, tick, h->stream));
    else
        HIP_TRY(h, launch_sample_discrete(static_cast<int32_t *>(d_actions), h->n, d.action_n, 0, seed,
                                          (uint64_t)h->cfg.lane_offset, tick, h->stream));
    return GYMNET_OK;
    });
}

int gymnet_vecenv_compose_actions_device(gymnet_vecenv *h, const int32_t *d_policy_actions, float epsilon, int32_t *d_actions_out,
                                         uint64_t seed, uint64_t tick) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!d_policy_actions || !d_actions_out) return fail(h, GYMNET_ERR_INVALID_ARG, "null action buffer");
    if (h->desc->box_action) return fail(h, GYMNET_ERR_UNSUPPORTED, "epsilon-greedy composition is defined for Discrete action spaces");
    if (!(epsilon >= 0.0f && epsilon <= 1.0f)) return fail(h, GYMNET_ERR_INVALID_ARG, "epsilon must be in [0, 1]");
    HIP_TRY(h, launch_compose_discrete(d_policy_actions, d_actions_out, h->n, h->desc->action_n, epsilon, seed,
                                       (uint64_t)h->cfg.lane_offset, tick, h->stream));
    return GYMNET_OK;
    });
}

int gymnet_vecenv_sample_actions(gymnet_vecenv *h, void *actions_out, uint64_t seed, uint64_t tick) {
    return guarded([&]() -> int {
    if (!h) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "null handle");
    if (!actions_out) return fail(h, GYMNET_ERR_INVALID_ARG, "actions_out is null");
    {
        ENTER(h);
        ST_TRY(ensure_staging(h, true, false, false));
    }
    ST_TRY(gymnet_vecenv_sample_actions_device(h, h->d_actions, seed, tick));
    ENTER(h);
    HIP_TRY(h, hipMemcpyAsync(actions_out, h->d_actions, (size_t)h->n * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return GYMNET_OK;
    });
}

int gymnet_vecenv_set_launch_policy(gymnet_vecenv *h, const gymnet_launch_policy *p) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!p) return fail(h, GYMNET_ERR_INVALID_ARG, "policy is null");
    if (p->struct_size != sizeof(gymnet_launch_policy))
        return fail(h, GYMNET_ERR_INVALID_ARG, "policy.struct_size %u != %zu (ABI mismatch)", p->struct_size, sizeof(gymnet_launch_policy));
    HIP_TRY(h, hipStreamSynchronize(h->stream));     // no launch of the old configuration is still being captured / replayed
    return apply_policy(h, *p, /*strict=*/true);
    });
}

int gymnet_vecenv_get_launch_policy(gymnet_vecenv *h, gymnet_launch_policy *out) {
    return guarded([&]() -> int {
    if (!h || !out) return fail(h, GYMNET_ERR_INVALID_ARG, "null argument");
    out->struct_size = sizeof *out;
    out->vec = h->lcfg.vec; out->block = h->lcfg.block; out->nt = h->lcfg.nt; out->sequential_lanes = h->lcfg.items;
    out->reset_form = h->lcfg.reset_form; out->lds_pipe = h->lcfg.lds_pipe; out->occupancy_lds_bytes = h->lcfg.lds_bytes;
    out->graph = h->graph_mode;
    return GYMNET_OK;
    });
}

int gymnet_vecenv_get_seed(gymnet_vecenv *h, uint64_t *seed, int32_t *per_lane) {
    return guarded([&]() -> int {
    if (!h) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "null handle");
    if (seed) *seed = h->seed;
    if (per_lane) *per_lane = h->d_lane_seed ? 1 : 0;
    return GYMNET_OK;
    });
}

}  // extern "C"

namespace {
// the per-lane array behind a gymnet_array_id: device pointer, bytes, row stride / rows for the one 2-D member
struct ArrayRef { void *p; size_t row_bytes; int rows; size_t pitch; const char *needs; };
ArrayRef array_ref(gymnet_vecenv *h, int which) {
    const size_t n = (size_t)h->n;
    switch (which) {
        case GYMNET_ARRAY_REWARD: return {h->d_reward, n * 4, 1, 0, nullptr};
        case GYMNET_ARRAY_DONE: return {h->d_done, n, 1, 0, nullptr};
        case GYMNET_ARRAY_STEPS_BEYOND_DONE: return {h->d_sbd, n * 4, 1, 0, "CartPole without GYMNET_FLAG_AUTORESET"};
        case GYMNET_ARRAY_EPISODE_RETURN: return {h->d_ep_ret, n * 4, 1, 0, "GYMNET_FLAG_EPISODE_STATS"};
        case GYMNET_ARRAY_EPISODE_LENGTH: return {h->d_ep_len, n * 4, 1, 0, "GYMNET_FLAG_EPISODE_STATS"};
        case GYMNET_ARRAY_FINISHED_RETURN: return {h->d_fin_ret, n * 4, 1, 0, "GYMNET_FLAG_EPISODE_STATS"};
        case GYMNET_ARRAY_FINISHED_LENGTH: return {h->d_fin_len, n * 4, 1, 0, "GYMNET_FLAG_EPISODE_STATS"};
        case GYMNET_ARRAY_FINAL_OBS: return {h->d_final_obs, n * h->esz, h->desc->obs_dim, n * h->esz, "GYMNET_FLAG_FINAL_OBS"};
        case GYMNET_ARRAY_LANE_SEEDS: return {h->d_lane_seed, n * 8, 1, 0, "per-lane seeds (gymnet_vecenv_seed_lanes)"};
        default: return {nullptr, 0, 0, 0, "a gymnet_array_id"};
    }
}
}  // namespace

extern "C" {

int gymnet_vecenv_get_array(gymnet_vecenv *h, int32_t which, void *out, int64_t bytes) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!out) return fail(h, GYMNET_ERR_INVALID_ARG, "out is null");
    // with compact records only, the dense "last finished episode" arrays are brought up to date first (as their getters do)
    if (h->compact_only && h->last_cparity >= 0 &&
        (((which == GYMNET_ARRAY_FINISHED_RETURN || which == GYMNET_ARRAY_FINISHED_LENGTH) && h->d_rec_ret) || (which == GYMNET_ARRAY_FINAL_OBS && h->d_rec_obs)))
        ST_TRY(compact_done(h, nullptr, nullptr, nullptr, nullptr, 0, nullptr, true));
    const ArrayRef a = array_ref(h, which);
    if (!a.p) return fail(h, which < 0 || which > GYMNET_ARRAY_LANE_SEEDS ? GYMNET_ERR_INVALID_ARG : GYMNET_ERR_UNSUPPORTED, "array %d needs %s", which, a.needs);
    if (bytes != (int64_t)(a.row_bytes * a.rows)) return fail(h, GYMNET_ERR_INVALID_ARG, "array %d holds %zu bytes, caller passed %lld", which, a.row_bytes * a.rows, (long long)bytes);
    HIP_TRY(h, hipMemcpyAsync(out, a.p, a.row_bytes * a.rows, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return GYMNET_OK;
    });
}

int gymnet_vecenv_set_array(gymnet_vecenv *h, int32_t which, const void *in, int64_t bytes) {
    return guarded([&]() -> int {
    ENTER(h);
    if (!in) return fail(h, GYMNET_ERR_INVALID_ARG, "in is null");
    if (which == GYMNET_ARRAY_LANE_SEEDS && !h->d_lane_seed) {
        // installs per-lane keys WITHOUT touching the engine tick (gymnet_vecenv_seed_lanes rewinds it): checkpoint restore
        if (bytes != h->n * 8) return fail(h, GYMNET_ERR_INVALID_ARG, "array %d holds %lld bytes, caller passed %lld", which, (long long)h->n * 8, (long long)bytes);
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (!h->d_lane_seed_buf) ST_TRY(dalloc(h, &h->d_lane_seed_buf, (size_t)h->n));
        h->d_lane_seed = h->d_lane_seed_buf;
        recompute_extras(h);
        drop_graphs(h);
    }
    const ArrayRef a = array_ref(h, which);
    if (!a.p) return fail(h, which < 0 || which > GYMNET_ARRAY_LANE_SEEDS ? GYMNET_ERR_INVALID_ARG : GYMNET_ERR_UNSUPPORTED, "array %d needs %s", which, a.needs);
    if (bytes != (int64_t)(a.row_bytes * a.rows)) return fail(h, GYMNET_ERR_INVALID_ARG, "array %d holds %zu bytes, caller passed %lld", which, a.row_bytes * a.rows, (long long)bytes);
    HIP_TRY(h, hipMemcpyAsync(a.p, in, a.row_bytes * a.rows, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    // the compacted done list / records describe the step that produced the PREVIOUS flags: not part of a restored state
    if (which == GYMNET_ARRAY_DONE) h->last_cparity = -1;
    return GYMNET_OK;
    });
}

}  // extern "C"
