// handle.hpp — what a gymnet_vecenv handle IS, plus the host-side helpers capi.hip and group.hip share.
// Internal to the library: nothing here crosses the C ABI (include/gymnet_amd.h is the boundary).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/gymnet_amd.h"
#include "kernels.hpp"

namespace gymnet {

struct EnvDesc {
    const char *name;
    int state_dim, obs_dim;
    bool alias, box_action, has_sbd;
    int action_n;
    float action_low, action_high;
    float obs_low[8], obs_high[8];
    float reward_low, reward_high;
    int algorithmic_bytes;
    int traffic_bytes;          // bytes one env-step really moves (< algorithmic where a state row is stored once, in the observation)
    int state_row_in_obs[4];    // state row k lives in observation row state_row_in_obs[k] (-1: its own row of the state array)
};
extern const EnvDesc kEnvs[4];

struct GraphEntry {
    const void *actions;
    int64_t len, stride, ring;
    int parity, cparity, cur;
    hipGraph_t graph;
    hipGraphExec_t exec;
    uint64_t last_use;
};
constexpr size_t kMaxGraphs = 8;       // per handle; least-recently-used entry is destroyed beyond this

}  // namespace gymnet

struct gymnet_vecenv {
    gymnet_config cfg{};
    const gymnet::EnvDesc *desc = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t n = 0, padded = 0, sstride = 0, ostride = 0;
    bool autoreset = false, extras = false;
    // The STATE SCALAR of the handle: float, or double with GYMNET_FLAG_F64 (CartPole only: the reference's own arithmetic,
    // CartPoleEnv.cs:141-166,185).  Every buffer that holds state or observation values — d_state / d_obs and their double-buffer
    // twins, the row-major staging, terminal observations and their compact records, the host-mapped observation staging — holds
    // elements of that type; they are kept as void* here and typed where they are used (esz = element size in bytes).
    bool f64 = false;
    size_t esz = 4;
    // d_state / d_obs always point at the CURRENT (most recently written) buffers; with GYMNET_FLAG_DOUBLE_BUFFER
    // d_state_alt / d_obs_alt are what the next step writes, and the pairs swap after every step launch.
    void *d_state = nullptr, *d_obs = nullptr;
    void *d_state_alt = nullptr, *d_obs_alt = nullptr;
    bool double_buffer = false;
    int cur = 0;                   // index of the buffer d_obs points at (0 = the one reset first wrote)
    float *d_reward = nullptr;
    uint8_t *d_done = nullptr, *d_mask = nullptr;
    int32_t *d_sbd = nullptr;
    uint64_t *d_tick2 = nullptr;
    void *d_actions = nullptr;     // staging for host-path / broadcast actions   (allocated on first use)
    void *d_pack = nullptr;        // row-major obs staging                       (allocated on first use)
    void *d_final_obs = nullptr;
    int32_t *d_done_list = nullptr, *d_done_compact = nullptr;    // sharded segments / compact list (on demand)
    uint32_t *d_done_count2 = nullptr, *d_done_total = nullptr;
    int64_t done_cap = 0;
    float *d_ep_ret = nullptr, *d_fin_ret = nullptr;
    int32_t *d_ep_len = nullptr, *d_fin_len = nullptr;
    // compact per-step records beside the sharded done list (DONE_LIST + EPISODE_STATS / FINAL_OBS), and their gathered copies
    float *d_rec_ret = nullptr, *d_rec_ret_c = nullptr;
    void *d_rec_obs = nullptr, *d_rec_obs_c = nullptr;
    int32_t *d_rec_len = nullptr, *d_rec_len_c = nullptr;
    uint64_t *d_lane_seed = nullptr;       // active per-lane keys (NULL = one key for all lanes)
    uint64_t *d_lane_seed_buf = nullptr;   // the one allocation Seed(int[]) reuses
    unsigned long long *d_after_done = nullptr;
    // small batches (n <= kSmallHostPath): host-mapped staging, so a host-boundary step is 2 kernel launches + 1 sync
    void *hm_actions = nullptr;    // pinned + mapped: the step kernel reads the actions straight from it
    void *hm_obs = nullptr;
    float *hm_reward = nullptr;
    uint8_t *hm_done = nullptr;
    void *hm_block = nullptr;
    // gymnet_vecenv_host_buffers: library-owned page-locked, device-mapped host buffers for the host-boundary path (any batch
    // size); the step's export kernel writes results straight into them, actions are DMA'd out of them
    void *pin_block = nullptr, *pin_actions = nullptr;
    void *pin_obs = nullptr;
    float *pin_reward = nullptr;
    uint8_t *pin_done = nullptr;
    uint32_t *d_bad = nullptr;
    // GYMNET_FLAG_RESIDENT (num_envs <= 64): the mailbox in coherent host memory and the state of the resident kernel
    gymnet::Mailbox *mb = nullptr;         // host address (page-locked, device-mapped, coherent)
    gymnet::Mailbox *mb_dev = nullptr;     // the same memory as the device sees it
    bool resident = false;                 // the handle serves host-boundary steps / resets through the resident kernel
    bool resident_running = false;         // a resident kernel is (or may still be) on the stream
    uint64_t mb_seq = 0;                   // sequence number of the last command posted
    void *d_ep_seg = nullptr;      // fused rollout: segmented episode records + shard counters (allocated on first use, grown on demand)
    int64_t ep_seg_cap = 0;        // records per shard segment
    int64_t ep_ov_cap = 0;         // records of the shared overflow segment (= the largest ep_capacity asked for so far)
    uint64_t seed = 0, tick = 0, lane_steps = 0, step_launches = 0;
    int tslot = 0;                 // which half of d_tick2 the NEXT launch reads (it writes the other half)
    int last_cparity = -1;
    bool async_pending = false;
    std::atomic<bool> busy{false};
    int simds = 1024;              // SIMD units of the device (multiProcessorCount x 4), read at create
    gymnet::LaunchCfg lcfg{4, 256, 0, 0, 1};
    int graph_mode = -1;           // gymnet_launch_policy.graph: -1 = by batch size, 0 = eager launches, 1 = hipGraph replay
    bool can_vec4 = false, can_vec2 = false, lds_ok = false;   // what the buffers' alignment / the batch size allow (set at create)
    bool compact_only = false;     // GYMNET_FLAG_COMPACT_RECORDS_ONLY
    std::vector<gymnet::GraphEntry> graphs;
    uint64_t graph_clock = 0;
    std::vector<void *> owned;     // device allocations to free
    std::string err;
};

namespace gymnet {

// sets the thread-local last-error message (and the handle's), returns `status`
int fail(gymnet_vecenv *h, int status, const char *fmt, ...);
void set_last_error(const char *msg);

// Restores the caller's current HIP device when it goes out of scope (a process that holds handles on several GPUs —
// or shares the runtime with torch — must not find its current device changed by a library call).
struct DeviceScope {
    int prev = -1;
    DeviceScope() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct BusyGuard {
    gymnet_vecenv *h;
    bool ok;
    explicit BusyGuard(gymnet_vecenv *hh) : h(hh), ok(false) {
        bool expect = false;
        ok = h->busy.compare_exchange_strong(expect, true);
    }
    ~BusyGuard() { if (ok) h->busy.store(false); }
};

#define HIP_TRY(h, expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return ::gymnet::fail(h, e_ == hipErrorOutOfMemory ? GYMNET_ERR_OOM : GYMNET_ERR_HIP, "%s failed: %s", #expr, \
                                  hipGetErrorString(e_));                                             \
    } while (0)

#define ST_TRY(expr)                  \
    do {                              \
        int s_ = (expr);              \
        if (s_ != GYMNET_OK) return s_; \
    } while (0)

// Entry-point prologue: null check, single-caller guard, switch to the handle's device (restored on return).
// ENTER_KEEP_RESIDENT: the entry points the resident kernel itself serves (host-boundary step / reset).  ENTER: everything else —
// a running resident kernel is told to leave first (it owns the stream and the tick until it has), see resident_stop.
#define ENTER_KEEP_RESIDENT(h)                                                                                     \
    if (!(h)) return ::gymnet::fail(nullptr, GYMNET_ERR_INVALID_ARG, "null handle");                                \
    ::gymnet::BusyGuard guard_(h);                                                                                 \
    if (!guard_.ok) return ::gymnet::fail(h, GYMNET_ERR_ALREADY_STEPPING, "handle is in use by another call");      \
    ::gymnet::DeviceScope dev_scope_;                                                                              \
    (void)hipGetLastError();   /* a stale error left on this thread by anyone must not fail this call's launches */ \
    HIP_TRY(h, hipSetDevice((h)->device))
#define ENTER(h)                                                                                                   \
    ENTER_KEEP_RESIDENT(h);                                                                                        \
    if ((h)->resident_running) { int rs_ = ::gymnet::resident_stop(h); if (rs_ != GYMNET_OK) return rs_; }

// Nothing may throw across the C ABI: every entry point body runs inside this guard.
template <class F>
int guarded(F &&f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        set_last_error("host allocation failed (std::bad_alloc)");
        return GYMNET_ERR_OOM;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return GYMNET_ERR_HIP;
    } catch (...) {
        set_last_error("unexpected C++ exception inside the library");
        return GYMNET_ERR_HIP;
    }
}

// address of row k of a [rows][stride] structure-of-arrays buffer of esz-byte elements
inline void *row_at(void *base, int64_t k, int64_t stride, size_t esz) { return static_cast<char *>(base) + (size_t)k * (size_t)stride * esz; }
// bytes one env-step of this handle moves by the algorithmic count: 41 for float32 CartPole, 73 in float64
// (32 B state read + 4 B action + 32 B state written + 4 B reward + 1 B done)
inline size_t bytes_per_step(const gymnet_vecenv *h) { return h->f64 ? (size_t)73 : (size_t)h->desc->algorithmic_bytes; }

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned_to(const void *p, int bytes) { return (reinterpret_cast<uintptr_t>(p) & (uintptr_t)(bytes - 1)) == 0; }

// ---- helpers that assume the caller already ENTERed the handle (device set, busy flag held) ----------------------
int launch_one_step(gymnet_vecenv *h, const void *d_actions);       // one vector step = one kernel launch
int launch_reset_lanes(gymnet_vecenv *h, const uint8_t *d_mask);    // NULL = all lanes
int stage_host_actions(gymnet_vecenv *h, const void *actions, const void **d_use, bool validate_now);
int validate_staged_actions(gymnet_vecenv *h, const void *d_actions);
// obs_out: float32 [n, obs_dim], or float64 for a GYMNET_FLAG_F64 handle
int queue_copy_out(gymnet_vecenv *h, void *obs_out, float *reward_out, uint8_t *done_out);   // no final sync (large-batch path)
int copy_out(gymnet_vecenv *h, void *obs_out, float *reward_out, uint8_t *done_out);         // blocks
// graph_mode: -1 = by batch size (replay only while launch-bound), 0 = eager launches, 1 = always replay a captured graph
int rollout_steps(gymnet_vecenv *h, const void *d_actions, int64_t steps, int64_t action_stride, int64_t ring, int graph_mode = -1);
int write_tick(gymnet_vecenv *h);
int seed_handle(gymnet_vecenv *h, uint64_t seed);                    // Env.Seed(int): new key, tick 0, captured graphs dropped
int resident_stop(gymnet_vecenv *h);                                 // tells a running resident kernel to leave and waits until it has

}  // namespace gymnet
