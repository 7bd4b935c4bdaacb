// group.hip — gymnet_group_*: ONE process driving G members (one per GPU) of a lane-sharded batch, with a hand-written
// direct (full-mesh) all-gather of observations over peer-mapped memory and an RCCL variant behind the same call.
//
// The reference has no multi-device code (its Distributed* classes are an in-process thread pool, src/Gym/Internal/
// Threading/*); what shards is the independence of VecEnvWrapper's sequential map (src/Gym/Envs/VecEnvWrapper.cs:22-24).
// This is the form a P/Invoking C# host uses (it has no torch.distributed): SURVEY.md §8(e).
//
// Layout.  Member m owns global lanes [m*n, (m+1)*n), n = N/G.  Every member holds, on its own GPU, a replica of the whole
// batch's observations, rank-major [G][D][n]; the member's LIVE observation arrays are slice [m] of its own replica
// (gymnet_config.d_ext_obs), so the send side of the gather is zero-copy.  With GYMNET_FLAG_DOUBLE_BUFFER there are two
// replicas per member, written alternately by consecutive steps, and the gather of one overlaps the step into the other.
//
// Ordering.  All cross-member dependencies are HIP events, O(G) host calls per gather:
//   ev_step[m]  member m's stream has passed the gather call (its step is done, and every consumer of the replica that is
//               about to be overwritten was queued before)            -> joined into ev_ready on the join stream
//   ev_push[b][m]  member m's push of buffer b has completed            -> joined into ev_done[b]
//   wait_gather: every member's stream waits for ev_done[b] — work queued afterwards sees all G slices.
//   A step that is about to overwrite buffer b first waits for the member's own ev_push[b][m] (its slice is still being read).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>

#include "handle.hpp"

using namespace gymnet;

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

// librccl is loaded on demand (gather mode RCCL only), by soname first so that a process which already carries an RCCL
// (torch ships one) keeps exactly one copy.
int load_rccl(RcclApi &r) {
    if (r.lib) return GYMNET_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) return fail(nullptr, GYMNET_ERR_RCCL, "cannot load librccl: %s", dlerror());
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    if (!r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString)
        return fail(nullptr, GYMNET_ERR_RCCL, "librccl lacks a required symbol");
    return GYMNET_OK;
}

}  // namespace

struct gymnet_group {
    gymnet_group_config cfg{};
    int G = 0, obs_dim = 0, nbuf = 1;
    int64_t n_local = 0, slice = 0;            // slice = obs_dim * n_local observation ELEMENTS of one member
    size_t esz = 4;                            // bytes per observation element: 4, or 8 for GYMNET_FLAG_F64 members
    bool overlap = false;
    std::vector<int> devices;
    std::vector<gymnet_vecenv *> members;
    std::vector<char *> replica[2];            // [buffer][member] -> [G][D][n_local] elements on the member's device
    std::vector<hipStream_t> gstream;          // per member: the stream its push / ncclAllGather runs on
    std::vector<hipEvent_t> ev_step, ev_push[2];
    hipStream_t join_stream = nullptr;         // on devices[0]
    hipEvent_t ev_ready = nullptr, ev_done[2] = {nullptr, nullptr};
    bool push_pending[2] = {false, false};
    int last_gathered = -1;
    RcclApi rccl;
    std::vector<ncclComm_t> comms;
    std::atomic<bool> busy{false};
};

namespace {

// Takes the busy flag of every member for the duration of a group call (released in reverse on exit): while the group
// steps / gathers, a caller holding a borrowed member handle gets GYMNET_ERR_ALREADY_STEPPING instead of desynchronising
// the group's buffer bookkeeping.
struct MembersBusy {
    gymnet_group *g;
    size_t taken = 0;
    bool ok = true;
    explicit MembersBusy(gymnet_group *gg) : g(gg) {
        for (auto *h : g->members) {
            if (!h) { ++taken; continue; }
            bool expect = false;
            if (!h->busy.compare_exchange_strong(expect, true)) { ok = false; break; }
            ++taken;
        }
        if (!ok) release();
    }
    void release() {
        for (size_t m = 0; m < taken && m < g->members.size(); ++m) if (g->members[m]) g->members[m]->busy.store(false);
        taken = 0;
    }
    ~MembersBusy() { release(); }
};

#define GROUP_ENTER(g)                                                                                         \
    if (!(g)) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "null group");                                      \
    bool expect_ = false;                                                                                      \
    if (!(g)->busy.compare_exchange_strong(expect_, true))                                                     \
        return fail(nullptr, GYMNET_ERR_ALREADY_STEPPING, "group is in use by another call");                  \
    struct Release_ { gymnet_group *p; ~Release_() { p->busy.store(false); } } release_{g};                     \
    MembersBusy members_busy_(g);   /* a borrowed member handle (gymnet_group_member) cannot be stepped concurrently */ \
    if (!members_busy_.ok) return fail(nullptr, GYMNET_ERR_ALREADY_STEPPING, "a member handle of this group is in use by another call"); \
    DeviceScope dev_scope_;                                                                                    \
    (void)hipGetLastError()

#define RCCL_TRY(g, expr)                                                                                       \
    do {                                                                                                        \
        ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess) return fail(nullptr, GYMNET_ERR_RCCL, "%s failed: %s", #expr, (g)->rccl.GetErrorString(r_)); \
    } while (0)

int buffer_of(gymnet_group *g) { return g->members[0]->cur; }
// member m's slice inside a replica (bytes)
inline size_t slice_bytes(const gymnet_group *g) { return (size_t)g->slice * g->esz; }

// before member m's next step overwrites buffer b, its own push of b (which is reading the member's slice) must be done
int guard_overwrite(gymnet_group *g, int b) {
    if (!g->push_pending[b]) return GYMNET_OK;
    for (int m = 0; m < g->G; ++m) {
        HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
        HIP_TRY(nullptr, hipStreamWaitEvent(g->members[m]->stream, g->ev_push[b][m], 0));
    }
    g->push_pending[b] = false;
    return GYMNET_OK;
}

int wait_gather(gymnet_group *g) {
    if (g->last_gathered < 0) return GYMNET_OK;
    for (int m = 0; m < g->G; ++m) {
        HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
        HIP_TRY(nullptr, hipStreamWaitEvent(g->members[m]->stream, g->ev_done[g->last_gathered], 0));
    }
    return GYMNET_OK;
}

int allgather(gymnet_group *g) {
    if (g->cfg.gather == GYMNET_GATHER_NONE) return fail(nullptr, GYMNET_ERR_UNSUPPORTED, "group was created with GYMNET_GATHER_NONE");
    const int b = buffer_of(g), G = g->G;
    // 1. every member's stream reaches the gather point -> ev_ready
    for (int m = 0; m < G; ++m) {
        HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
        HIP_TRY(nullptr, hipEventRecord(g->ev_step[m], g->members[m]->stream));
    }
    HIP_TRY(nullptr, hipSetDevice(g->devices[0]));
    for (int m = 0; m < G; ++m) HIP_TRY(nullptr, hipStreamWaitEvent(g->join_stream, g->ev_step[m], 0));
    HIP_TRY(nullptr, hipEventRecord(g->ev_ready, g->join_stream));
    // 2. the exchange, on the gather streams
    for (int m = 0; m < G; ++m) {
        HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
        HIP_TRY(nullptr, hipStreamWaitEvent(g->gstream[m], g->ev_ready, 0));
    }
    if (g->cfg.gather == GYMNET_GATHER_DIRECT) {
        for (int m = 0; m < G; ++m) {
            PushArgs a{};            // the push moves 4-byte words: a float64 slice is two per element
            a.src = reinterpret_cast<const float *>(g->replica[b][m] + (size_t)m * slice_bytes(g));
            a.count = (int64_t)(slice_bytes(g) / 4);
            for (int p = 0; p < G; ++p)
                if (p != m) a.dst[a.npeers++] = reinterpret_cast<float *>(g->replica[b][p] + (size_t)m * slice_bytes(g));
            HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
            HIP_TRY(nullptr, launch_push_obs(a, g->gstream[m]));
        }
    } else {
        RCCL_TRY(g, g->rccl.GroupStart());
        for (int m = 0; m < G; ++m) {
            HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
            RCCL_TRY(g, g->rccl.AllGather(g->replica[b][m] + (size_t)m * slice_bytes(g), g->replica[b][m], (size_t)g->slice,
                                          g->esz == 8 ? ncclDouble : ncclFloat, g->comms[m], g->gstream[m]));
        }
        RCCL_TRY(g, g->rccl.GroupEnd());
        (void)hipGetLastError();      // see gymnet_group_create: RCCL may leave a stale HIP error behind
    }
    for (int m = 0; m < G; ++m) {
        HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
        HIP_TRY(nullptr, hipEventRecord(g->ev_push[b][m], g->gstream[m]));
    }
    // 3. all exchanges done -> ev_done[b]
    HIP_TRY(nullptr, hipSetDevice(g->devices[0]));
    for (int m = 0; m < G; ++m) HIP_TRY(nullptr, hipStreamWaitEvent(g->join_stream, g->ev_push[b][m], 0));
    HIP_TRY(nullptr, hipEventRecord(g->ev_done[b], g->join_stream));
    g->push_pending[b] = true;
    g->last_gathered = b;
    if (!g->overlap) {           // single buffer: the next step would overwrite what is being sent — order it behind the gather now
        ST_TRY(wait_gather(g));
        g->push_pending[b] = false;
    }
    return GYMNET_OK;
}

void destroy_group(gymnet_group *g) {
    if (!g) return;
    DeviceScope scope;
    for (int m = 0; m < (int)g->members.size(); ++m) {
        (void)hipSetDevice(g->devices[m]);
        if (g->members[m]) (void)hipStreamSynchronize(g->members[m]->stream);
        if (m < (int)g->gstream.size() && g->gstream[m]) (void)hipStreamSynchronize(g->gstream[m]);
    }
    if (g->join_stream) { (void)hipSetDevice(g->devices[0]); (void)hipStreamSynchronize(g->join_stream); }
    for (auto c : g->comms) if (c && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(c);
    for (auto *h : g->members) if (h) (void)gymnet_vecenv_destroy(h);
    for (int m = 0; m < g->G; ++m) {
        if (m >= (int)g->devices.size() || m >= (int)g->gstream.size()) break;   // nothing was created yet (e.g. a rejected ordinal)
        (void)hipSetDevice(g->devices[m]);
        for (int b = 0; b < 2; ++b) {
            if (m < (int)g->replica[b].size() && g->replica[b][m]) (void)hipFree(g->replica[b][m]);
            if (m < (int)g->ev_push[b].size() && g->ev_push[b][m]) (void)hipEventDestroy(g->ev_push[b][m]);
        }
        if (m < (int)g->ev_step.size() && g->ev_step[m]) (void)hipEventDestroy(g->ev_step[m]);
        if (m < (int)g->gstream.size() && g->gstream[m]) (void)hipStreamDestroy(g->gstream[m]);
    }
    if (!g->devices.empty() && !g->gstream.empty()) (void)hipSetDevice(g->devices[0]);
    if (g->ev_ready) (void)hipEventDestroy(g->ev_ready);
    for (int b = 0; b < 2; ++b) if (g->ev_done[b]) (void)hipEventDestroy(g->ev_done[b]);
    if (g->join_stream) (void)hipStreamDestroy(g->join_stream);
    // librccl stays loaded for the life of the process (unloading a library that owns GPU state is not safe)
    delete g;
}

}  // namespace

extern "C" {

// ---- peer buffers (one process per GPU): HIP IPC export / import + the same push kernel ------------------------------
int gymnet_peer_buffer_create(int device, int64_t bytes, void **d_ptr, gymnet_ipc_handle *handle) {
    return guarded([&]() -> int {
    if (!d_ptr || !handle || bytes <= 0) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "bad d_ptr/handle/bytes");
    static_assert(sizeof(gymnet_ipc_handle) == sizeof(hipIpcMemHandle_t), "gymnet_ipc_handle must be a hipIpcMemHandle_t");
    DeviceScope scope;
    HIP_TRY(nullptr, hipSetDevice(device));
    // The export (dmabuf under HSA_ENABLE_IPC_MODE_LEGACY=0) has been seen to fail with "invalid argument" now and then
    // when several processes create and tear down peer buffers on one GPU at the same moment; a fresh allocation at
    // another address succeeds.  So: a few attempts, each keeping the rejected allocation alive until the end (the next
    // hipMalloc then cannot hand back the same range), with a short back-off.
    constexpr int kAttempts = 4;
    void *rejected[kAttempts] = {};
    hipError_t e = hipSuccess;
    const char *what = "";
    void *p = nullptr;
    bool oom = false;
    for (int attempt = 0; attempt < kAttempts; ++attempt) {
        p = nullptr;
        e = hipMalloc(&p, (size_t)bytes);
        what = "hipMalloc";
        if (e != hipSuccess) { p = nullptr; oom = true; break; }
        e = hipMemset(p, 0, (size_t)bytes);
        what = "hipMemset";
        if (e == hipSuccess) {
            e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t *>(handle), p);
            what = "hipIpcGetMemHandle";
        }
        if (e == hipSuccess) break;
        (void)hipGetLastError();
        rejected[attempt] = p;
        p = nullptr;
        usleep(2000u << attempt);
    }
    for (void *r : rejected) if (r) (void)hipFree(r);
    if (e != hipSuccess || !p) {
        if (p) (void)hipFree(p);
        return fail(nullptr, oom ? GYMNET_ERR_OOM : GYMNET_ERR_HIP,
                    "%s(%lld bytes) failed: %s (peer buffers need HSA_ENABLE_IPC_MODE_LEGACY=0 on this platform)", what,
                    (long long)bytes, hipGetErrorString(e));
    }
    *d_ptr = p;
    return GYMNET_OK;
    });
}

int gymnet_peer_buffer_open(int device, const gymnet_ipc_handle *handle, void **d_ptr) {
    return guarded([&]() -> int {
    if (!d_ptr || !handle) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "null argument");
    DeviceScope scope;
    HIP_TRY(nullptr, hipSetDevice(device));
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof h);
    void *p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(nullptr, GYMNET_ERR_HIP, "hipIpcOpenMemHandle failed: %s", hipGetErrorString(e));
    *d_ptr = p;
    return GYMNET_OK;
    });
}

int gymnet_peer_buffer_close(int device, void *d_ptr) {
    return guarded([&]() -> int {
    if (!d_ptr) return GYMNET_OK;
    DeviceScope scope;
    HIP_TRY(nullptr, hipSetDevice(device));
    HIP_TRY(nullptr, hipIpcCloseMemHandle(d_ptr));
    return GYMNET_OK;
    });
}

int gymnet_peer_buffer_destroy(int device, void *d_ptr) {
    return guarded([&]() -> int {
    if (!d_ptr) return GYMNET_OK;
    DeviceScope scope;
    HIP_TRY(nullptr, hipSetDevice(device));
    HIP_TRY(nullptr, hipFree(d_ptr));
    return GYMNET_OK;
    });
}

int gymnet_push_obs_device(int device, void *stream, const float *d_src, float *const *d_dst, int32_t npeers, int64_t count) {
    return guarded([&]() -> int {
    if (!d_src || count < 0 || npeers < 0 || npeers > kMaxPeers || (npeers > 0 && !d_dst)) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "bad src/dst/npeers/count");
    PushArgs a{};
    a.src = d_src; a.count = count; a.npeers = npeers;
    for (int p = 0; p < npeers; ++p) { if (!d_dst[p]) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "d_dst[%d] is null", p); a.dst[p] = d_dst[p]; }
    DeviceScope scope;
    HIP_TRY(nullptr, hipSetDevice(device));
    HIP_TRY(nullptr, launch_push_obs(a, static_cast<hipStream_t>(stream)));
    return GYMNET_OK;
    });
}

int gymnet_group_create(const gymnet_group_config *cfg, gymnet_group **out) {
    return guarded([&]() -> int {
    if (!out) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (!cfg) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "cfg is null");
    if (cfg->struct_size != sizeof(gymnet_group_config))
        return fail(nullptr, GYMNET_ERR_INVALID_ARG, "cfg.struct_size %u != %zu (ABI mismatch)", cfg->struct_size, sizeof(gymnet_group_config));
    if (cfg->env_id < 0 || cfg->env_id > 3) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "unknown env_id %d", cfg->env_id);
    const int G = cfg->num_members;
    if (G < 1 || G > kMaxPeers + 1) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "num_members %d not in [1, %d]", G, kMaxPeers + 1);
    if (cfg->global_num_envs < G || cfg->global_num_envs % G != 0)   // equal blocks: the replica is [G][D][N/G]
        return fail(nullptr, GYMNET_ERR_INVALID_ARG, "global_num_envs %lld must be a positive multiple of num_members %d",
                    (long long)cfg->global_num_envs, G);
    if (cfg->gather < GYMNET_GATHER_NONE || cfg->gather > GYMNET_GATHER_RCCL) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "bad gather mode");
    if ((cfg->flags & GYMNET_FLAG_F64) && cfg->env_id != GYMNET_ENV_CARTPOLE)
        return fail(nullptr, GYMNET_ERR_UNSUPPORTED, "GYMNET_FLAG_F64 exists for CartPole only");
    int ndev = 0;
    ST_TRY(gymnet_device_count(&ndev));

    gymnet_group *g = new (std::nothrow) gymnet_group();
    if (!g) return fail(nullptr, GYMNET_ERR_OOM, "host allocation failed");
    DeviceScope scope;
    g->cfg = *cfg;
    g->cfg.devices = nullptr;
    g->G = G;
    g->obs_dim = kEnvs[cfg->env_id].obs_dim;
    g->n_local = cfg->global_num_envs / G;
    g->slice = (int64_t)g->obs_dim * g->n_local;
    g->esz = (cfg->flags & GYMNET_FLAG_F64) ? 8 : 4;
    g->overlap = (cfg->flags & GYMNET_FLAG_DOUBLE_BUFFER) != 0;
    g->nbuf = g->overlap ? 2 : 1;
    g->devices.resize(G);
    for (int m = 0; m < G; ++m) g->devices[m] = cfg->devices ? cfg->devices[m] : m;

#define G_FAIL(status, ...) do { int s_ = fail(nullptr, status, __VA_ARGS__); destroy_group(g); return s_; } while (0)
#define G_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) G_FAIL(e_ == hipErrorOutOfMemory ? GYMNET_ERR_OOM : GYMNET_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

    for (int m = 0; m < G; ++m)
        if (g->devices[m] < 0 || g->devices[m] >= ndev) G_FAIL(GYMNET_ERR_INVALID_ARG, "devices[%d] = %d not in [0, %d)", m, g->devices[m], ndev);
    if (cfg->gather == GYMNET_GATHER_RCCL)
        for (int a = 0; a < G; ++a)
            for (int b2 = a + 1; b2 < G; ++b2)
                if (g->devices[a] == g->devices[b2])
                    G_FAIL(GYMNET_ERR_UNSUPPORTED, "GYMNET_GATHER_RCCL needs one distinct GPU per member (device %d is used twice); "
                           "use GYMNET_GATHER_DIRECT for logical members on one GPU", g->devices[a]);

    // peer access for the direct push: every member stores into every other member's replica
    if (cfg->gather == GYMNET_GATHER_DIRECT) {
        for (int a = 0; a < G; ++a)
            for (int b2 = 0; b2 < G; ++b2) {
                const int da = g->devices[a], db = g->devices[b2];
                if (da == db) continue;
                int can = 0;
                G_HIP(hipDeviceCanAccessPeer(&can, da, db));
                if (!can) G_FAIL(GYMNET_ERR_UNSUPPORTED, "device %d cannot access device %d's memory (no xGMI/PCIe peer path)", da, db);
                G_HIP(hipSetDevice(da));
                hipError_t e = hipDeviceEnablePeerAccess(db, 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                else if (e != hipSuccess) G_FAIL(GYMNET_ERR_HIP, "hipDeviceEnablePeerAccess(%d -> %d) failed: %s", da, db, hipGetErrorString(e));
            }
    }

    for (int b = 0; b < 2; ++b) { g->replica[b].assign(G, nullptr); g->ev_push[b].assign(G, nullptr); }
    g->gstream.assign(G, nullptr);
    g->ev_step.assign(G, nullptr);
    g->members.assign(G, nullptr);
    for (int m = 0; m < G; ++m) {
        G_HIP(hipSetDevice(g->devices[m]));
        for (int b = 0; b < g->nbuf; ++b) {
            void *q = nullptr;
            G_HIP(hipMalloc(&q, (size_t)G * slice_bytes(g)));
            g->replica[b][m] = static_cast<char *>(q);
            G_HIP(hipMemset(q, 0, (size_t)G * slice_bytes(g)));
        }
        for (int b = 0; b < 2; ++b) G_HIP(hipEventCreateWithFlags(&g->ev_push[b][m], hipEventDisableTiming));
        G_HIP(hipEventCreateWithFlags(&g->ev_step[m], hipEventDisableTiming));
        G_HIP(hipStreamCreateWithFlags(&g->gstream[m], hipStreamNonBlocking));
        gymnet_config mc{};
        mc.struct_size = sizeof mc;
        mc.env_id = cfg->env_id;
        mc.num_envs = g->n_local;
        mc.lane_offset = (int64_t)m * g->n_local;
        mc.device = g->devices[m];
        mc.flags = cfg->flags;
        mc.seed = cfg->seed;
        mc.stream = nullptr;
        mc.d_ext_obs = g->replica[0][m] + (size_t)m * slice_bytes(g);
        mc.ext_obs_stride = g->n_local;
        mc.max_episode_steps = cfg->max_episode_steps;
        mc.d_ext_obs_alt = g->overlap ? g->replica[1][m] + (size_t)m * slice_bytes(g) : nullptr;
        int s = gymnet_vecenv_create(&mc, &g->members[m]);
        if (s != GYMNET_OK) { destroy_group(g); return s; }
    }
    G_HIP(hipSetDevice(g->devices[0]));
    G_HIP(hipStreamCreateWithFlags(&g->join_stream, hipStreamNonBlocking));
    G_HIP(hipEventCreateWithFlags(&g->ev_ready, hipEventDisableTiming));
    for (int b = 0; b < 2; ++b) G_HIP(hipEventCreateWithFlags(&g->ev_done[b], hipEventDisableTiming));

    if (cfg->gather == GYMNET_GATHER_RCCL) {
        int s = load_rccl(g->rccl);
        if (s != GYMNET_OK) { destroy_group(g); return s; }
        g->comms.assign(G, nullptr);
        ncclResult_t r = g->rccl.CommInitAll(g->comms.data(), G, g->devices.data());
        if (r != ncclSuccess) G_FAIL(GYMNET_ERR_RCCL, "ncclCommInitAll failed: %s", g->rccl.GetErrorString(r));
        // RCCL probes devices / peers while it initialises and leaves HIP's sticky per-thread "last error" set (seen:
        // hipErrorInvalidDevice); the kernel launchers read that slot after each launch, so it has to be cleared here
        (void)hipGetLastError();
    }
#undef G_HIP
#undef G_FAIL
    *out = g;
    return GYMNET_OK;
    });
}

int gymnet_group_destroy(gymnet_group *g) {
    return guarded([&]() -> int {
    destroy_group(g);
    return GYMNET_OK;
    });
}

int gymnet_group_size(gymnet_group *g, int32_t *num_members, int64_t *lanes_per_member) {
    return guarded([&]() -> int {
    if (!g) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "null group");
    if (num_members) *num_members = g->G;
    if (lanes_per_member) *lanes_per_member = g->n_local;
    return GYMNET_OK;
    });
}

int gymnet_group_member(gymnet_group *g, int32_t member, gymnet_vecenv **out) {
    return guarded([&]() -> int {
    if (!g || !out) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "null argument");
    if (member < 0 || member >= g->G) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "member %d not in [0, %d)", member, g->G);
    *out = g->members[member];
    return GYMNET_OK;
    });
}

int gymnet_group_seed(gymnet_group *g, uint64_t seed) {
    return guarded([&]() -> int {
    GROUP_ENTER(g);
    for (int m = 0; m < g->G; ++m) {            // members are held busy by GROUP_ENTER: the internal form, per device
        HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
        ST_TRY(seed_handle(g->members[m], seed));
    }
    return GYMNET_OK;
    });
}

int gymnet_group_reset_device(gymnet_group *g) {
    return guarded([&]() -> int {
    GROUP_ENTER(g);
    ST_TRY(guard_overwrite(g, buffer_of(g)));       // a reset rewrites the CURRENT buffer in place
    for (int m = 0; m < g->G; ++m) {
        HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
        ST_TRY(launch_reset_lanes(g->members[m], nullptr));
    }
    return GYMNET_OK;
    });
}

int gymnet_group_step_device(gymnet_group *g, const void *const *d_actions) {
    return guarded([&]() -> int {
    GROUP_ENTER(g);
    if (!d_actions) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "d_actions is null");
    for (int m = 0; m < g->G; ++m) {
        if (!d_actions[m]) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "d_actions[%d] is null", m);
        if (g->members[m]->lcfg.vec > 1 && !aligned_to(d_actions[m], 4 * g->members[m]->lcfg.vec))
            return fail(nullptr, GYMNET_ERR_INVALID_ARG, "d_actions[%d] must be %d-byte aligned", m, 4 * g->members[m]->lcfg.vec);
    }
    if (g->cfg.flags & GYMNET_FLAG_VALIDATE_ACTIONS)     // Discrete.Contains over the WHOLE batch before any member changes state
        for (int m = 0; m < g->G; ++m) {
            HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
            ST_TRY(validate_staged_actions(g->members[m], d_actions[m]));
        }
    ST_TRY(guard_overwrite(g, g->overlap ? buffer_of(g) ^ 1 : buffer_of(g)));
    for (int m = 0; m < g->G; ++m) {
        HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
        ST_TRY(launch_one_step(g->members[m], d_actions[m]));
    }
    return GYMNET_OK;
    });
}

int gymnet_group_rollout_device(gymnet_group *g, const void *const *d_actions, int64_t steps, int64_t action_stride, int64_t ring) {
    return guarded([&]() -> int {
    GROUP_ENTER(g);
    if (!d_actions) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "d_actions is null");
    for (int m = 0; m < g->G; ++m) if (!d_actions[m]) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "d_actions[%d] is null", m);
    ST_TRY(guard_overwrite(g, 0));
    ST_TRY(guard_overwrite(g, 1));
    // Members are independent.  ONE host thread cannot feed G GPUs with one eager launch per 7 us step (~3.5 us of host time
    // per launch), so with G > 1 every member replays a captured graph of `ring` steps: one host call per `ring` launches,
    // and the members' trains run concurrently on their own streams.
    for (int m = 0; m < g->G; ++m) {
        HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
        ST_TRY(rollout_steps(g->members[m], d_actions[m], steps, action_stride, ring, g->G > 1 ? 1 : -1));
    }
    return GYMNET_OK;
    });
}

int gymnet_group_allgather_obs(gymnet_group *g) {
    return guarded([&]() -> int {
    GROUP_ENTER(g);
    return allgather(g);
    });
}

int gymnet_group_wait_gather(gymnet_group *g) {
    return guarded([&]() -> int {
    GROUP_ENTER(g);
    return wait_gather(g);
    });
}

int gymnet_group_global_obs(gymnet_group *g, int32_t member, void **d_obs_all) {
    return guarded([&]() -> int {
    if (!g || !d_obs_all) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "null argument");
    if (member < 0 || member >= g->G) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "member %d not in [0, %d)", member, g->G);
    *d_obs_all = g->replica[g->last_gathered < 0 ? buffer_of(g) : g->last_gathered][member];
    return GYMNET_OK;
    });
}

int gymnet_group_read_replica(gymnet_group *g, int32_t member, void *replica_out) {
    return guarded([&]() -> int {
    GROUP_ENTER(g);
    if (!replica_out) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "replica_out is null");
    if (member < 0 || member >= g->G) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "member %d not in [0, %d)", member, g->G);
    ST_TRY(wait_gather(g));
    const int b = g->last_gathered < 0 ? buffer_of(g) : g->last_gathered;
    HIP_TRY(nullptr, hipSetDevice(g->devices[member]));
    hipStream_t st = g->members[member]->stream;
    HIP_TRY(nullptr, hipMemcpyAsync(replica_out, g->replica[b][member], (size_t)g->G * slice_bytes(g), hipMemcpyDeviceToHost, st));
    HIP_TRY(nullptr, hipStreamSynchronize(st));
    return GYMNET_OK;
    });
}

int gymnet_group_sync(gymnet_group *g) {
    return guarded([&]() -> int {
    GROUP_ENTER(g);
    for (int m = 0; m < g->G; ++m) {
        HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
        HIP_TRY(nullptr, hipStreamSynchronize(g->members[m]->stream));
        HIP_TRY(nullptr, hipStreamSynchronize(g->gstream[m]));
    }
    HIP_TRY(nullptr, hipSetDevice(g->devices[0]));
    HIP_TRY(nullptr, hipStreamSynchronize(g->join_stream));
    return GYMNET_OK;
    });
}

// Host-boundary forms: the whole batch in NDArray layout.  Members are queued first and synchronized last, so the G GPUs
// (and their PCIe links) work concurrently.
static int group_copy_out(gymnet_group *g, void *obs_out, float *reward_out, uint8_t *done_out) {
    const int64_t n = g->n_local;
    for (int m = 0; m < g->G; ++m) {
        HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
        gymnet_vecenv *h = g->members[m];
        void *o = obs_out ? static_cast<char *>(obs_out) + (size_t)m * n * g->obs_dim * g->esz : nullptr;
        float *r = reward_out ? reward_out + (size_t)m * n : nullptr;
        uint8_t *d = done_out ? done_out + (size_t)m * n : nullptr;
        if (h->hm_block) ST_TRY(copy_out(h, o, r, d));
        else ST_TRY(queue_copy_out(h, o, r, d));
    }
    for (int m = 0; m < g->G; ++m) {
        HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
        HIP_TRY(nullptr, hipStreamSynchronize(g->members[m]->stream));
    }
    return GYMNET_OK;
}

int gymnet_group_reset(gymnet_group *g, void *obs_out) {
    return guarded([&]() -> int {
    GROUP_ENTER(g);
    ST_TRY(guard_overwrite(g, buffer_of(g)));
    for (int m = 0; m < g->G; ++m) {
        HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
        ST_TRY(launch_reset_lanes(g->members[m], nullptr));
    }
    return group_copy_out(g, obs_out, nullptr, nullptr);
    });
}

int gymnet_group_step(gymnet_group *g, const void *actions, void *obs_out, float *reward_out, uint8_t *done_out) {
    return guarded([&]() -> int {
    GROUP_ENTER(g);
    if (!actions) return fail(nullptr, GYMNET_ERR_INVALID_ARG, "actions is null");
    const int64_t n = g->n_local;
    std::vector<const void *> d_act((size_t)g->G, nullptr);
    for (int m = 0; m < g->G; ++m) {
        HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
        ST_TRY(stage_host_actions(g->members[m], static_cast<const char *>(actions) + (size_t)m * n * 4, &d_act[m], false));
    }
    if (g->cfg.flags & GYMNET_FLAG_VALIDATE_ACTIONS)
        for (int m = 0; m < g->G; ++m) {
            HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
            ST_TRY(validate_staged_actions(g->members[m], d_act[m]));
        }
    ST_TRY(guard_overwrite(g, g->overlap ? buffer_of(g) ^ 1 : buffer_of(g)));
    for (int m = 0; m < g->G; ++m) {
        HIP_TRY(nullptr, hipSetDevice(g->devices[m]));
        ST_TRY(launch_one_step(g->members[m], d_act[m]));
    }
    return group_copy_out(g, obs_out, reward_out, done_out);
    });
}

}  // extern "C"
