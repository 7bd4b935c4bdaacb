// kernels.hip — the env-INDEPENDENT HIP kernels of the batched classic-control engine (row-major packing / host export of the
// observations, done-list gather, action validation, batched space sampling, the direct all-gather push) and the dispatch from
// (env_id, state scalar) to the env's own translation unit (env_*.hip: the step / rollout / reset kernels, step_kernels.hpp).
// Written for gfx950 (CDNA4, wave64); compiled with -ffp-contract=off.
#include "kernels.hpp"

#include <cstdio>

#include "lanes.hpp"
#include "philox.hpp"

namespace gymnet {

// SoA [O][stride] -> row-major [n][O] (the NDArray layout at the host boundary), T = float or double.  Reads are coalesced per
// component; each lane then writes its O contiguous elements (16 bytes per store where O allows), so a wave writes
// 64 * O * sizeof(T) contiguous bytes.
template <class T, int O>
__global__ __launch_bounds__(256) void pack_obs_kernel(const T *__restrict__ obs, int64_t stride, T *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T v[O];
#pragma unroll
    for (int k = 0; k < O; ++k) v[k] = obs[k * stride + i];
    constexpr int P = (int)(16 / sizeof(T));          // elements per 16-byte store
    if constexpr (O % P == 0) {
        typedef typename VecOf<T, P>::type V;
#pragma unroll
        for (int k = 0; k < O; k += P) {
            V t;
#pragma unroll
            for (int q = 0; q < P; ++q) t[q] = v[k + q];
            *reinterpret_cast<V *>(out + i * O + k) = t;
        }
    } else if constexpr (sizeof(T) == 4 && O % 2 == 0) {
#pragma unroll
        for (int k = 0; k < O; k += 2) *reinterpret_cast<float2 *>(out + i * O + k) = make_float2(v[k], v[k + 1]);
    } else {
#pragma unroll
        for (int k = 0; k < O; ++k) out[i * O + k] = v[k];
    }
}

// Gathers the kShards segments of one step's done list — and of the records written beside it — into compact arrays, and /
// or applies the records to the dense per-lane arrays (the "last finished episode of every lane" view).  One workgroup per
// shard; every workgroup recomputes the (tiny) exclusive scan of the 256 shard counts in LDS, then copies its segment coalesced.
template <class R>
__global__ __launch_bounds__(256) void compact_done_kernel(const CompactArgsT<R> a) {
    __shared__ uint32_t scan[kShards];
    const int t = threadIdx.x;
    const uint32_t mine = a.counts[t * kCountStride];
    scan[t] = mine;
    __syncthreads();
    for (int d = 1; d < kShards; d <<= 1) {            // Hillis-Steele inclusive scan, 8 rounds
        const uint32_t v = t >= d ? scan[t - d] : 0u;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    const int shard = blockIdx.x;
    const uint32_t cnt = a.counts[shard * kCountStride];
    const uint32_t start = scan[shard] - cnt;
    if (shard == 0 && t == 0 && a.out_count) *a.out_count = scan[kShards - 1];
    const int64_t seg0 = (int64_t)shard * a.cap;
    const int O = a.obs_dim;
    for (uint32_t k = t; k < cnt; k += 256) {
        const int32_t lane = a.list[seg0 + k];
        const uint32_t dst = start + k;
        const bool fits = (int64_t)dst < a.out_capacity;
        if (a.out_list && fits) a.out_list[dst] = lane;
        if (a.rec_ret) {
            const float r = a.rec_ret[seg0 + k];
            const int32_t l = a.rec_len[seg0 + k];
            if (a.out_ret && fits) a.out_ret[dst] = r;
            if (a.out_len && fits) a.out_len[dst] = l;
            if (a.dense_ret) { a.dense_ret[lane] = r; a.dense_len[lane] = l; }
        }
        if (a.rec_obs) {
            for (int c = 0; c < O; ++c) {
                const R v = a.rec_obs[((int64_t)shard * O + c) * a.cap + k];
                if (a.out_obs && fits) a.out_obs[(int64_t)dst * O + c] = v;
                if (a.dense_obs) a.dense_obs[(int64_t)c * a.n + lane] = v;
            }
        }
    }
}

// The episode records of a fused rollout (RolloutArgs::ep_*: kShards segments of (t, lane, return, length)) gathered into compact
// arrays; same shape as compact_done_kernel — every workgroup recomputes the scan of the 256 shard counts in LDS — but with
// kGatherSplit workgroups per shard (blockIdx.y): a 256-step CartPole rollout at 2^20 lanes leaves 12 M records, and one workgroup
// per shard moved them in 304 us (1.2 us per vector step of the rollout); split eight ways the copy runs at memory speed.
// A shard's count may exceed its capacity: the records beyond it went to the shared overflow segment (RolloutArgs::ov_*), which the
// extra row of workgroups (blockIdx.x == kShards) appends after the last shard's records.
constexpr int kGatherSplit = 8;
// thread t of the (fixed) 256-thread workgroup owns shard t in the scan and in the raw-count reduction
static_assert(kShards == 256, "gather_episodes_kernel / compact_done_kernel index the shard counters by threadIdx.x of a 256-thread workgroup");
__global__ __launch_bounds__(256) void gather_episodes_kernel(const EpisodeGatherArgs a) {
    __shared__ uint32_t scan[kShards];
    const int t = threadIdx.x;
    const uint32_t raw = a.counts[t * kCountStride];
    const uint32_t mine = (int64_t)raw < a.cap ? raw : (uint32_t)a.cap;
    scan[t] = mine;
    __syncthreads();
    for (int d = 1; d < kShards; d <<= 1) {            // Hillis-Steele inclusive scan, 8 rounds
        const uint32_t v = t >= d ? scan[t - d] : 0u;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    const int shard = blockIdx.x;
    if (shard == kShards) {
        // the extra row of workgroups: the shared overflow segment (segment kShards of the same arrays), appended after the last shard's records
        const uint32_t ovraw = a.counts[kShards * kCountStride];
        const uint32_t ovcnt = (int64_t)ovraw < a.ov_cap ? ovraw : (uint32_t)a.ov_cap;
        const uint32_t start = scan[kShards - 1];
        const int64_t seg0 = (int64_t)kShards * a.cap;
        for (uint32_t k = blockIdx.y * 256 + t; k < ovcnt; k += 256 * kGatherSplit) {
            const uint32_t dst = start + k;
            if ((int64_t)dst >= a.out_capacity) continue;
            if (a.out_t) a.out_t[dst] = a.ep_t[seg0 + k];
            if (a.out_lane) a.out_lane[dst] = a.ep_lane[seg0 + k];
            if (a.out_ret && a.ep_ret) a.out_ret[dst] = a.ep_ret[seg0 + k];
            if (a.out_len && a.ep_len) a.out_len[dst] = a.ep_len[seg0 + k];
        }
        return;
    }
    const uint32_t raw_s = a.counts[shard * kCountStride];
    const uint32_t cnt = (int64_t)raw_s < a.cap ? raw_s : (uint32_t)a.cap;
    const uint32_t start = scan[shard] - cnt;
    if (shard == 0 && blockIdx.y == 0 && a.out_count) {
        // out_count[0]: records written to the out arrays; out_count[1]: episodes that ENDED during the rollout (sum of the raw
        // shard counts: larger than [0] when a segment or the caller's capacity overflowed)
        __shared__ uint32_t raw_sum[kShards];
        raw_sum[t] = raw;
        __syncthreads();
        for (int d = kShards / 2; d > 0; d >>= 1) { if (t < d) raw_sum[t] += raw_sum[t + d]; __syncthreads(); }
        if (t == 0) {
            const uint32_t ovraw = a.counts[kShards * kCountStride];
            const uint64_t kept = (uint64_t)scan[kShards - 1] + ((int64_t)ovraw < a.ov_cap ? ovraw : (uint32_t)a.ov_cap);
            a.out_count[0] = (int64_t)kept < a.out_capacity ? (uint32_t)kept : (uint32_t)a.out_capacity;
            a.out_count[1] = raw_sum[0];
        }
    }
    const int64_t seg0 = (int64_t)shard * a.cap;
    for (uint32_t k = blockIdx.y * 256 + t; k < cnt; k += 256 * kGatherSplit) {
        const uint32_t dst = start + k;
        if ((int64_t)dst >= a.out_capacity) continue;
        if (a.out_t) a.out_t[dst] = a.ep_t[seg0 + k];
        if (a.out_lane) a.out_lane[dst] = a.ep_lane[seg0 + k];
        if (a.out_ret && a.ep_ret) a.out_ret[dst] = a.ep_ret[seg0 + k];
        if (a.out_len && a.ep_len) a.out_len[dst] = a.ep_len[seg0 + k];
    }
}

template <class T, int O>
__global__ __launch_bounds__(256) void export_small_kernel(const T *__restrict__ obs, int64_t stride, const float *__restrict__ reward,
                                                           const uint8_t *__restrict__ done, T *out_obs, float *out_reward,
                                                           uint8_t *out_done, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int k = 0; k < O; ++k) out_obs[i * O + k] = obs[k * stride + i];
    if (out_reward) out_reward[i] = reward[i];
    if (out_done) out_done[i] = done[i];
}

// The host boundary without staging: observations (SoA -> row-major [n][O]), rewards and done flags written STRAIGHT into
// page-locked, device-mapped host memory (gymnet_vecenv_host_buffers) — the stores themselves are the PCIe transfer, no
// device-side pack buffer, no memcpy calls.  A thread owns 4 consecutive lanes: it transposes their 4 x O observation words in
// registers and writes them as O 16-byte stores to 16 * O contiguous bytes, so a wave writes one contiguous 1 KiB * O block
// (full PCIe write payloads for any O, including the 3- and 6-wide observations).
template <int O>
__global__ __launch_bounds__(256) void export_host_kernel(const float *__restrict__ obs, int64_t stride, const float *__restrict__ reward,
                                                          const uint8_t *__restrict__ done, float *__restrict__ out_obs,
                                                          float *__restrict__ out_reward, uint8_t *__restrict__ out_done, int64_t n,
                                                          int vec_ok) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    if (vec_ok && i0 + 4 <= n) {
        if (out_obs) {
            float row[4 * O];
#pragma unroll
            for (int k = 0; k < O; ++k) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(obs + k * stride + i0);
                row[0 * O + k] = t.x; row[1 * O + k] = t.y; row[2 * O + k] = t.z; row[3 * O + k] = t.w;
            }
            f32x4 *dst = reinterpret_cast<f32x4 *>(out_obs + i0 * O);
#pragma unroll
            for (int q = 0; q < O; ++q) { f32x4 t; t.x = row[4 * q]; t.y = row[4 * q + 1]; t.z = row[4 * q + 2]; t.w = row[4 * q + 3]; dst[q] = t; }
        }
        if (out_reward) *reinterpret_cast<f32x4 *>(out_reward + i0) = *reinterpret_cast<const f32x4 *>(reward + i0);
        if (out_done) *reinterpret_cast<uint32_t *>(out_done + i0) = *reinterpret_cast<const uint32_t *>(done + i0);
    } else {
        for (int64_t i = i0; i < n && i < i0 + 4; ++i) {
            if (out_obs) for (int k = 0; k < O; ++k) out_obs[i * O + k] = obs[k * stride + i];
            if (out_reward) out_reward[i] = reward[i];
            if (out_done) out_done[i] = done[i];
        }
    }
}

// float64 handles (CartPole: 4 observation components): SoA [4][stride] -> row-major [n][4] doubles, reward / done beside it;
// `out_*` may be page-locked host memory, in which case the stores are the PCIe transfer (32 contiguous bytes per lane, 2 KiB per wave)
__global__ __launch_bounds__(256) void export_f64_kernel(const double *__restrict__ obs, int64_t stride, const float *__restrict__ reward,
                                                         const uint8_t *__restrict__ done, double *__restrict__ out_obs,
                                                         float *__restrict__ out_reward, uint8_t *__restrict__ out_done, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (out_obs) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 lo, hi;
        lo.x = obs[i]; lo.y = obs[stride + i]; hi.x = obs[2 * stride + i]; hi.y = obs[3 * stride + i];
        d2 *dst = reinterpret_cast<d2 *>(out_obs + i * 4);
        dst[0] = lo; dst[1] = hi;
    }
    if (out_reward) out_reward[i] = reward[i];
    if (out_done) out_done[i] = done[i];
}

__global__ __launch_bounds__(256) void fill_i32_kernel(int32_t *p, int32_t v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// Discrete.Contains(int) (Discrete.cs:38-40) over a batch: counts actions outside [0, nvals)
__global__ __launch_bounds__(256) void validate_discrete_kernel(const int32_t *__restrict__ a, int64_t n, int32_t nvals,
                                                                uint32_t *bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool invalid = i < n && (a[i] < 0 || a[i] >= nvals);
    const uint64_t m = __ballot(invalid);
    if (m && lane_id() == (uint32_t)(__ffsll((unsigned long long)m) - 1)) atomicAdd(bad, (uint32_t)__popcll(m));
}

// ---- stand-alone space sampling: action stream v2 (philox.hpp) ------------------------------------------------------------------
// One thread per GROUP of four consecutive GLOBAL lanes (the lanes that share a Philox call): thread t of a launch serves group
// (lane_offset >> 2) + t, whose first lane has local index i0 = 4 t - (lane_offset & 3) — negative for a batch that starts inside a
// group; elements outside [0, n) are not written.  A whole group that lies inside the batch on a 16-byte aligned address leaves as
// one dwordx4 store.
struct LaneGroup { uint64_t group; int64_t i0; };
__device__ __forceinline__ LaneGroup my_lane_group(uint64_t lane_offset) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    return LaneGroup{(lane_offset >> 2) + (uint64_t)t, 4 * t - (int64_t)(lane_offset & 3u)};
}

template <class T>
__device__ __forceinline__ void store_group(T *__restrict__ out, int64_t i0, int64_t n, const T (&v)[4]) {
    typedef typename VecOf<T, 4>::type V;
    if (i0 >= 0 && i0 + 4 <= n && (reinterpret_cast<uintptr_t>(out + i0) & 15u) == 0) {
        *reinterpret_cast<V *>(out + i0) = V{v[0], v[1], v[2], v[3]};
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (i0 + j >= 0 && i0 + j < n) out[i0 + j] = v[j];
}

// Discrete.Sample() (Discrete.cs:17-28, no mask): start + randint(0, n) = start + hi32(word A * n)
__global__ __launch_bounds__(256) void sample_discrete_kernel(int32_t *__restrict__ out, int64_t n, int32_t nvals,
                                                              int32_t start, uint64_t seed, uint64_t lane_offset,
                                                              uint64_t tick) {
    const LaneGroup g = my_lane_group(lane_offset);
    if (g.i0 >= n) return;
    const PhiloxWords r = action_group_words<true>(seed, g.group, tick);
    int32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = start + (int32_t)__umulhi(r.w[j], (uint32_t)nvals);
    store_group<int32_t>(out, g.i0, n, v);
}

// Discrete.Sample(mask) (Discrete.cs:18-26): valid = nonzero(mask == 1); any -> start + valid[choice(len(valid))], none -> start.
// choice(k) = hi32(word A * k) with the same Philox word the unmasked draw uses.  One row of `nvals` mask bytes per lane
// (mask_stride = nvals) or one shared row (mask_stride = 0).
__global__ __launch_bounds__(256) void sample_discrete_masked_kernel(int32_t *__restrict__ out, int64_t n, int32_t nvals, int32_t start,
                                                                     const uint8_t *__restrict__ mask, int64_t mask_stride,
                                                                     uint64_t seed, uint64_t lane_offset, uint64_t tick) {
    const LaneGroup g = my_lane_group(lane_offset);
    if (g.i0 >= n) return;
    const PhiloxWords r = action_group_words<true>(seed, g.group, tick);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = g.i0 + j;
        if (i < 0 || i >= n) continue;
        const uint8_t *m = mask + i * mask_stride;
        int32_t valid = 0;
        for (int32_t k = 0; k < nvals; ++k) valid += m[k] == 1 ? 1 : 0;
        int32_t pick = 0;
        if (valid > 0) {
            int32_t want = (int32_t)__umulhi(r.w[j], (uint32_t)valid);     // index into the list of valid actions
            for (int32_t k = 0; k < nvals; ++k) {
                if (m[k] == 1) { if (want == 0) { pick = k; break; } --want; }
            }
        }
        out[i] = start + pick;
    }
}

// The caller's epsilon-greedy composer (examples/.../PlaySessions/TrainingPlaySession.cs:46-52), batched:
//   if (Random.NextDouble() <= epsilon) action = ActionSpace.Sample(); else action = policy action
// Lane i: word A is the sampled action (identical to sample_discrete_kernel for the same seed / tick), word B the 24-bit uniform
// that is compared with epsilon.
__global__ __launch_bounds__(256) void compose_discrete_kernel(const int32_t *__restrict__ policy, int32_t *__restrict__ out,
                                                               int64_t n, int32_t nvals, float epsilon, uint64_t seed,
                                                               uint64_t lane_offset, uint64_t tick) {
    const LaneGroup g = my_lane_group(lane_offset);
    if (g.i0 >= n) return;
    const PhiloxWords r = action_group_words<true>(seed, g.group, tick), c = aux_group_words<true>(seed, g.group, tick);
    const uint32_t explore_at_or_below = coin_threshold(epsilon);
    int32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = g.i0 + j;
        const int32_t pol = policy[i < 0 ? 0 : i < n ? i : n - 1];       // clamped: a lane outside the batch stores nothing
        v[j] = c.w[j] <= explore_at_or_below ? (int32_t)__umulhi(r.w[j], (uint32_t)nvals) : pol;     // u01_24(word B) <= epsilon
    }
    store_group<int32_t>(out, g.i0, n, v);
}

// Box.Sample() of one element (Box.cs:69-90): the reference's four regimes, selected by which bounds are finite.  wa = the
// element's word A; the unbounded regime alone needs a second uniform — word B, fetched through `wb()` only there.
template <class AuxWord>
__device__ __forceinline__ float box_sample_value(float low, float high, uint32_t wa, AuxWord wb) {
    const bool blo = low > -INFINITY, bhi = high < INFINITY;     // Box.CheckBounded (Box.cs:53-58)
    const float u = u01_24(wa);
    if (blo && bhi) return low + (high - low) * u;                // Box.cs:85 uniform(low, high)
    if (blo) return -logf(1.0f - u) + low;                        // Box.cs:83 exponential(1) + low
    if (bhi) return -logf(1.0f - u) + high;                       // Box.cs:84 exponential(1) + high (sic)
    const float u1 = (float)((wa >> 8) + 1u) * (1.0f / 16777216.0f);   // (0, 1]
    const float u2 = u01_24(wb());
    return 0.5f + sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);    // Box.cs:82 normal(0.5, 1) (sic)
}

// Box.Sample() with scalar bounds (the four envs' action spaces): one regime for the launch
__global__ __launch_bounds__(256) void sample_box_kernel(float *__restrict__ out, int64_t n, float low, float high,
                                                         uint64_t seed, uint64_t lane_offset, uint64_t tick) {
    const LaneGroup g = my_lane_group(lane_offset);
    if (g.i0 >= n) return;
    const PhiloxWords r = action_group_words<true>(seed, g.group, tick);
    PhiloxWords c{};
    if (!(low > -INFINITY) && !(high < INFINITY)) c = aux_group_words<true>(seed, g.group, tick);     // launch-uniform
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = box_sample_value(low, high, r.w[j], [&]() { return c.w[j]; });
    store_group<float>(out, g.i0, n, v);
}

// Box.Sample() for a Box whose Low / High are ARRAYS (Box.cs:25-51): the regime is chosen PER ELEMENT from that element's own
// bounds (Box.cs:74-85: unbounded / low-bounded / high-bounded / bounded masks), `dim` elements per lane, output row-major
// [count][dim].  Element e of lane i draws words A / B of the stream keyed by seed + e * odd constant: element 0 uses exactly the
// words of the scalar sampler above, so a (1,)-shaped Box samples the same values either way.  One thread per ELEMENT (coalesced
// row-major stores), each making its group's call and keeping its own word: a utility (ObservationSpace.Sample()), not a hot path.
__global__ __launch_bounds__(256) void sample_box_elementwise_kernel(float *__restrict__ out, int64_t n, int32_t dim,
                                                                     const float *__restrict__ low, const float *__restrict__ high,
                                                                     uint64_t seed, uint64_t lane_offset, uint64_t tick) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * dim) return;
    const int64_t i = idx / dim;
    const int32_t e = (int32_t)(idx - i * dim);
    const uint64_t key = seed + (uint64_t)e * 0xD1B54A32D192ED03ull, lane = lane_offset + (uint64_t)i;
    out[idx] = box_sample_value(low[e], high[e], action_word(key, lane, tick), [&]() { return aux_word(key, lane, tick); });
}

// Direct (full-mesh) all-gather of observations, push form: this member's slice [D][N/G] is stored into
// the same offset of every peer's replica buffer.  blockIdx.y selects the peer, so all peers' links carry traffic
// concurrently (xGMI is point-to-point: 7 links x ~153 GB/s, one per peer); each lane moves 16 bytes per trip.  Plain stores:
// the bytes have to leave this GPU anyway, and the kernel boundary is the release the peers' next kernels acquire against.
__global__ __launch_bounds__(256) void push_obs_kernel(const PushArgs a) {
    float *__restrict__ dst = a.dst[blockIdx.y];
    const float *__restrict__ src = a.src;
    const int64_t nvec = a.count >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec_ok) {
        for (int64_t v = i; v < nvec; v += stride)
            reinterpret_cast<f32x4 *>(dst)[v] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src) + v);
        for (int64_t k = (nvec << 2) + i; k < a.count; k += stride) dst[k] = src[k];
    } else {
        for (int64_t k = i; k < a.count; k += stride) dst[k] = src[k];
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
static inline unsigned grid_for(int64_t items, int block) { return (unsigned)((items + block - 1) / block); }
// workgroups of a stand-alone sampler launch: one thread per group of four global lanes (my_lane_group)
static inline unsigned group_grid(int64_t n, uint64_t lane_offset) { return grid_for((n + (int64_t)(lane_offset & 3u) + 3) / 4, 256); }

hipError_t launch_step(int env_id, bool autoreset, bool extras, const StepArgsT<float> &a, LaunchCfg cfg, hipStream_t st) {
    switch (env_id) {
        case 0: return launch_step_cartpole(autoreset, extras, a, cfg, st);
        case 1: return launch_step_pendulum(autoreset, extras, a, cfg, st);
        case 2: return launch_step_mountaincar(autoreset, extras, a, cfg, st);
        case 3: return launch_step_acrobot(autoreset, extras, a, cfg, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_step(int env_id, bool autoreset, bool extras, const StepArgsT<double> &a, LaunchCfg cfg, hipStream_t st) {
    if (env_id != 0) return hipErrorInvalidValue;      // the reference defines float64 arithmetic for CartPole only
    return launch_step_cartpole64(autoreset, extras, a, cfg, st);
}

int describe_step_kernel(int env_id, bool f64, bool autoreset, bool extras, LaunchCfg cfg, int64_t n, char *buf, size_t cap) {
    if (f64) return env_id == 0 ? describe_step_cartpole64(autoreset, extras, cfg, n, buf, cap) : -1;
    switch (env_id) {
        case 0: return describe_step_cartpole(autoreset, extras, cfg, n, buf, cap);
        case 1: return describe_step_pendulum(autoreset, extras, cfg, n, buf, cap);
        case 2: return describe_step_mountaincar(autoreset, extras, cfg, n, buf, cap);
        case 3: return describe_step_acrobot(autoreset, extras, cfg, n, buf, cap);
        default: return -1;
    }
}

void resolved_step_shape(int env_id, bool f64, bool autoreset, bool extras, LaunchCfg cfg, int64_t n, int *vec, int *sequential) {
    *vec = 1; *sequential = 1;
    if (f64) { if (env_id == 0) resolved_shape_cartpole64(autoreset, extras, cfg, n, vec, sequential); return; }
    switch (env_id) {
        case 0: resolved_shape_cartpole(autoreset, extras, cfg, n, vec, sequential); break;
        case 1: resolved_shape_pendulum(autoreset, extras, cfg, n, vec, sequential); break;
        case 2: resolved_shape_mountaincar(autoreset, extras, cfg, n, vec, sequential); break;
        case 3: resolved_shape_acrobot(autoreset, extras, cfg, n, vec, sequential); break;
        default: break;
    }
}

hipError_t launch_rollout_fused(int env_id, bool autoreset, bool extras, const StepArgsT<float> &a, const RolloutArgsT<float> &r, LaunchCfg cfg, hipStream_t st) {
    switch (env_id) {
        case 0: return launch_rollout_cartpole(autoreset, extras, a, r, cfg, st);
        case 1: return launch_rollout_pendulum(autoreset, extras, a, r, cfg, st);
        case 2: return launch_rollout_mountaincar(autoreset, extras, a, r, cfg, st);
        case 3: return launch_rollout_acrobot(autoreset, extras, a, r, cfg, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_rollout_fused(int env_id, bool autoreset, bool extras, const StepArgsT<double> &a, const RolloutArgsT<double> &r, LaunchCfg cfg, hipStream_t st) {
    if (env_id != 0) return hipErrorInvalidValue;
    return launch_rollout_cartpole64(autoreset, extras, a, r, cfg, st);
}

hipError_t launch_gather_episodes(const EpisodeGatherArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(gather_episodes_kernel, dim3(kShards + 1, kGatherSplit), dim3(256), 0, st, a);     // + 1: the overflow segment's row
    return hipGetLastError();
}

hipError_t launch_reset(int env_id, const ResetArgsT<float> &a, hipStream_t st) {
    switch (env_id) {
        case 0: return launch_reset_cartpole(a, st);
        case 1: return launch_reset_pendulum(a, st);
        case 2: return launch_reset_mountaincar(a, st);
        case 3: return launch_reset_acrobot(a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_reset(int env_id, const ResetArgsT<double> &a, hipStream_t st) {
    if (env_id != 0) return hipErrorInvalidValue;
    return launch_reset_cartpole64(a, st);
}

hipError_t launch_resident(int env_id, bool autoreset, bool extras, const StepArgsT<float> &a, const ResetArgsT<float> &r, Mailbox *mb,
                           uint64_t idle_polls, hipStream_t st) {
    switch (env_id) {
        case 0: return launch_resident_cartpole(autoreset, extras, a, r, mb, idle_polls, st);
        case 1: return launch_resident_pendulum(autoreset, extras, a, r, mb, idle_polls, st);
        case 2: return launch_resident_mountaincar(autoreset, extras, a, r, mb, idle_polls, st);
        case 3: return launch_resident_acrobot(autoreset, extras, a, r, mb, idle_polls, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_resident(int env_id, bool autoreset, bool extras, const StepArgsT<double> &a, const ResetArgsT<double> &r, Mailbox *mb,
                           uint64_t idle_polls, hipStream_t st) {
    if (env_id != 0) return hipErrorInvalidValue;
    return launch_resident_cartpole64(autoreset, extras, a, r, mb, idle_polls, st);
}

hipError_t launch_observe(int env_id, const float *state, int64_t sstride, float *obs, int64_t ostride, int64_t n,
                          hipStream_t st) {
    switch (env_id) {
        case 1: return launch_observe_pendulum(state, sstride, obs, ostride, n, st);
        case 3: return launch_observe_acrobot(state, sstride, obs, ostride, n, st);
        default: return hipSuccess;   // aliasing envs: nothing to recompute
    }
}

template <class T>
static hipError_t pack_obs_any(int obs_dim, const T *obs, int64_t stride, T *out, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const dim3 grid(grid_for(n, 256)), blk(256);
    switch (obs_dim) {
        case 2: hipLaunchKernelGGL((pack_obs_kernel<T, 2>), grid, blk, 0, st, obs, stride, out, n); break;
        case 3: hipLaunchKernelGGL((pack_obs_kernel<T, 3>), grid, blk, 0, st, obs, stride, out, n); break;
        case 4: hipLaunchKernelGGL((pack_obs_kernel<T, 4>), grid, blk, 0, st, obs, stride, out, n); break;
        case 6: hipLaunchKernelGGL((pack_obs_kernel<T, 6>), grid, blk, 0, st, obs, stride, out, n); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_pack_obs(int obs_dim, const float *obs, int64_t stride, float *out, int64_t n, hipStream_t st) { return pack_obs_any(obs_dim, obs, stride, out, n, st); }
hipError_t launch_pack_obs(int obs_dim, const double *obs, int64_t stride, double *out, int64_t n, hipStream_t st) { return pack_obs_any(obs_dim, obs, stride, out, n, st); }

hipError_t launch_export_small(int obs_dim, const float *obs, int64_t stride, const float *reward, const uint8_t *done,
                               float *out_obs, float *out_reward, uint8_t *out_done, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const dim3 grid(grid_for(n, 256)), blk(256);
    switch (obs_dim) {
        case 2: hipLaunchKernelGGL((export_small_kernel<float, 2>), grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n); break;
        case 3: hipLaunchKernelGGL((export_small_kernel<float, 3>), grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n); break;
        case 4: hipLaunchKernelGGL((export_small_kernel<float, 4>), grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n); break;
        case 6: hipLaunchKernelGGL((export_small_kernel<float, 6>), grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_export_small(int obs_dim, const double *obs, int64_t stride, const float *reward, const uint8_t *done,
                               double *out_obs, float *out_reward, uint8_t *out_done, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (obs_dim != 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(export_f64_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n);
    return hipGetLastError();
}

hipError_t launch_export_host(int obs_dim, const double *obs, int64_t stride, const float *reward, const uint8_t *done,
                              double *out_obs, float *out_reward, uint8_t *out_done, int64_t n, hipStream_t st) {
    return launch_export_small(obs_dim, obs, stride, reward, done, out_obs, out_reward, out_done, n, st);
}

hipError_t launch_export_host(int obs_dim, const float *obs, int64_t stride, const float *reward, const uint8_t *done,
                              float *out_obs, float *out_reward, uint8_t *out_done, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    auto al = [](const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    const int vec_ok = al(obs, 16) && stride % 4 == 0 && al(reward, 16) && al(done, 4) && (!out_obs || al(out_obs, 16)) &&
                       (!out_reward || al(out_reward, 16)) && (!out_done || al(out_done, 4));
    const dim3 grid(grid_for((n + 3) / 4, 256)), blk(256);
    switch (obs_dim) {
        case 2: hipLaunchKernelGGL(export_host_kernel<2>, grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n, vec_ok); break;
        case 3: hipLaunchKernelGGL(export_host_kernel<3>, grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n, vec_ok); break;
        case 4: hipLaunchKernelGGL(export_host_kernel<4>, grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n, vec_ok); break;
        case 6: hipLaunchKernelGGL(export_host_kernel<6>, grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n, vec_ok); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_fill_i32(int32_t *p, int32_t v, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, p, v, n);
    return hipGetLastError();
}

hipError_t launch_compact_done(const CompactArgsT<float> &a, hipStream_t st) {
    hipLaunchKernelGGL(compact_done_kernel<float>, dim3(kShards), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_compact_done(const CompactArgsT<double> &a, hipStream_t st) {
    hipLaunchKernelGGL(compact_done_kernel<double>, dim3(kShards), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_validate_discrete(const int32_t *a, int64_t n, int32_t nvals, uint32_t *bad, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(validate_discrete_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, a, n, nvals, bad);
    return hipGetLastError();
}

hipError_t launch_sample_discrete(int32_t *out, int64_t n, int32_t nvals, int32_t start, uint64_t seed,
                                  uint64_t lane_offset, uint64_t tick, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(sample_discrete_kernel, dim3(group_grid(n, lane_offset)), dim3(256), 0, st, out, n, nvals, start, seed,
                       lane_offset, tick);
    return hipGetLastError();
}

hipError_t launch_sample_discrete_masked(int32_t *out, int64_t n, int32_t nvals, int32_t start, const uint8_t *mask,
                                         int64_t mask_stride, uint64_t seed, uint64_t lane_offset, uint64_t tick, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(sample_discrete_masked_kernel, dim3(group_grid(n, lane_offset)), dim3(256), 0, st, out, n, nvals, start, mask,
                       mask_stride, seed, lane_offset, tick);
    return hipGetLastError();
}

hipError_t launch_push_obs(const PushArgs &a, hipStream_t st) {
    if (a.count <= 0 || a.npeers <= 0) return hipSuccess;
    // enough workgroups per peer to keep a link busy, few enough that G-1 peers do not oversubscribe the chip
    int64_t per_peer = (a.count / 4 + 255) / 256;
    if (per_peer > 256) per_peer = 256;
    if (per_peer < 1) per_peer = 1;
    hipLaunchKernelGGL(push_obs_kernel, dim3((unsigned)per_peer, (unsigned)a.npeers), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_sample_box_elementwise(float *out, int64_t n, int32_t dim, const float *low, const float *high, uint64_t seed,
                                         uint64_t lane_offset, uint64_t tick, hipStream_t st) {
    if (n <= 0 || dim <= 0) return hipSuccess;
    hipLaunchKernelGGL(sample_box_elementwise_kernel, dim3(grid_for(n * dim, 256)), dim3(256), 0, st, out, n, dim, low, high, seed,
                       lane_offset, tick);
    return hipGetLastError();
}

hipError_t launch_compose_discrete(const int32_t *policy, int32_t *out, int64_t n, int32_t nvals, float epsilon, uint64_t seed,
                                   uint64_t lane_offset, uint64_t tick, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(compose_discrete_kernel, dim3(group_grid(n, lane_offset)), dim3(256), 0, st, policy, out, n, nvals, epsilon, seed,
                       lane_offset, tick);
    return hipGetLastError();
}

hipError_t launch_sample_box(float *out, int64_t n, float low, float high, uint64_t seed, uint64_t lane_offset,
                             uint64_t tick, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(sample_box_kernel, dim3(group_grid(n, lane_offset)), dim3(256), 0, st, out, n, low, high, seed,
                       lane_offset, tick);
    return hipGetLastError();
}

}  // namespace gymnet
