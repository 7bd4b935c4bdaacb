/* tests/cpp/abi_stub.c — TEST-ONLY stand-in for libgymnet_amd.so, for the SANITIZER build of the C++ host mirror
 * (tests/test_sanitizers.py: g++ -fsanitize=address,undefined host_mirror_test.cpp abi_stub.c).
 *
 * It is NOT an environment engine and contains no physics, no oracle and no GPU code: every entry point the C++ host classes
 * (include/gymnet_amd.hpp) call is implemented as "touch exactly the bytes the header documents": a handle is a small host struct,
 * outputs are FILLED to their documented sizes with a deterministic pattern, inputs are READ to their documented sizes.  Under
 * AddressSanitizer that turns every buffer the host classes size wrongly (an observation vector of N instead of N * obs_dim floats,
 * a record array shorter than `capacity`, a use of a handle after Close) into a report.  The product never links this file. */
#include <stdlib.h>
#include <string.h>

#include "gymnet_amd.h"

struct gymnet_vecenv { gymnet_config cfg; int obs_dim, state_dim, box; int64_t n; unsigned long long tick; int pending; float *state; float *pin; };
struct gymnet_group { int G; int64_t n; int obs_dim; gymnet_vecenv **m; };

static _Thread_local char g_err[256];
static int fail(int s, const char *msg) { strncpy(g_err, msg, sizeof g_err - 1); return s; }
static const int kObs[4] = {4, 3, 2, 6}, kState[4] = {4, 2, 2, 4};
static volatile unsigned char g_sink;
static void read_all(const void *p, size_t bytes) { const unsigned char *b = (const unsigned char *)p; unsigned char s = 0; for (size_t i = 0; i < bytes; ++i) s ^= b[i]; g_sink = s; }

int gymnet_abi_version(void) { return GYMNET_ABI_VERSION; }
const char *gymnet_status_string(int status) {
    switch (status) {
        case GYMNET_OK: return "ok";
        case GYMNET_ERR_INVALID_ACTION: return "Action is outside of the configured action space.";
        case GYMNET_ERR_ALREADY_STEPPING: return "already running an async step";
        case GYMNET_ERR_NOT_STEPPING: return "not running an async step";
        default: return "error";
    }
}
const char *gymnet_last_error(void) { return g_err; }
int gymnet_device_count(int *count) { if (count) *count = 1; return GYMNET_OK; }   /* the stub pretends to have one device */
int gymnet_env_describe(int env_id, gymnet_env_info *out) {
    if (!out || env_id < 0 || env_id > 3) return fail(GYMNET_ERR_INVALID_ARG, "unknown env_id");
    memset(out, 0, sizeof *out);
    out->struct_size = sizeof *out; out->env_id = env_id; out->obs_dim = kObs[env_id]; out->state_dim = kState[env_id];
    out->obs_aliases_state = env_id == 0 || env_id == 2; out->action_is_box = env_id == 1; out->action_n = env_id == 0 ? 2 : (env_id == 1 ? 0 : 3);
    out->action_low = -2.0f; out->action_high = 2.0f;
    for (int k = 0; k < 8; ++k) { out->obs_low[k] = -1.0f; out->obs_high[k] = 1.0f; out->state_row_in_obs[k] = -1; }
    out->algorithmic_bytes_per_step = env_id == 0 ? 41 : 37; out->traffic_bytes_per_step = out->algorithmic_bytes_per_step;
    return GYMNET_OK;
}
int gymnet_vecenv_create(const gymnet_config *cfg, gymnet_vecenv **out) {
    if (!cfg || !out || cfg->struct_size != sizeof *cfg) return fail(GYMNET_ERR_INVALID_ARG, "bad cfg");
    if (cfg->num_envs <= 0 || cfg->env_id < 0 || cfg->env_id > 3) return fail(GYMNET_ERR_INVALID_ARG, "bad num_envs / env_id");
    gymnet_vecenv *h = (gymnet_vecenv *)calloc(1, sizeof *h);
    h->cfg = *cfg; h->n = cfg->num_envs; h->obs_dim = kObs[cfg->env_id]; h->state_dim = kState[cfg->env_id]; h->box = cfg->env_id == 1;
    h->state = (float *)calloc((size_t)h->n * h->state_dim, sizeof(float));
    *out = h;
    return GYMNET_OK;
}
int gymnet_vecenv_destroy(gymnet_vecenv *h) { if (h) { free(h->state); free(h->pin); free(h); } return GYMNET_OK; }
int gymnet_vecenv_seed(gymnet_vecenv *h, uint64_t seed) { h->cfg.seed = seed; h->tick = 0; return GYMNET_OK; }
int gymnet_vecenv_seed_lanes(gymnet_vecenv *h, const uint64_t *seeds, int64_t count) {
    if (count != h->n) return fail(GYMNET_ERR_INVALID_ARG, "Number of seeds passed should be equals to number of environments");
    read_all(seeds, (size_t)count * 8);
    return GYMNET_OK;
}
static size_t esz(const gymnet_vecenv *h) { return (h->cfg.flags & GYMNET_FLAG_F64) ? 8 : 4; }
static void fill_outputs(gymnet_vecenv *h, void *obs, float *reward, uint8_t *done) {
    if (obs) memset(obs, 0x3c, (size_t)h->n * h->obs_dim * esz(h));        /* exactly [num_envs, obs_dim] elements of the handle's type */
    if (reward) for (int64_t i = 0; i < h->n; ++i) reward[i] = 1.0f;
    if (done) for (int64_t i = 0; i < h->n; ++i) done[i] = (uint8_t)(((uint64_t)i + h->tick) % 5 == 0);
    h->tick += 1;
}
int gymnet_vecenv_reset(gymnet_vecenv *h, void *obs_out) { fill_outputs(h, obs_out, NULL, NULL); return GYMNET_OK; }
int gymnet_vecenv_reset_where(gymnet_vecenv *h, const uint8_t *mask, void *obs_out) { if (mask) read_all(mask, (size_t)h->n); fill_outputs(h, obs_out, NULL, NULL); return GYMNET_OK; }
int gymnet_vecenv_step(gymnet_vecenv *h, const void *actions, void *obs_out, float *reward_out, uint8_t *done_out) {
    if (!actions) return fail(GYMNET_ERR_INVALID_ARG, "actions is null");
    if (h->pending) return fail(GYMNET_ERR_ALREADY_STEPPING, "already running an async step");
    read_all(actions, (size_t)h->n * 4);
    if ((h->cfg.flags & GYMNET_FLAG_VALIDATE_ACTIONS) && !h->box)
        for (int64_t i = 0; i < h->n; ++i) { int32_t a = ((const int32_t *)actions)[i]; if (a < 0 || a >= (h->cfg.env_id == 0 ? 2 : 3)) return fail(GYMNET_ERR_INVALID_ACTION, "Action is outside of the configured action space."); }
    fill_outputs(h, obs_out, reward_out, done_out);
    return GYMNET_OK;
}
int gymnet_vecenv_step_broadcast(gymnet_vecenv *h, int32_t action, void *obs_out, float *reward_out, uint8_t *done_out) {
    if ((h->cfg.flags & GYMNET_FLAG_VALIDATE_ACTIONS) && !h->box && (action < 0 || action >= (h->cfg.env_id == 0 ? 2 : 3))) return fail(GYMNET_ERR_INVALID_ACTION, "Action is outside of the configured action space.");
    fill_outputs(h, obs_out, reward_out, done_out);
    return GYMNET_OK;
}
int gymnet_vecenv_step_async(gymnet_vecenv *h, const void *actions) {
    if (h->pending) return fail(GYMNET_ERR_ALREADY_STEPPING, "already running an async step");
    read_all(actions, (size_t)h->n * 4);
    h->pending = 1;
    return GYMNET_OK;
}
int gymnet_vecenv_step_wait(gymnet_vecenv *h, void *obs_out, float *reward_out, uint8_t *done_out) {
    if (!h->pending) return fail(GYMNET_ERR_NOT_STEPPING, "not running an async step");
    h->pending = 0;
    fill_outputs(h, obs_out, reward_out, done_out);
    return GYMNET_OK;
}
int gymnet_vecenv_host_buffers(gymnet_vecenv *h, void **actions, void **obs, float **reward, uint8_t **done) {
    const size_t a = (size_t)h->n * 4, o = (size_t)h->n * h->obs_dim * esz(h), r = (size_t)h->n * 4, d = (size_t)h->n;
    if (!h->pin) h->pin = (float *)calloc(1, a + o + r + d);
    char *b = (char *)h->pin;
    if (actions) *actions = b;
    if (obs) *obs = b + a;
    if (reward) *reward = (float *)(b + a + o);
    if (done) *done = (uint8_t *)(b + a + o + r);
    return GYMNET_OK;
}
int gymnet_vecenv_get_state(gymnet_vecenv *h, void *state_soa) { memcpy(state_soa, h->state, (size_t)h->n * h->state_dim * 4); return GYMNET_OK; }
int gymnet_vecenv_set_state(gymnet_vecenv *h, const void *state_soa) { memcpy(h->state, state_soa, (size_t)h->n * h->state_dim * 4); return GYMNET_OK; }
int gymnet_vecenv_get_steps_beyond_done(gymnet_vecenv *h, int32_t *out) { for (int64_t i = 0; i < h->n; ++i) out[i] = -1; return GYMNET_OK; }
int gymnet_vecenv_counters(gymnet_vecenv *h, gymnet_counters *out) { memset(out, 0, sizeof *out); out->struct_size = sizeof *out; out->tick = h->tick; return GYMNET_OK; }
int gymnet_vecenv_kernel_name(gymnet_vecenv *h, char *buf, int32_t capacity) { (void)h; if (capacity > 0) { strncpy(buf, "stub", (size_t)capacity - 1); buf[capacity - 1] = 0; } return GYMNET_OK; }
int gymnet_vecenv_set_launch_policy(gymnet_vecenv *h, const gymnet_launch_policy *p) { (void)h; return p && p->struct_size == sizeof *p ? GYMNET_OK : fail(GYMNET_ERR_INVALID_ARG, "policy"); }
int gymnet_vecenv_get_launch_policy(gymnet_vecenv *h, gymnet_launch_policy *out) { (void)h; memset(out, 0, sizeof *out); out->struct_size = sizeof *out; return GYMNET_OK; }
int gymnet_vecenv_get_array(gymnet_vecenv *h, int32_t which, void *out, int64_t bytes) {
    const int64_t want = (which == GYMNET_ARRAY_DONE ? 1 : (which == GYMNET_ARRAY_LANE_SEEDS ? 8 : 4)) * h->n * (which == GYMNET_ARRAY_FINAL_OBS ? h->obs_dim : 1);
    if (which == GYMNET_ARRAY_FINAL_OBS && !(h->cfg.flags & GYMNET_FLAG_FINAL_OBS)) return fail(GYMNET_ERR_UNSUPPORTED, "needs GYMNET_FLAG_FINAL_OBS");
    if (bytes != want) return fail(GYMNET_ERR_INVALID_ARG, "wrong size");
    memset(out, 0, (size_t)bytes);
    return GYMNET_OK;
}
int gymnet_vecenv_set_array(gymnet_vecenv *h, int32_t which, const void *in, int64_t bytes) { (void)h; (void)which; read_all(in, (size_t)bytes); return GYMNET_OK; }
int gymnet_vecenv_done_records(gymnet_vecenv *h, int32_t *lanes_out, float *return_out, int32_t *length_out, void *final_obs_out, int64_t capacity, int64_t *count) {
    int64_t c = h->n / 5 < capacity ? h->n / 5 : capacity;                  /* writes `count` records into arrays of `capacity` */
    for (int64_t k = 0; k < c; ++k) { if (lanes_out) lanes_out[k] = (int32_t)(k * 5); if (return_out) return_out[k] = 9.0f; if (length_out) length_out[k] = 9; }
    if (final_obs_out) memset(final_obs_out, 0x3c, (size_t)c * h->obs_dim * esz(h));
    *count = h->n / 5;
    return GYMNET_OK;
}
int gymnet_vecenv_reset_device(gymnet_vecenv *h) { h->tick += 1; return GYMNET_OK; }
int gymnet_vecenv_step_device(gymnet_vecenv *h, const void *d_actions) { if (!d_actions) return fail(GYMNET_ERR_INVALID_ARG, "d_actions is null"); h->tick += 1; return GYMNET_OK; }
int gymnet_vecenv_rollout_device(gymnet_vecenv *h, const void *d_actions, int64_t steps, int64_t action_stride, int64_t ring) { (void)d_actions; (void)action_stride; (void)ring; h->tick += (unsigned long long)steps; return GYMNET_OK; }
int gymnet_vecenv_sync(gymnet_vecenv *h) { (void)h; return GYMNET_OK; }
int gymnet_vecenv_device_view(gymnet_vecenv *h, gymnet_device_view *out) { memset(out, 0, sizeof *out); out->struct_size = sizeof *out; out->num_envs = h->n; out->obs_dim = h->obs_dim; out->state_dim = h->state_dim; return GYMNET_OK; }

/* ---- groups: G stub members, the host-boundary forms over the whole batch ---- */
int gymnet_group_create(const gymnet_group_config *cfg, gymnet_group **out) {
    if (!cfg || !out || cfg->num_members < 1 || cfg->global_num_envs % cfg->num_members) return fail(GYMNET_ERR_INVALID_ARG, "bad group cfg");
    gymnet_group *g = (gymnet_group *)calloc(1, sizeof *g);
    g->G = cfg->num_members; g->n = cfg->global_num_envs / g->G; g->obs_dim = kObs[cfg->env_id];
    g->m = (gymnet_vecenv **)calloc((size_t)g->G, sizeof *g->m);
    for (int i = 0; i < g->G; ++i) { gymnet_config c; memset(&c, 0, sizeof c); c.struct_size = sizeof c; c.env_id = cfg->env_id; c.num_envs = g->n; c.flags = cfg->flags; gymnet_vecenv_create(&c, &g->m[i]); }
    *out = g;
    return GYMNET_OK;
}
int gymnet_group_destroy(gymnet_group *g) { if (g) { for (int i = 0; i < g->G; ++i) gymnet_vecenv_destroy(g->m[i]); free(g->m); free(g); } return GYMNET_OK; }
int gymnet_group_seed(gymnet_group *g, uint64_t seed) { for (int i = 0; i < g->G; ++i) gymnet_vecenv_seed(g->m[i], seed); return GYMNET_OK; }
int gymnet_group_reset(gymnet_group *g, void *obs_out) { for (int i = 0; i < g->G; ++i) fill_outputs(g->m[i], obs_out ? (char *)obs_out + (size_t)i * g->n * g->obs_dim * 4 : NULL, NULL, NULL); return GYMNET_OK; }
int gymnet_group_step(gymnet_group *g, const void *actions, void *obs_out, float *reward_out, uint8_t *done_out) {
    read_all(actions, (size_t)g->G * g->n * 4);
    for (int i = 0; i < g->G; ++i) fill_outputs(g->m[i], obs_out ? (char *)obs_out + (size_t)i * g->n * g->obs_dim * 4 : NULL, reward_out ? reward_out + i * g->n : NULL, done_out ? done_out + i * g->n : NULL);
    return GYMNET_OK;
}
int gymnet_group_allgather_obs(gymnet_group *g) { (void)g; return GYMNET_OK; }
int gymnet_group_wait_gather(gymnet_group *g) { (void)g; return GYMNET_OK; }
int gymnet_group_sync(gymnet_group *g) { (void)g; return GYMNET_OK; }
int gymnet_group_read_replica(gymnet_group *g, int32_t member, void *replica_out) { (void)member; memset(replica_out, 0x3c, (size_t)g->G * g->obs_dim * g->n * 4); return GYMNET_OK; }
