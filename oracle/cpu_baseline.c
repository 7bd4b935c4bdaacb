/*
 * oracle/cpu_baseline.c — timed CPU baseline: the per-instance float64 CartPole path.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py `cpu_baseline` leg, kind = "port").
 * The reference's C# cannot run in this image (no .NET), so this is a C restatement of how the
 * reference executes the path on a CPU, labelled "port" everywhere it is reported:
 *   - one heap object per environment instance, stepped ONE AT A TIME in a sequential map
 *     (src/Gym/Envs/VecEnvWrapper.cs:22-24: Environments.Select(e => e.Step(action)));
 *   - alloc_faithful = 1 additionally heap-allocates a fresh 4-double state array and a result
 *     record every step, like `np.array(x, x_dot, theta, theta_dot)` and `new Step(...)`
 *     (src/Gym.Environments/Envs/Classic/CartPoleEnv.cs:166,185);
 *   - reset-on-done by the caller (README.md:34-47 loop shape), uniform(-0.05,0.05)^4
 *     (CartPoleEnv.cs:63-67);
 *   - envs are split in contiguous blocks over `threads` POSIX threads (the reference itself is
 *     single-threaded per VecEnv; using all cores is generous to it).
 * The arithmetic is ref_cartpole_step_f64 from classic_control_ref.c.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

int ref_cartpole_step_f64(double *state, int action, int *sbd, float *reward);

typedef struct {
    double *state;     /* heap array of 4 doubles, like the NDArray the env holds */
    int sbd;
} cp_env;

typedef struct {
    double *obs;       /* aliases env state, like Step.Observation */
    float reward;
    int done;
} cp_step_record;

typedef struct {
    int64_t env_begin, env_end, steps;
    int alloc_faithful;
    uint64_t seed;
    double checksum;
    int64_t dones;
} worker_arg;

static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline double u01(uint64_t *s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }

static void env_reset(cp_env *e, uint64_t *rng) {
    e->sbd = -1;
    for (int k = 0; k < 4; ++k) e->state[k] = -0.05 + 0.1 * u01(rng);
}

static void *worker(void *p) {
    worker_arg *w = (worker_arg *)p;
    int64_t n = w->env_end - w->env_begin;
    cp_env *envs = (cp_env *)malloc((size_t)n * sizeof(cp_env));
    uint64_t rng = w->seed ^ (0xA0761D6478BD642Full * (uint64_t)(w->env_begin + 1));
    for (int64_t i = 0; i < n; ++i) {
        envs[i].state = (double *)malloc(4 * sizeof(double));
        env_reset(&envs[i], &rng);
    }
    double checksum = 0.0;
    int64_t dones = 0;
    for (int64_t t = 0; t < w->steps; ++t) {
        uint64_t bits = 0; int nbits = 0;
        for (int64_t i = 0; i < n; ++i) {
            if (nbits == 0) { bits = splitmix64(&rng); nbits = 64; }
            int action = (int)(bits & 1u); bits >>= 1; --nbits;
            cp_env *e = &envs[i];
            float reward; int done;
            if (w->alloc_faithful) {
                double *ns = (double *)malloc(4 * sizeof(double));
                ns[0] = e->state[0]; ns[1] = e->state[1]; ns[2] = e->state[2]; ns[3] = e->state[3];
                done = ref_cartpole_step_f64(ns, action, &e->sbd, &reward);
                free(e->state);
                e->state = ns;
                cp_step_record *r = (cp_step_record *)malloc(sizeof(cp_step_record));
                r->obs = ns; r->reward = reward; r->done = done;
                checksum += r->reward;
                done = r->done;
                free(r);
            } else {
                done = ref_cartpole_step_f64(e->state, action, &e->sbd, &reward);
                checksum += reward;
            }
            if (done) { ++dones; env_reset(e, &rng); }
        }
    }
    for (int64_t i = 0; i < n; ++i) { checksum += envs[i].state[0]; free(envs[i].state); }
    free(envs);
    w->checksum = checksum;
    w->dones = dones;
    return 0;
}

/* Runs n_envs per-instance environments for `steps` steps on `threads` threads.
 * Returns wall seconds of the stepping phase (thread create/join included, env construction too —
 * negligible for steps >= 16).  *env_steps = n_envs*steps; *checksum defeats dead-code removal. */
double ref_cpu_baseline_run(int64_t n_envs, int64_t steps, int threads, int alloc_faithful,
                            uint64_t seed, int64_t *env_steps, double *checksum, int64_t *dones) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t tid[256];
    worker_arg args[256];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int k = 0; k < threads; ++k) {
        args[k].env_begin = n_envs * k / threads;
        args[k].env_end = n_envs * (k + 1) / threads;
        args[k].steps = steps;
        args[k].alloc_faithful = alloc_faithful;
        args[k].seed = seed;
        args[k].checksum = 0.0;
        args[k].dones = 0;
        pthread_create(&tid[k], 0, worker, &args[k]);
    }
    double cs = 0.0; int64_t dn = 0;
    for (int k = 0; k < threads; ++k) { pthread_join(tid[k], 0); cs += args[k].checksum; dn += args[k].dones; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (env_steps) *env_steps = n_envs * steps;
    if (checksum) *checksum = cs;
    if (dones) *dones = dn;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
