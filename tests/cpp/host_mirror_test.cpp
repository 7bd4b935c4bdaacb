// tests/cpp/host_mirror_test.cpp — exercises include/gymnet_amd.hpp (the compiled host mirror of the reference
// interface) against libgymnet_amd.so.  `--cpu`: space / error semantics, no GPU needed (mirrors
// tests/Gym.Tests/Spaces/BoxTest.cs:14-42).  `--gpu`: the reference's own loop shapes
// (tests/Gym.Tests/Envs/Classic/CartpoleEnvironment.cs:19-27, README.md:32-52) and parity checks through the
// C++ classes.  Exit code 0 = all checks passed.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "gymnet_amd.hpp"

static int g_failed = 0;
#define CHECK(cond, msg)                                                          \
    do {                                                                          \
        if (!(cond)) { std::printf("FAIL %s:%d  %s  [%s]\n", __FILE__, __LINE__, #cond, msg); ++g_failed; } \
    } while (0)

template <class Ex, class F>
static bool throws(F f) {
    try { f(); } catch (const Ex &) { return true; } catch (...) { return false; }
    return false;
}

static void cpu_tests() {
    using gymnet::Box;
    const float inf = std::numeric_limits<float>::infinity();
    {   // BoxTest.TestBoxBoundedTest (BoxTest.cs:14-33)
        Box a(-5.0f, 5.0f);
        CHECK(a.IsBounded(Box::BoundedManner::Both), "Box should be bounded at both boundaries.");
        Box b(-inf, 5.0f);
        CHECK(!b.IsBounded(Box::BoundedManner::Below) && b.IsBounded(Box::BoundedManner::Above) && !b.IsBounded(), "unbound low");
        Box c(5.0f, inf);
        CHECK(!c.IsBounded(Box::BoundedManner::Above) && c.IsBounded(Box::BoundedManner::Below) && !c.IsBounded(), "unbound high");
        Box d(-inf, inf);
        CHECK(!d.IsBounded(Box::BoundedManner::Above) && !d.IsBounded(Box::BoundedManner::Below) && !d.IsBounded(), "unbounded");
    }
    {   // BoxTest.TestBoxBoundedSampling (BoxTest.cs:36-41)
        Box box(-5.0f, 5.0f, 1, 3);
        for (int i = 0; i < 200; ++i) { float s = box.Sample()[0]; CHECK(s >= -5.0f && s <= 5.0f, "Box sampling should be on the range [-5.0,5.0]"); }
        Box lo(std::vector<float>{2.0f, -inf}, std::vector<float>{inf, 7.0f}, 1);
        for (int i = 0; i < 200; ++i) { auto s = lo.Sample(); CHECK(s[0] >= 2.0f && s[1] >= 7.0f, "one-sided regimes as the reference writes them (Box.cs:83-84)"); }
        CHECK(lo.Contains({3.0f, 1.0f}) && !lo.Contains({1.0f, 1.0f}), "Box.Contains");
    }
    {   // Discrete.Contains (Discrete.cs:38-40)
        gymnet::Discrete d(2, 5);
        CHECK(d.Contains(0) && d.Contains(1) && !d.Contains(2) && !d.Contains(-1), "0 <= x < N");
        bool saw0 = false, saw1 = false;
        for (int i = 0; i < 64; ++i) { int s = d.Sample(); saw0 |= s == 0; saw1 |= s == 1; CHECK(s == 0 || s == 1, "sample in range"); }
        CHECK(saw0 && saw1, "both values sampled");
        CHECK(gymnet::Discrete(3, 1, 10).Sample({0, 0, 1}) == 12 && gymnet::Discrete(3, 1, 10).Sample({0, 0, 0}) == 10, "masked sample");
    }
    CHECK(std::string(gymnet_status_string(GYMNET_ERR_INVALID_ACTION)) == "Action is outside of the configured action space.", "InvalidActionError text");
    CHECK(std::string(gymnet::AlreadySteppingError().what()) == "already running an async step", "AlreadySteppingError text");
    CHECK(gymnet_abi_version() == GYMNET_ABI_VERSION, "abi version");
    gymnet_env_info info{};
    gymnet::check(gymnet_env_describe(GYMNET_ENV_CARTPOLE, &info));
    CHECK(info.obs_dim == 4 && info.action_n == 2 && info.algorithmic_bytes_per_step == 41, "CartPole description");
    CHECK(throws<std::invalid_argument>([] { gymnet_env_info i{}; gymnet::check(gymnet_env_describe(42, &i)); }), "bad env id -> ArgumentException");
    int ndev = 0;
    if (gymnet_device_count(&ndev) != GYMNET_OK) {
        // no GPU here: the engine must refuse loudly, never fall back to a CPU path
        CHECK(ndev == 0, "count reported as 0");
        CHECK(throws<gymnet::NoDeviceError>([] { gymnet::VectorEnv e(GYMNET_ENV_CARTPOLE, 16); }), "no device -> NoDeviceError");
        CHECK(throws<gymnet::NoDeviceError>([] { gymnet::CartPoleEnv e; }), "no device -> NoDeviceError (single env)");
    }
}

// CartPoleEnv.Step restated in double for one state (CartPoleEnv.cs:24-36,146-167) — the test's own closed form
static void ref_step(const double s[4], int action, double out[4], bool &done) {
    const double g = (double)9.8f, mp = (double)0.1f, M = (double)(0.1f + 1.0f), L = 0.5, pml = (double)(0.1f * 0.5f);
    const double tau = (double)0.02f, xt = (double)2.4f, tt = (double)(float)(24.0 * M_PI / 360.0);
    const double force = action == 1 ? 10.0 : -10.0, c = std::cos(s[2]), sn = std::sin(s[2]);
    const double temp = (force + pml * s[3] * s[3] * sn) / M;
    const double thetaacc = (g * sn - c * temp) / (L * (4.0 / 3.0 - mp * c * c / M));
    const double xacc = temp - pml * thetaacc * c / M;
    out[0] = s[0] + tau * s[1]; out[1] = s[1] + tau * xacc; out[2] = s[2] + tau * s[3]; out[3] = s[3] + tau * thetaacc;
    done = out[0] < -xt || out[0] > xt || out[2] < -tt || out[2] > tt;
}

static void gpu_tests() {
    {   // the reference's test loop: 1000 x (Reset if done else Step(i % 2)), CartpoleEnvironment.cs:19-27
        gymnet::CartPoleEnv cp(0, 1234);
        bool done = true;
        int episodes = 0, steps = 0;
        for (int i = 0; i < 1000; ++i) {
            if (done) { auto obs = cp.Reset(); CHECK(obs.size() == 4, "Reset() -> NDArray[4]"); done = false; ++episodes;
                        for (float v : obs) CHECK(v >= -0.05f && v < 0.05f, "reset ~ U(-0.05, 0.05)"); }
            else { gymnet::Step st = cp.Step(i % 2); done = st.Done; ++steps; CHECK(st.Reward == 1.0f && st.Observation.size() == 4, "reward 1 until reset"); }
        }
        CHECK(episodes > 10 && episodes < 80 && steps + episodes == 1000, "alternating-action episodes average ~37 steps");
        cp.CloseEnvironment();
    }
    {   // Env<TAction>.Step(TAction) where TAction : Enum (Env.cs:43-53): an enum-typed action equals its integer value
        enum class Push { Left = 0, Right = 1 };
        gymnet::CartPoleEnv a(0, 7), b(0, 7);
        auto oa = a.Reset(); auto ob = b.Reset();
        CHECK(oa == ob, "same seed, same first observation");
        for (int i = 0; i < 12; ++i) {
            gymnet::Step sa = a.Step(i % 3 == 0 ? Push::Left : Push::Right), sb = b.Step(i % 3 == 0 ? 0 : 1);
            CHECK(sa.Observation == sb.Observation && sa.Done == sb.Done && sa.Reward == sb.Reward, "enum-typed Step == int Step");
        }
        a.CloseEnvironment(); b.CloseEnvironment();
    }
    {   // README.md:32-52 — the 100 000-iteration variant of the same loop (without Render / Thread.Sleep)
        gymnet::CartPoleEnv cp(0, 99);
        bool done = true;
        long episodes = 0, steps = 0;
        for (int i = 0; i < 100000; ++i) {
            if (done) { cp.Reset(); done = false; ++episodes; }
            else { auto st = cp.Step(i % 2); done = st.Done; ++steps; }
        }
        const double mean_len = (double)steps / (double)episodes;
        CHECK(mean_len > 25.0 && mean_len < 50.0, "alternating-action episodes average ~37.5 steps (SURVEY App. C)");
        cp.CloseEnvironment();
    }
    {   // ABI 3 through the compiled mirror: kernel name, pinned host buffers (no staging), compact done records
        const int64_t n = 8192;
        gymnet::VectorEnv a(GYMNET_ENV_CARTPOLE, n, 0, 5, GYMNET_FLAG_AUTORESET | GYMNET_FLAG_DONE_LIST | GYMNET_FLAG_EPISODE_STATS | GYMNET_FLAG_FINAL_OBS);
        gymnet::VectorEnv b(GYMNET_ENV_CARTPOLE, n, 0, 5, GYMNET_FLAG_AUTORESET);
        CHECK(a.KernelName().rfind("step_kernel<CartPole,", 0) == 0 && b.KernelName().find(",true,false,") != std::string::npos, "kernel names");
        auto pin = a.HostBuffers();
        CHECK(pin.actions && pin.obs && pin.reward && pin.done && a.HostBuffers().obs == pin.obs, "pinned buffers are stable");
        a.Reset(); b.Reset();
        std::vector<int32_t> acts((size_t)n);
        std::vector<int32_t> len((size_t)n, 0);
        long finished = 0;
        for (int t = 0; t < 40; ++t) {
            for (int64_t i = 0; i < n; ++i) acts[(size_t)i] = (int32_t)((i * 7 + t * 3) % 2);
            std::memcpy(pin.actions, acts.data(), (size_t)n * 4);
            a.StepInto(pin.actions, pin.obs, pin.reward, pin.done);
            gymnet::BatchStep sb = b.Step(acts);
            CHECK(std::memcmp(pin.obs, sb.Observation.data(), (size_t)n * 16) == 0 && std::memcmp(pin.done, sb.Done.data(), (size_t)n) == 0,
                  "pinned-buffer step == staged step (bookkeeping variant == lean variant)");
            auto rec = a.DoneRecords(true, true);
            long want = 0;
            for (int64_t i = 0; i < n; ++i) { len[(size_t)i] += 1; want += pin.done[i] != 0; }
            CHECK((long)rec.lanes.size() == want && rec.final_obs.size() == rec.lanes.size() * 4, "one record per finished lane");
            for (size_t k = 0; k < rec.lanes.size(); ++k) {
                const int32_t lane = rec.lanes[k];
                CHECK(pin.done[lane] != 0 && rec.episode_length[k] == len[(size_t)lane] && rec.episode_return[k] == (float)len[(size_t)lane], "record = (lane, return, length)");
                CHECK(std::fabs(rec.final_obs[k * 4]) > 2.4f || std::fabs(rec.final_obs[k * 4 + 2]) > 0.2094f, "terminal observation is past a threshold");
                len[(size_t)lane] = 0;
            }
            finished += want;
        }
        CHECK(finished > n / 2, "episodes finished during the run");
    }
    {   // teacher-forced single steps vs the double closed form: |err| <= 1e-5, done exact (north_star bar)
        const int64_t n = 4096;
        gymnet::VectorEnv env(GYMNET_ENV_CARTPOLE, n, 0, 7);
        env.Reset();
        std::vector<float> s((size_t)4 * n);
        std::mt19937 rng(5);
        std::uniform_real_distribution<float> ux(-2.6f, 2.6f), uv(-2.5f, 2.5f), ut(-0.23f, 0.23f);
        for (int64_t i = 0; i < n; ++i) { s[i] = ux(rng); s[n + i] = uv(rng); s[2 * n + i] = ut(rng); s[3 * n + i] = uv(rng); }
        std::vector<int32_t> a((size_t)n);
        for (auto &v : a) v = (int32_t)(rng() & 1u);
        env.SetState(s);
        gymnet::BatchStep out = env.Step(a);
        auto got = env.GetState();
        double worst = 0; int mism = 0, ndone = 0;
        for (int64_t i = 0; i < n; ++i) {
            const double in[4] = {s[i], s[n + i], s[2 * n + i], s[3 * n + i]};
            double want[4]; bool d;
            ref_step(in, a[i], want, d);
            for (int k = 0; k < 4; ++k) worst = std::fmax(worst, std::fabs((double)got[k * n + i] - want[k]));
            const bool near = std::fabs(std::fabs(want[0]) - (double)2.4f) < 1e-6 || std::fabs(std::fabs(want[2]) - (double)(float)(24.0 * M_PI / 360.0)) < 1e-6;
            if (!near && d != (out.Done[i] != 0)) ++mism;
            ndone += d;
            for (int k = 0; k < 4; ++k) CHECK(out.Observation[i * 4 + k] == got[k * n + i], "observation is the new state, row-major [N,4]");
            CHECK(out.Reward[i] == 1.0f, "first done still pays 1");
        }
        CHECK(worst <= 1e-5, "state within 1e-5 of the float64 reference arithmetic");
        CHECK(mism == 0 && ndone > 0 && ndone < n, "done flags exact");
    }
    {   // steps_beyond_done reward stream (CartPoleEnv.cs:168-183) through IVecEnv.Step(int)
        gymnet::VectorEnv env(GYMNET_ENV_CARTPOLE, 3, 0, 1);
        env.Reset();
        env.SetState({0, 0, 0, 0, 0, 0, 0.05f, 0.05f, 0.05f, 0, 0, 0});
        int after = 0; bool seen = false;
        for (int t = 0; t < 40 && after < 3; ++t) {
            auto out = env.Step(1);
            if (out.Done[0]) { CHECK(out.Reward[0] == (seen ? 0.0f : 1.0f), "1 on the falling step, 0 afterwards"); if (seen) ++after; seen = true; }
            else CHECK(out.Reward[0] == 1.0f && !seen, "1 while alive");
        }
        CHECK(seen && after == 3, "pole fell and was stepped past done");
        CHECK(env.GetStepsBeyondDone()[0] == 3 && env.Counters().stepped_after_done == 9, "sbd counts; warning counted, not printed");
    }
    {   // one process, G members (here: four logical members on device 0): the host-boundary step of the group equals the
        // single batch bit for bit, and after the direct all-gather every member's replica holds all observations
        const int64_t n = 4 * 640;
        gymnet::GroupVectorEnv grp(GYMNET_ENV_CARTPOLE, n, {0, 0, 0, 0}, 11);
        gymnet::VectorEnv one(GYMNET_ENV_CARTPOLE, n, 0, 11, GYMNET_FLAG_AUTORESET);
        CHECK(grp.Reset() == one.Reset(), "group Reset == single batch");
        std::mt19937 rng(9);
        bool same = true;
        gymnet::BatchStep last;
        for (int t = 0; t < 30; ++t) {
            std::vector<int32_t> a((size_t)n);
            for (auto &x : a) x = (int32_t)(rng() & 1u);
            gymnet::BatchStep g = grp.Step(a), o = one.Step(a);
            same = same && g.Observation == o.Observation && g.Reward == o.Reward && g.Done == o.Done;
            last = o;
        }
        CHECK(same, "group Step == single batch over 30 steps with auto-reset");
        grp.AllGatherObs(); grp.WaitGather();
        const int64_t nl = n / 4;
        for (int m = 0; m < 4; ++m) {
            const std::vector<float> rep = grp.ReadReplica(m);                  // [G][4][nl]
            bool ok = true;
            for (int r = 0; r < 4 && ok; ++r)
                for (int k = 0; k < 4 && ok; ++k)
                    for (int64_t i = 0; i < nl && ok; ++i)
                        ok = rep[((size_t)r * 4 + k) * nl + i] == last.Observation[((size_t)r * nl + i) * 4 + k];
            CHECK(ok, "every member's replica holds all observations after the direct all-gather");
        }
        CHECK(throws<std::invalid_argument>([] { gymnet::GroupVectorEnv bad(GYMNET_ENV_CARTPOLE, 1001, {0, 0}); }), "N not a multiple of G -> ArgumentException");
    }
    {   // ABI 4: the reference-exact float64 single instance; the float32 one beside it stays within 1e-5 per teacher-forced step
        gymnet::CartPoleEnv64 e64(0, 11);
        gymnet::CartPoleEnv e32(0, 11);
        auto o = e64.Reset();
        CHECK(o.size() == 4 && std::fabs(o[0]) < 0.05 && std::fabs(o[3]) < 0.05, "float64 Reset ~ U(-0.05, 0.05)^4");
        e32.Reset();
        bool close = true, done_same = true;
        int steps = 0;
        for (int i = 0; i < 200; ++i) {
            const std::vector<double> s = e64.GetState();
            e32.vector().SetState(std::vector<float>{(float)s[0], (float)s[1], (float)s[2], (float)s[3]});     // teacher-forced
            e64.SetState({(double)(float)s[0], (double)(float)s[1], (double)(float)s[2], (double)(float)s[3]});
            const gymnet::Step64 a = e64.Step(i % 2);
            const gymnet::Step b = e32.Step(i % 2);
            for (int k = 0; k < 4; ++k) close = close && std::fabs(a.Observation[k] - (double)b.Observation[k]) <= 1e-5;
            done_same = done_same && a.Done == b.Done && a.Reward == b.Reward;
            ++steps;
            if (a.Done) { e64.Reset(); e32.Reset(); }
        }
        CHECK(close && done_same && steps == 200, "float32 engine within 1e-5 of the float64 one per teacher-forced step, same done / reward");
        {   // ABI 5: the float64 mode combines with every other flag (one kernel skeleton for both state scalars)
            gymnet::CartPoleEnv64 full(0, 1, GYMNET_FLAG_DONE_LIST | GYMNET_FLAG_AUTORESET | GYMNET_FLAG_FINAL_OBS | GYMNET_FLAG_EPISODE_STATS | GYMNET_FLAG_DOUBLE_BUFFER);
            full.Reset();
            bool finished = false;
            for (int i = 0; i < 200 && !finished; ++i) finished = full.Step(1).Done;
            CHECK(finished, "F64 + DONE_LIST + FINAL_OBS + EPISODE_STATS + DOUBLE_BUFFER: steps, and the always-right episode ends");
        }
        CHECK(throws<std::logic_error>([] { gymnet::VectorEnv bad(GYMNET_ENV_CARTPOLE, 8, 0, 1, GYMNET_FLAG_F64); }), "the float32 host class refuses a float64 handle (its buffers are float)");
    }
    {   // ABI 4: launch policy through the ABI, arrays by id
        gymnet::VectorEnv env(GYMNET_ENV_CARTPOLE, 4096, 0, 5, GYMNET_FLAG_AUTORESET | GYMNET_FLAG_EPISODE_STATS);
        gymnet_launch_policy p = gymnet::VectorEnv::KeepPolicy();
        p.vec = 4; p.nt = 12; p.reset_form = 1;
        env.SetLaunchPolicy(p);
        CHECK(env.KernelName() == "step_kernel<CartPole,4,true,true,12,1>", "set_launch_policy selects the kernel form");
        p = gymnet::VectorEnv::KeepPolicy(); p.vec = 2;
        CHECK(throws<std::invalid_argument>([&] { env.SetLaunchPolicy(p); }), "vec = 2 is Acrobot's form -> ArgumentException");
        CHECK(env.GetLaunchPolicy().vec == 4, "a refused policy changes nothing");
        env.Reset();
        for (int t = 0; t < 30; ++t) env.Step(t % 2);
        auto len = env.GetArray<int32_t>(GYMNET_ARRAY_EPISODE_LENGTH, 4096);
        auto ret = env.GetArray<float>(GYMNET_ARRAY_EPISODE_RETURN, 4096);
        bool ok = true;
        for (size_t i = 0; i < len.size(); ++i) ok = ok && len[i] >= 0 && len[i] <= 30 && ret[i] == (float)len[i];
        CHECK(ok, "running episode return == length for CartPole (reward 1 per step)");
        CHECK(throws<std::logic_error>([&] { env.GetArray<float>(GYMNET_ARRAY_FINAL_OBS, 4 * 4096); }), "array the configuration lacks -> NotSupportedException");
        CHECK(throws<std::invalid_argument>([&] { env.GetArray<float>(GYMNET_ARRAY_REWARD, 17); }), "wrong size -> ArgumentException");
    }
    {   // error behaviour
        gymnet::VectorEnv env(GYMNET_ENV_CARTPOLE, 8, 0, 1, GYMNET_FLAG_VALIDATE_ACTIONS);
        env.Reset();
        CHECK(throws<gymnet::InvalidActionError>([&] { env.Step(2); }), "InvalidActionError (LunarLanderEnv.cs:604-607 convention)");
        CHECK(throws<std::invalid_argument>([&] { env.Seed(std::vector<int>{1, 2, 3}); }), "seed count mismatch -> ArgumentException (VecEnv.cs:49)");
        CHECK(throws<std::invalid_argument>([&] { env.Step(std::vector<int32_t>{1, 0}); }), "action count mismatch");
        CHECK(throws<gymnet::NotSteppingError>([&] { env.StepWait(); }), "NotSteppingError");
        env.StepAsync(std::vector<int32_t>(8, 1));
        CHECK(throws<gymnet::AlreadySteppingError>([&] { env.StepAsync(std::vector<int32_t>(8, 0)); }), "AlreadySteppingError");
        auto out = env.StepWait();
        CHECK(out.size() == 8 && out[0].Observation.size() == 4, "StepWait returns the batch");
        CHECK(throws<std::invalid_argument>([] { gymnet::VectorEnv bad(GYMNET_ENV_CARTPOLE, 0); }), "num_envs 0 -> ArgumentException");
    }
}

// `--stub`: the host classes against tests/cpp/abi_stub.c (a test-only, memory-honest stand-in of the C ABI: it fills / reads exactly the
// bytes the header documents and computes nothing), built with -fsanitize=address,undefined by tests/test_sanitizers.py.  What is
// under test is the HOST side: every buffer the classes allocate and hand across the ABI, handle lifetime, error mapping.
static void stub_tests() {
    for (gymnet_env_id env : {GYMNET_ENV_CARTPOLE, GYMNET_ENV_PENDULUM, GYMNET_ENV_MOUNTAINCAR, GYMNET_ENV_ACROBOT}) {
        for (int64_t n : {1, 7, 4097}) {
            gymnet::VectorEnv e(env, n, 0, 3, GYMNET_FLAG_AUTORESET | GYMNET_FLAG_DONE_LIST | GYMNET_FLAG_EPISODE_STATS | GYMNET_FLAG_FINAL_OBS);
            auto obs = e.Reset();
            CHECK((int64_t)obs.size() == n * e.ObsDim(), "Reset() -> N * obs_dim floats");
            gymnet::BatchStep b = env == GYMNET_ENV_PENDULUM ? e.Step(std::vector<float>((size_t)n, 0.5f)) : e.Step(std::vector<int32_t>((size_t)n, 1));
            CHECK(b.size() == n && (int64_t)b.Observation.size() == n * e.ObsDim() && b[n - 1].Observation.size() == (size_t)e.ObsDim(), "BatchStep shapes");
            b = e.Step(0);
            CHECK((int64_t)b.Reward.size() == n, "broadcast step");
            std::vector<uint8_t> mask((size_t)n, 1);
            CHECK((int64_t)e.ResetWhere(&mask).size() == n * e.ObsDim() && (int64_t)e.ResetWhere().size() == n * e.ObsDim(), "ResetWhere");
            auto st = e.GetState();
            CHECK((int64_t)st.size() == n * e.StateDim(), "GetState");
            e.SetState(st);
            auto rec = e.DoneRecords(true, true);
            CHECK(rec.lanes.size() == (size_t)(n / 5) && rec.final_obs.size() == rec.lanes.size() * (size_t)e.ObsDim() && rec.episode_length.size() == rec.lanes.size(), "DoneRecords sizes");
            auto pin = e.HostBuffers();
            std::memset(pin.actions, 0, (size_t)n * 4);
            e.StepInto(pin.actions, pin.obs, pin.reward, pin.done);
            CHECK(pin.reward[n - 1] == 1.0f, "pinned buffers written to their end");
            CHECK(e.GetArray<float>(GYMNET_ARRAY_FINAL_OBS, (size_t)n * (size_t)e.ObsDim()).size() == (size_t)n * (size_t)e.ObsDim(), "GetArray(FINAL_OBS)");
            CHECK(throws<std::invalid_argument>([&] { e.GetArray<float>(GYMNET_ARRAY_REWARD, (size_t)n + 1); }), "wrong size -> ArgumentException");
            CHECK(e.GetStepsBeyondDone().size() == (size_t)n && e.Counters().tick > 0 && e.KernelName() == "stub", "small getters");
            if (env != GYMNET_ENV_PENDULUM) {
                e.StepAsync(std::vector<int32_t>((size_t)n, 0));
                CHECK(throws<gymnet::AlreadySteppingError>([&] { e.StepAsync(std::vector<int32_t>((size_t)n, 0)); }), "AlreadySteppingError");
                CHECK(e.StepWait().size() == n, "StepWait");
                CHECK(throws<gymnet::NotSteppingError>([&] { e.StepWait(); }), "NotSteppingError");
            }
            CHECK(throws<std::invalid_argument>([&] { e.Seed(std::vector<int>((size_t)n + 2, 1)); }), "seed count mismatch -> ArgumentException");
            CHECK(throws<std::invalid_argument>([&] { e.Step(std::vector<int32_t>((size_t)n + 1, 0)); }), "action count mismatch");
            e.Seed(std::vector<int>((size_t)n, 4));
            e.Close();
            e.Close();                                              // idempotent; the handle is not touched again
        }
    }
    {
        gymnet::CartPoleEnv64 e64(0, 1, GYMNET_FLAG_AUTORESET);
        CHECK(e64.Reset().size() == 4 && e64.Step(1).Observation.size() == 4 && e64.GetState().size() == 4, "float64 single env: 4 doubles everywhere");
        gymnet::CartPoleEnv e32(0, 1);
        CHECK(e32.Reset().size() == 4 && e32.Step(0).Observation.size() == 4, "float32 single env");
        CHECK(throws<std::logic_error>([] { gymnet::VectorEnv bad(GYMNET_ENV_CARTPOLE, 8, 0, 1, GYMNET_FLAG_F64); }), "float32 host class refuses a float64 handle");
    }
    {
        gymnet::GroupVectorEnv g(GYMNET_ENV_ACROBOT, 4 * 513, std::vector<int32_t>{0, 0, 0, 0}, 9, GYMNET_FLAG_AUTORESET, GYMNET_GATHER_DIRECT);
        CHECK((int64_t)g.Reset().size() == 4 * 513 * 6, "group Reset over the whole batch");
        gymnet::BatchStep b = g.Step(std::vector<int32_t>(4 * 513, 2));
        CHECK(b.size() == 4 * 513 && b.Observation.size() == (size_t)4 * 513 * 6, "group Step");
        g.AllGatherObs(); g.WaitGather(); g.Sync();
        CHECK(g.ReadReplica(3).size() == (size_t)4 * 6 * 513, "replica [G][D][N/G]");
    }
}

int main(int argc, char **argv) {
    const bool gpu = argc > 1 && std::strcmp(argv[1], "--gpu") == 0;
    const bool stub = argc > 1 && std::strcmp(argv[1], "--stub") == 0;
    try {
        if (stub) { stub_tests(); std::printf("stub: %d failed check(s)\n", g_failed); return g_failed ? 1 : 0; }
        cpu_tests();
        if (gpu) gpu_tests();
    } catch (const std::exception &e) {
        std::printf("FAIL unexpected exception: %s\n", e.what());
        return 2;
    }
    std::printf("%s: %d failed check(s)\n", gpu ? "cpu+gpu" : "cpu", g_failed);
    return g_failed ? 1 : 0;
}
