// GpuEnv.cs — the reference's SINGLE-INSTANCE environments (Env, src/Gym/Envs/Env.cs:13-41) on the HIP engine: a 1-lane handle
// behind the same members, so the reference's own loops run unmodified with `new GpuCartPoleEnv()` in place of
// `new CartPoleEnv(...)`:
//     README.md:32-52                                       Reset(); Step(i % 2); if (done) Reset(); Close
//     tests/Gym.Tests/Envs/Classic/CartpoleEnvironment.cs:14-35   1000 x (Reset-if-done else Step(i % 2))
// UNVERIFIED: never compiled (no .NET toolchain in the build image).  tests/test_host_api.py lexes it, checks every native
// call against the header's arity, and that the four classes derive from Env and override every abstract member.
//
// What is NOT here: rendering (Render returns null — NullEnvViewer semantics; the viewers are out of scope).
// Dtype: GpuCartPoleEnv defaults to float64 = true — GymnetFlags.F64, the reference's own arithmetic (float64 state, the literal
// CartPoleEnv.cs:141-167 sequence) and the float64 observation NDArray the reference actually returns (:166,185; SURVEY F5) — so
// the loops above see the reference's states to the last few ulps and its exact episode lengths, free-running.  float64 = false
// (and the three envs the reference does not have) return float32, the DECLARED dtype of ObservationSpace (CartPoleEnv.cs:48).
using System;
using System.Threading.Tasks;
using Gym.Collections;
using Gym.Observations;
using Gym.Spaces;
using NumSharp;
using SixLabors.ImageSharp;

namespace Gym.Envs.Amd {
    /// Env (Env.cs:13-41) over ONE lane of the engine.  Reference-faithful mode: no auto-reset, so Step after done returns
    /// reward 0 like CartPoleEnv.cs:176-183 (the console warning is a counter: gymnet_vecenv_counters).
    public abstract unsafe class GpuEnv : Env {
        private IntPtr _h;
        private readonly int _obsDim;
        private readonly bool _boxAction;
        private readonly bool _f64;                          // GymnetFlags.F64: observations are double (CartPole only)
        private readonly float[] _obs;                       // reused: one observation row (float32 handles)
        private readonly double[] _obs64;                    // reused: one observation row (float64 handles)
        private readonly float[] _rew = new float[1];
        private readonly byte[] _done = new byte[1];

        /// maxEpisodeSteps > 0 adds the TimeLimit wrapper upstream gym registers with the env (500 / 200; the reference has none,
        /// SURVEY F6): the step that reaches the limit returns Done = true with Information["TimeLimit.truncated"] = true.
        protected GpuEnv(GymnetEnvId env, int device, ulong seed, int maxEpisodeSteps, bool validateActions, bool float64 = false, bool resident = false) {
            Native.Check(Native.gymnet_env_describe((int) env, out GymnetEnvInfo info));
            _obsDim = info.obs_dim; _boxAction = info.action_is_box != 0;
            _f64 = float64;
            _obs = new float[_obsDim];
            _obs64 = new double[_obsDim];
            var lo = new float[_obsDim]; var hi = new float[_obsDim];
            for (int k = 0; k < _obsDim; k++) { lo[k] = info.obs_low[k]; hi[k] = info.obs_high[k]; }
            ObservationSpace = new Box(np.array(lo), np.array(hi), np.float32);                                   // CartPoleEnv.cs:46-48
            ActionSpace = _boxAction ? (Space) new Box(info.action_low, info.action_high, new Shape(1), np.float32) : new Discrete(info.action_n);
            Metadata = new Dict("render.modes", new[] {"human", "rgb_array"}, "video.frames_per_second", 50);     // CartPoleEnv.cs:51
            RewardRange = (info.reward_low, info.reward_high);
            GymnetFlags flags = GymnetFlags.None;
            if (validateActions) flags |= GymnetFlags.ValidateActions;
            if (maxEpisodeSteps > 0) flags |= GymnetFlags.EpisodeStats;
            if (float64) flags |= GymnetFlags.F64;
            // resident (opt-in since round 6): Step / Reset go through the resident kernel's mailbox — no kernel launch, no stream
            // synchronize per call, bit-identical — at the price that a device-wide synchronize elsewhere in the process waits for the
            // kernel's idle timeout (~5 ms): a host that also trains on the GPU keeps the default, one launch per call
            if (resident) flags |= GymnetFlags.Resident;
            var cfg = new GymnetConfig {
                struct_size = (uint) sizeof(GymnetConfig), env_id = (int) env, num_envs = 1, lane_offset = 0,
                device = device, flags = (uint) flags, seed = seed, max_episode_steps = maxEpisodeSteps
            };
            Native.Check(Native.gymnet_vecenv_create(ref cfg, out _h));
        }

        /// the observation row just written by the library, as a fresh NDArray of the handle's dtype (a COPY, like CartPoleEnv.cs:66)
        private NDArray Observation() => _f64 ? np.array((double[]) _obs64.Clone()) : np.array((float[]) _obs.Clone());

        public override NDArray Reset() {                                                                         // CartPoleEnv.cs:63-67
            fixed (float* p = _obs) fixed (double* p64 = _obs64)
                Native.Check(Native.gymnet_vecenv_reset(_h, _f64 ? (void*) p64 : (void*) p));
            return Observation();
        }

        private Step Result() {
            bool truncated = (_done[0] & 2) != 0;
            Dict info = truncated ? new Dict("TimeLimit.truncated", true) : null;
            return new Step(Observation(), _rew[0], _done[0] != 0, info);                                         // Step.cs:15-20
        }

        /// One boxed action, cast like the reference: a Discrete env does `(int) action` (InvalidCastException for anything that
        /// is not a boxed int, CartPoleEnv.cs:138); a Box env takes an NDArray, a float or an int (LunarLanderEnv.cs:581 idiom).
        private void Stage(object action, int* ia, float* fa) {
            if (_boxAction) {
                if (action is NDArray nd) *fa = nd.astype(np.float32).GetSingle(0);
                else *fa = Convert.ToSingle(action);
            } else {
                *ia = action is Enum ? (int) Convert.ChangeType(action, typeof(int)) : (int) action;
            }
        }

        public override Step Step(object action) {                                                                // CartPoleEnv.cs:137-186
            int ia = 0; float fa = 0f;
            Stage(action, &ia, &fa);
            fixed (float* p32 = _obs) fixed (double* p64 = _obs64) fixed (float* pr = _rew) fixed (byte* pd = _done) {
                void* po = _f64 ? (void*) p64 : (void*) p32;
                if (_boxAction) Native.Check(Native.gymnet_vecenv_step(_h, &fa, po, pr, pd));
                else Native.Check(Native.gymnet_vecenv_step(_h, &ia, po, pr, pd));
            }
            return Result();
        }

        /// Env.StepAsync (Env.cs:23-25) on the engine's own queue: the step is queued on the handle's stream before this
        /// returns; the Task completes in gymnet_vecenv_step_wait.  A second StepAsync before the first finished throws the
        /// reference's AlreadySteppingError (AlreadySteppingError.cs:8-10).
        public override Task<Step> StepAsync(object action) {
            int ia = 0; float fa = 0f;
            Stage(action, &ia, &fa);
            if (_boxAction) Native.Check(Native.gymnet_vecenv_step_async(_h, &fa));
            else Native.Check(Native.gymnet_vecenv_step_async(_h, &ia));
            return Task.Run(() => {
                fixed (float* p32 = _obs) fixed (double* p64 = _obs64) fixed (float* pr = _rew) fixed (byte* pd = _done)
                    Native.Check(Native.gymnet_vecenv_step_wait(_h, _f64 ? (void*) p64 : (void*) p32, pr, pd));
                return Result();
            });
        }

        public override Image Render(string mode = "human") => null;              // viewers are out of scope for the engine

        public override void CloseEnvironment() {                                                                 // CartPoleEnv.cs:189-194
            if (_h != IntPtr.Zero) { Native.gymnet_vecenv_destroy(_h); _h = IntPtr.Zero; }
        }

        public override void Seed(int seed) => Native.Check(Native.gymnet_vecenv_seed(_h, (ulong) seed));          // CartPoleEnv.cs:196-198
    }

    /// CartPoleEnv (src/Gym.Environments/Envs/Classic/CartPoleEnv.cs) on the GPU engine.  The viewer-delegate ctor argument of
    /// the reference (CartPoleEnv.cs:54-61) has no counterpart: nothing is rendered.
    public sealed class GpuCartPoleEnv : GpuEnv {
        public GpuCartPoleEnv(int device = 0, ulong seed = 0, int maxEpisodeSteps = 0, bool validateActions = false, bool float64 = true, bool resident = false)
            : base(GymnetEnvId.CartPole, device, seed, maxEpisodeSteps, validateActions, float64, resident) { }
    }

    /// Pendulum-v1 / MountainCar-v0 / Acrobot-v1: unchecked roadmap items of the reference (README.md:69-76), upstream gym semantics.
    public sealed class GpuPendulumEnv : GpuEnv {
        public GpuPendulumEnv(int device = 0, ulong seed = 0, int maxEpisodeSteps = 0, bool resident = false)
            : base(GymnetEnvId.Pendulum, device, seed, maxEpisodeSteps, false, false, resident) { }
    }

    public sealed class GpuMountainCarEnv : GpuEnv {
        public GpuMountainCarEnv(int device = 0, ulong seed = 0, int maxEpisodeSteps = 0, bool validateActions = false, bool resident = false)
            : base(GymnetEnvId.MountainCar, device, seed, maxEpisodeSteps, validateActions, false, resident) { }
    }

    public sealed class GpuAcrobotEnv : GpuEnv {
        public GpuAcrobotEnv(int device = 0, ulong seed = 0, int maxEpisodeSteps = 0, bool validateActions = false, bool resident = false)
            : base(GymnetEnvId.Acrobot, device, seed, maxEpisodeSteps, validateActions, false, resident) { }
    }
}
