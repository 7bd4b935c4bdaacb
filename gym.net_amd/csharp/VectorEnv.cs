// VectorEnv.cs — a VecEnv (src/Gym/Envs/VecEnv.cs:12-93) whose Step/Reset are ONE HIP kernel launch.
// UNVERIFIED: never compiled (no .NET toolchain in the build image).  Drop next to VecEnvWrapper.cs.
using System;
using System.Threading.Tasks;
using Gym.Collections;
using Gym.Observations;
using Gym.Spaces;
using NumSharp;

namespace Gym.Envs.Amd {
    public sealed unsafe class VectorEnv : VecEnv, IDisposable {
        private IntPtr _h;
        private readonly int _obsDim;
        private readonly bool _boxAction;

        public VectorEnv(GymnetEnvId env, int numEnvs, int device = 0, ulong seed = 0, GymnetFlags flags = GymnetFlags.None,
                         long laneOffset = 0)
            : base(numEnvs, MakeObservationSpace(env, out int obsDim), MakeActionSpace(env, out bool box)) {
            _obsDim = obsDim; _boxAction = box;
            var cfg = new GymnetConfig {
                struct_size = (uint) sizeof(GymnetConfig), env_id = (int) env, num_envs = numEnvs, lane_offset = laneOffset,
                device = device, flags = (uint) flags, seed = seed
            };
            Native.Check(Native.gymnet_vecenv_create(ref cfg, out _h));
            Metadata = new Dict("render.modes", new[] {"human", "rgb_array"}, "video.frames_per_second", 50);   // CartPoleEnv.cs:51
            // Environments stays empty: 2^20 IEnv objects are never materialised (SURVEY F7).
        }

        private static Space MakeObservationSpace(GymnetEnvId env, out int obsDim) {
            Native.Check(Native.gymnet_env_describe((int) env, out GymnetEnvInfo i));
            obsDim = i.obs_dim;
            var lo = new float[obsDim]; var hi = new float[obsDim];
            for (int k = 0; k < obsDim; k++) { lo[k] = i.obs_low[k]; hi[k] = i.obs_high[k]; }
            return new Box(np.array(lo), np.array(hi), np.float32);                                               // CartPoleEnv.cs:46-48
        }

        private static Space MakeActionSpace(GymnetEnvId env, out bool box) {
            Native.Check(Native.gymnet_env_describe((int) env, out GymnetEnvInfo i));
            box = i.action_is_box != 0;
            return box ? (Space) new Box(i.action_low, i.action_high, new Shape(1), np.float32) : new Discrete(i.action_n);
        }

        /// IVecEnv.Reset() (IVecEnv.cs:14).  Batched form: one NDArray of shape (N, D).
        public NDArray ResetBatch() {
            var obs = new float[NumberOfEnvironments * _obsDim];
            fixed (float* p = obs) Native.Check(Native.gymnet_vecenv_reset(_h, p));
            return np.array(obs).reshape(NumberOfEnvironments, _obsDim);
        }

        public override NDArray[] Reset() {                                                                       // VecEnvWrapper.cs:18-20
            var all = ResetBatch();
            var rows = new NDArray[NumberOfEnvironments];
            for (int i = 0; i < rows.Length; i++) rows[i] = all[i];
            return rows;
        }

        /// IVecEnv.Step(int) (IVecEnv.cs:15): ONE scalar action broadcast to every lane; Step[] materialised per lane.
        public override Step[] Step(int action) {                                                                 // VecEnvWrapper.cs:22-24
            var (obs, rew, done) = StepBroadcastArrays(action);
            var steps = new Step[NumberOfEnvironments];
            for (int i = 0; i < steps.Length; i++) steps[i] = new Step(obs[i], rew.GetSingle(i), done.GetByte(i) != 0, null);
            return steps;
        }

        public (NDArray obs, NDArray reward, NDArray done) StepBroadcastArrays(int action) {
            int n = NumberOfEnvironments;
            var obs = new float[n * _obsDim]; var rew = new float[n]; var done = new byte[n];
            fixed (float* po = obs) fixed (float* pr = rew) fixed (byte* pd = done)
                Native.Check(Native.gymnet_vecenv_step_broadcast(_h, action, po, pr, pd));
            return (np.array(obs).reshape(n, _obsDim), np.array(rew), np.array(done));
        }

        /// EXTENSION: one action per lane (int32 for Discrete, float32 for Box), array-valued results.
        public (NDArray obs, NDArray reward, NDArray done) Step(NDArray actions) {
            int n = NumberOfEnvironments;
            if (actions.size != n) throw new ArgumentException("Number of actions passed should be equals to number of environments");
            var obs = new float[n * _obsDim]; var rew = new float[n]; var done = new byte[n];
            fixed (float* po = obs) fixed (float* pr = rew) fixed (byte* pd = done) {
                if (_boxAction) { var a = actions.astype(np.float32).ToArray<float>(); fixed (float* pa = a) Native.Check(Native.gymnet_vecenv_step(_h, pa, po, pr, pd)); }
                else { var a = actions.astype(np.int32).ToArray<int>(); fixed (int* pa = a) Native.Check(Native.gymnet_vecenv_step(_h, pa, po, pr, pd)); }
            }
            return (np.array(obs).reshape(n, _obsDim), np.array(rew), np.array(done));
        }

        public new Task<Step[]> StepAsync(int action) => Task.Run(() => Step(action));                          // VecEnv.cs:63-65

        public new void Seed(int seed) => Native.Check(Native.gymnet_vecenv_seed(_h, (ulong) seed));             // VecEnv.cs:44-46 (see DESIGN.md: lanes differ)
        public new void Seed(int[] seed) {                                                                       // VecEnv.cs:48-53
            var s = Array.ConvertAll(seed, x => (ulong) x);
            Native.Check(Native.gymnet_vecenv_seed_lanes(_h, s, s.Length));
        }

        public override void Close() {                                                                            // VecEnvWrapper.cs:26-30
            if (_h != IntPtr.Zero) { Native.gymnet_vecenv_destroy(_h); _h = IntPtr.Zero; }
        }

        public void Dispose() => Close();
    }
}
