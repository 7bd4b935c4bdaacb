// VectorEnv.cs — a VecEnv (src/Gym/Envs/VecEnv.cs:12-93) whose Step/Reset are ONE HIP kernel launch.
// UNVERIFIED: never compiled (no .NET toolchain in the build image).  Drop next to VecEnvWrapper.cs.
//
// Dispatch through the reference's own types.  VecEnv.Seed(int), Seed(int[]) and StepAsync(int) are NOT virtual
// (VecEnv.cs:44-65), so a derived class cannot override them, and `public new` members are invisible to callers that hold
// an IVecEnv or a VecEnv.  Two things make the polymorphic callers work:
//   1. the class RE-LISTS IVecEnv and implements Seed(int) / Seed(int[]) explicitly: every call through IVecEnv
//      (IVecEnv.cs:17-18) lands in the native seed functions;
//   2. `Environments` (VecEnv.cs:25, settable) is a virtual list of N lane proxies.  The non-virtual base Seed — what a
//      VecEnv-typed variable calls — walks Environments and calls Environments[i].Seed(seed[i]) (VecEnv.cs:50-52); each proxy
//      records its lane's seed and the last one flushes the batch to gymnet_vecenv_seed_lanes.  So the base path has the
//      reference's semantics (every env seeded with its own value; Seed(int) = the same value for all, VecEnv.cs:44-46) and
//      its ArgumentException on a length mismatch, without 2^20 IEnv objects ever being stored.
using System;
using System.Collections;
using System.Collections.Generic;
using System.Threading.Tasks;
using Gym.Collections;
using Gym.Observations;
using Gym.Spaces;
using NumSharp;
using SixLabors.ImageSharp;

namespace Gym.Envs.Amd {
    public sealed unsafe class VectorEnv : VecEnv, IVecEnv, IDisposable {
        private IntPtr _h;
        private readonly int _obsDim;
        private readonly bool _boxAction;
        private readonly bool _f64;                 // GymnetFlags.F64 (CartPole): every observation buffer holds doubles — the reference's
                                                    // actual dtype (CartPoleEnv.cs:166,185); the library writes 8 bytes per element
        internal ulong[] PendingLaneSeeds;          // filled by the lane proxies when the base-class Seed runs
        internal int PendingLaneSeedCount;

        public VectorEnv(GymnetEnvId env, int numEnvs, int device = 0, ulong seed = 0, GymnetFlags flags = GymnetFlags.None,
                         long laneOffset = 0, int maxEpisodeSteps = 0)
            : base(numEnvs, MakeObservationSpace(env, out int obsDim), MakeActionSpace(env, out bool box)) {
            _obsDim = obsDim; _boxAction = box;
            _f64 = (flags & GymnetFlags.F64) != 0;
            var cfg = new GymnetConfig {
                struct_size = (uint) sizeof(GymnetConfig), env_id = (int) env, num_envs = numEnvs, lane_offset = laneOffset,
                device = device, flags = (uint) flags, seed = seed, max_episode_steps = maxEpisodeSteps      // (needs GymnetFlags.EpisodeStats)
            };
            Native.Check(Native.gymnet_vecenv_create(ref cfg, out _h));
            Metadata = new Dict("render.modes", new[] {"human", "rgb_array"}, "video.frames_per_second", 50);   // CartPoleEnv.cs:51
            Native.Check(Native.gymnet_env_describe((int) env, out GymnetEnvInfo info));
            RewardRange = (info.reward_low, info.reward_high);
            Environments = new LaneList(this);      // virtual: N proxies materialised on demand, never stored (SURVEY F7)
        }

        internal IntPtr Handle => _h;

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

        /// An observation buffer of the handle's dtype (float32, or float64 with GymnetFlags.F64), pinned for one native call.
        /// EVERY call that hands the library an `obs_out` goes through here, so the element size can never disagree with the handle.
        private delegate int ObsCall(void* obs);
        private NDArray WithObs(ObsCall call) {
            int n = NumberOfEnvironments;
            if (_f64) {
                var obs = new double[n * _obsDim];
                fixed (double* p = obs) Native.Check(call(p));
                return np.array(obs).reshape(n, _obsDim);
            } else {
                var obs = new float[n * _obsDim];
                fixed (float* p = obs) Native.Check(call(p));
                return np.array(obs).reshape(n, _obsDim);
            }
        }

        /// IVecEnv.Reset() (IVecEnv.cs:14).  Batched form: one NDArray of shape (N, D), float32 — float64 for an F64 handle.
        public NDArray ResetBatch() {
            return WithObs(p => Native.gymnet_vecenv_reset(_h, p));
        }

        public override NDArray[] Reset() {                                                                       // VecEnvWrapper.cs:18-20
            var all = ResetBatch();
            var rows = new NDArray[NumberOfEnvironments];
            for (int i = 0; i < rows.Length; i++) rows[i] = all[i];
            return rows;
        }

        /// The caller's `if (done) Reset()` (README.md:36-40) for the lanes in mask (null = lanes whose last done flag is set).
        public NDArray ResetWhere(byte[] mask = null) {
            if (mask != null && mask.Length != NumberOfEnvironments) throw new ArgumentException("mask length must equal NumberOfEnvironments");
            return WithObs(p => { fixed (byte* m = mask) return Native.gymnet_vecenv_reset_where(_h, m, p); });
        }

        private Step[] ToSteps(NDArray all, float[] rew, byte[] done) {
            var steps = new Step[NumberOfEnvironments];
            for (int i = 0; i < steps.Length; i++) steps[i] = new Step(all[i], rew[i], done[i] != 0, null);       // Step.cs:15-20
            return steps;
        }

        /// IVecEnv.Step(int) (IVecEnv.cs:15): ONE scalar action broadcast to every lane; Step[] materialised per lane.
        public override Step[] Step(int action) {                                                                 // VecEnvWrapper.cs:22-24
            int n = NumberOfEnvironments;
            var rew = new float[n]; var done = new byte[n];
            var obs = WithObs(po => { fixed (float* pr = rew) fixed (byte* pd = done) return Native.gymnet_vecenv_step_broadcast(_h, action, po, pr, pd); });
            return ToSteps(obs, rew, done);
        }

        /// Env<TAction>.Step(TAction) (Env.cs:43-53) for enum-typed discrete actions: the enum's integer value.
        public Step[] Step<TAction>(TAction action) where TAction : Enum => Step((int) (object) action);

        /// EXTENSION: one action per lane (int32 for Discrete, float32 for Box), array-valued results.
        public (NDArray obs, NDArray reward, NDArray done) Step(NDArray actions) {
            int n = NumberOfEnvironments;
            if (actions.size != n) throw new ArgumentException("Number of actions passed should be equals to number of environments");
            var rew = new float[n]; var done = new byte[n];
            var obs = WithObs(po => {
                fixed (float* pr = rew) fixed (byte* pd = done) {
                    if (_boxAction) { var a = actions.astype(np.float32).ToArray<float>(); fixed (float* pa = a) return Native.gymnet_vecenv_step(_h, pa, po, pr, pd); }
                    else { var a = actions.astype(np.int32).ToArray<int>(); fixed (int* pa = a) return Native.gymnet_vecenv_step(_h, pa, po, pr, pd); }
                }
            });
            return (obs, np.array(rew), np.array(done));
        }

        /// ABI 3: the library's page-locked, device-mapped host buffers (valid until Close): actions int32 / float32 [N], obs
        /// float32 [N, D] row-major (float64 for an F64 handle), reward float32 [N], done uint8 [N].  A caller that keeps its NDArrays over this memory
        /// (or reads it through spans) steps with StepPinned(): no managed arrays, no staging copies — the export kernel writes
        /// the results straight across PCIe (0.50 ms per 2^20-lane step instead of 0.57-0.58 ms through pageable arrays).
        public struct PinnedBuffers { public IntPtr Actions; public IntPtr Obs; public IntPtr Reward; public IntPtr Done; }

        public PinnedBuffers HostBuffers() {
            Native.Check(Native.gymnet_vecenv_host_buffers(_h, out IntPtr a, out IntPtr o, out IntPtr r, out IntPtr d));
            return new PinnedBuffers { Actions = a, Obs = o, Reward = r, Done = d };
        }

        /// One vector step over the pinned buffers: reads HostBuffers().Actions, fills Obs / Reward / Done.  Blocks until they are written.
        public void StepPinned() {
            var b = HostBuffers();
            Native.Check(Native.gymnet_vecenv_step(_h, (void*) b.Actions, (void*) b.Obs, (float*) b.Reward, (byte*) b.Done));
        }

        /// ABI 4: override fields of the step kernel's launch configuration (every field of `policy` that is -1 stays as it is;
        /// start from KeepPolicy()).  All configurations compute bit-identical results; a value this handle cannot run throws
        /// ArgumentException.  This is the ONLY way to steer the policy: the library does not read the process environment.
        public static GymnetLaunchPolicy KeepPolicy() => new GymnetLaunchPolicy {
            struct_size = (uint) sizeof(GymnetLaunchPolicy), vec = -1, block = -1, nt = -1, sequential_lanes = -1, reset_form = -1,
            lds_pipe = -1, occupancy_lds_bytes = -1, graph = -1
        };

        public void SetLaunchPolicy(GymnetLaunchPolicy policy) {
            policy.struct_size = (uint) sizeof(GymnetLaunchPolicy);
            Native.Check(Native.gymnet_vecenv_set_launch_policy(_h, ref policy));
        }

        /// ABI 5: `steps` vector steps in ONE kernel launch with what the consumer needs (BasePlaySession.cs:58-69, ReplayMemory.cs:53-67,
        /// TrainingPlaySession.cs:46-52): actions from a device ring, drawn in the kernel (action_source 1: ActionSpace.Sample()) or
        /// epsilon-greedy over the ring (2); dense recording; compact (step, lane, return, length) records of the episodes that end.
        /// All pointers in `spec` are DEVICE pointers; stream-ordered, does not block.
        public void ResetDevice() => Native.Check(Native.gymnet_vecenv_reset_device(_h));      // device-resident path: nothing crosses PCIe
        public void Sync() => Native.Check(Native.gymnet_vecenv_sync(_h));
        public void RolloutFused(GymnetRolloutSpec spec) {
            spec.struct_size = (uint) sizeof(GymnetRolloutSpec);
            Native.Check(Native.gymnet_vecenv_rollout_fused_ex_device(_h, ref spec));
        }

        /// ABI 4: any per-lane array the handle keeps, by id — with GetState / the tick / the seed a complete checkpoint of every
        /// configuration (episode return / length, done flags, per-lane Philox keys, ...).  T must be the array's element type.
        public T[] GetArray<T>(GymnetArrayId which, int count) where T : unmanaged {
            var a = new T[count];
            fixed (T* p = a) Native.Check(Native.gymnet_vecenv_get_array(_h, (int) which, p, (long) count * sizeof(T)));
            return a;
        }

        public void SetArray<T>(GymnetArrayId which, T[] a) where T : unmanaged {
            fixed (T* p = a) Native.Check(Native.gymnet_vecenv_set_array(_h, (int) which, p, (long) a.Length * sizeof(T)));
        }

        /// Compact records of the lanes that finished in the most recent step (GymnetFlags.DoneList [+ EpisodeStats] [+ FinalObs]):
        /// what BasePlaySession.cs:58-69 accumulates per episode, without shipping N flags to the host.
        public (int[] lanes, float[] episodeReturn, int[] episodeLength, float[] finalObs) DoneRecords(bool episode, bool finalObs) {
            int n = NumberOfEnvironments;
            var lanes = new int[n];
            var ret = episode ? new float[n] : null; var len = episode ? new int[n] : null;
            var fo = finalObs ? new float[n * _obsDim] : null;
            long count;
            fixed (int* pl = lanes) fixed (float* pr = ret) fixed (int* pn = len) fixed (float* po = fo)
                Native.Check(Native.gymnet_vecenv_done_records(_h, pl, pr, pn, po, n, out count));
            Array.Resize(ref lanes, (int) count);
            if (episode) { Array.Resize(ref ret, (int) count); Array.Resize(ref len, (int) count); }
            if (finalObs) Array.Resize(ref fo, (int) count * _obsDim);
            return (lanes, ret, len, fo);
        }

        /// VecEnv.StepAsync (VecEnv.cs:63-65) on the native queue: gymnet_vecenv_step_async returns once the step is queued on
        /// the handle's stream; the Task completes in gymnet_vecenv_step_wait.  A second StepAsync before the first finished
        /// surfaces the reference's AlreadySteppingError (AlreadySteppingError.cs:8-10).
        public new Task<Step[]> StepAsync(int action) {
            int n = NumberOfEnvironments;
            if (_boxAction) { var a = new float[n]; for (int i = 0; i < n; i++) a[i] = action; fixed (float* pa = a) Native.Check(Native.gymnet_vecenv_step_async(_h, pa)); }
            else { var a = new int[n]; for (int i = 0; i < n; i++) a[i] = action; fixed (int* pa = a) Native.Check(Native.gymnet_vecenv_step_async(_h, pa)); }
            return Task.Run(() => {
                var rew = new float[n]; var done = new byte[n];
                var obs = WithObs(po => { fixed (float* pr = rew) fixed (byte* pd = done) return Native.gymnet_vecenv_step_wait(_h, po, pr, pd); });
                return ToSteps(obs, rew, done);
            });
        }

        // ---- seeding: explicit IVecEnv implementations (interface callers), `new` members (VectorEnv-typed callers), and the
        //      lane proxies (VecEnv-typed callers, see the header comment)
        void IVecEnv.Seed(int seed) => SeedAll(seed);                                                            // IVecEnv.cs:18
        void IVecEnv.Seed(int[] seed) => SeedLanes(seed);                                                        // IVecEnv.cs:17
        public new void Seed(int seed) => SeedAll(seed);
        public new void Seed(int[] seed) => SeedLanes(seed);

        /// VecEnv.Seed(int) (VecEnv.cs:44-46).  DEVIATION (INTEGRATION.md §0): one Philox key for the batch, lanes differ by counter.
        private void SeedAll(int seed) => Native.Check(Native.gymnet_vecenv_seed(_h, (ulong) seed));

        private void SeedLanes(int[] seed) {                                                                     // VecEnv.cs:48-53
            if (seed == null) throw new ArgumentNullException(nameof(seed));
            var s = Array.ConvertAll(seed, x => (ulong) x);
            Native.Check(Native.gymnet_vecenv_seed_lanes(_h, s, s.Length));     // length mismatch -> ArgumentException (VecEnv.cs:49)
        }

        internal void SeedLaneFromProxy(int lane, int seed) {
            if (PendingLaneSeeds == null) { PendingLaneSeeds = new ulong[NumberOfEnvironments]; PendingLaneSeedCount = 0; }
            PendingLaneSeeds[lane] = (ulong) seed;
            if (++PendingLaneSeedCount == NumberOfEnvironments) {                // the base Seed loop reached the last env
                var s = PendingLaneSeeds; PendingLaneSeeds = null; PendingLaneSeedCount = 0;
                Native.Check(Native.gymnet_vecenv_seed_lanes(_h, s, s.Length));
            }
        }

        public override void Close() {                                                                            // VecEnvWrapper.cs:26-30
            if (_h != IntPtr.Zero) { Native.gymnet_vecenv_destroy(_h); _h = IntPtr.Zero; }
        }

        public void Dispose() => Close();
    }

    /// `Environments` of a VectorEnv: an IList<IEnv> of N lane proxies created on demand (VecEnv.cs:25 lets it be replaced).
    internal sealed class LaneList : IList<IEnv> {
        private readonly VectorEnv _owner;
        public LaneList(VectorEnv owner) { _owner = owner; }
        public int Count => _owner.NumberOfEnvironments;
        public bool IsReadOnly => true;
        public IEnv this[int index] {
            get { if (index < 0 || index >= Count) throw new ArgumentOutOfRangeException(nameof(index)); return new LaneEnv(_owner, index); }
            set => throw new NotSupportedException("the lanes of a VectorEnv are fixed");
        }
        public IEnumerator<IEnv> GetEnumerator() { for (int i = 0; i < Count; i++) yield return new LaneEnv(_owner, i); }
        IEnumerator IEnumerable.GetEnumerator() => GetEnumerator();
        public bool Contains(IEnv item) => item is LaneEnv l && ReferenceEquals(l.Owner, _owner);
        public int IndexOf(IEnv item) => item is LaneEnv l && ReferenceEquals(l.Owner, _owner) ? l.Lane : -1;
        public void CopyTo(IEnv[] array, int arrayIndex) { for (int i = 0; i < Count; i++) array[arrayIndex + i] = new LaneEnv(_owner, i); }
        public void Add(IEnv item) => throw new NotSupportedException();
        public void Clear() => throw new NotSupportedException();
        public void Insert(int index, IEnv item) => throw new NotSupportedException();
        public bool Remove(IEnv item) => throw new NotSupportedException();
        public void RemoveAt(int index) => throw new NotSupportedException();
    }

    /// One lane of a VectorEnv seen as an IEnv (IEnv.cs:11-22).  Seed / Reset act on the lane; a lane cannot be stepped alone.
    internal sealed class LaneEnv : IEnv {
        internal readonly VectorEnv Owner;
        internal readonly int Lane;
        public LaneEnv(VectorEnv owner, int lane) { Owner = owner; Lane = lane; }
        public Dict Metadata { get => Owner.Metadata; set => Owner.Metadata = value; }
        public (float From, float To) RewardRange { get => Owner.RewardRange; set => Owner.RewardRange = value; }
        public Space ActionSpace { get => Owner.ActionSpace; set => Owner.ActionSpace = value; }
        public Space ObservationSpace { get => Owner.ObservationSpace; set => Owner.ObservationSpace = value; }
        public NDArray Reset() { var m = new byte[Owner.NumberOfEnvironments]; m[Lane] = 1; return Owner.ResetWhere(m)[Lane]; }   // CartPoleEnv.cs:63-67
        public Step Step(object action) => throw new NotSupportedException("step the whole batch: VectorEnv.Step(int) / Step(NDArray)");
        public Task<Step> StepAsync(object action) => throw new NotSupportedException("step the whole batch: VectorEnv.StepAsync(int)");
        public Image Render(string mode = "human") => null;                       // rendering is out of scope (NullEnvViewer semantics)
        public void CloseEnvironment() { }                                        // the batch owns the resources
        public void Seed(int seed) => Owner.SeedLaneFromProxy(Lane, seed);        // CartPoleEnv.cs:196-198
    }

    /// ONE process driving G GPUs through gymnet_group_* (include/gymnet_amd.h): member m owns lanes [m*N/G, (m+1)*N/G); every
    /// member keeps a replica of all observations, completed by AllGatherObs (hand-written direct push over xGMI, or RCCL).
    public sealed unsafe class GroupVectorEnv : IDisposable {
        private IntPtr _g;
        public int NumberOfEnvironments { get; }
        public int NumMembers { get; }
        public int ObsDim { get; }

        public GroupVectorEnv(GymnetEnvId env, long globalNumEnvs, int[] devices, ulong seed = 0, GymnetFlags flags = GymnetFlags.AutoReset,
                              GymnetGatherMode gather = GymnetGatherMode.Direct) {
            if (devices == null) throw new ArgumentNullException(nameof(devices));
            Native.Check(Native.gymnet_env_describe((int) env, out GymnetEnvInfo info));
            ObsDim = info.obs_dim; NumMembers = devices.Length; NumberOfEnvironments = (int) globalNumEnvs;
            fixed (int* pd = devices) {
                var cfg = new GymnetGroupConfig {
                    struct_size = (uint) sizeof(GymnetGroupConfig), env_id = (int) env, global_num_envs = globalNumEnvs,
                    num_members = devices.Length, flags = (uint) flags, seed = seed, devices = (IntPtr) pd, gather = (int) gather
                };
                Native.Check(Native.gymnet_group_create(ref cfg, out _g));
            }
        }

        public void Seed(int seed) => Native.Check(Native.gymnet_group_seed(_g, (ulong) seed));

        public NDArray Reset() {
            var obs = new float[NumberOfEnvironments * ObsDim];
            fixed (float* p = obs) Native.Check(Native.gymnet_group_reset(_g, p));
            return np.array(obs).reshape(NumberOfEnvironments, ObsDim);
        }

        public (NDArray obs, NDArray reward, NDArray done) Step(int[] actions) {
            int n = NumberOfEnvironments;
            if (actions.Length != n) throw new ArgumentException("Number of actions passed should be equals to number of environments");
            var obs = new float[n * ObsDim]; var rew = new float[n]; var done = new byte[n];
            fixed (int* pa = actions) fixed (float* po = obs) fixed (float* pr = rew) fixed (byte* pd = done)
                Native.Check(Native.gymnet_group_step(_g, pa, po, pr, pd));
            return (np.array(obs).reshape(n, ObsDim), np.array(rew), np.array(done));
        }

        public void StepDevice(IntPtr[] dActions) => Native.Check(Native.gymnet_group_step_device(_g, dActions));
        public void AllGatherObs() => Native.Check(Native.gymnet_group_allgather_obs(_g));
        public void WaitGather() => Native.Check(Native.gymnet_group_wait_gather(_g));
        public IntPtr GlobalObs(int member) { Native.Check(Native.gymnet_group_global_obs(_g, member, out IntPtr p)); return p; }
        public void Sync() => Native.Check(Native.gymnet_group_sync(_g));

        public void Dispose() { if (_g != IntPtr.Zero) { Native.gymnet_group_destroy(_g); _g = IntPtr.Zero; } }
    }
}
