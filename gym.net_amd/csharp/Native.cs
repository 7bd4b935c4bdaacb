// Native.cs — P/Invoke declarations for include/gymnet_amd.h (libgymnet_amd.so), ABI version 4.
// UNVERIFIED: no .NET toolchain exists in the build image, so this file has never been compiled.
// It is the binding a Gym.NET maintainer would add next to src/Gym/Envs/VecEnv.cs.  What CAN be checked here is checked by
// tests/test_host_api.py: one [DllImport] per header entry point (no exceptions), and every [StructLayout(Sequential)]
// struct below against the sizeof / offsetof manifest a C program prints from the header itself (tools/abi_manifest.c):
// same field names, same order, same sizes, same offsets.
using System;
using System.Runtime.InteropServices;

namespace Gym.Envs.Amd {
    public enum GymnetStatus {
        Ok = 0, InvalidArg = -1, InvalidAction = -2, Hip = -3, Oom = -4, NoDevice = -5,
        AlreadyStepping = -6, NotStepping = -7, Unsupported = -8, Rccl = -9
    }

    public enum GymnetEnvId { CartPole = 0, Pendulum = 1, MountainCar = 2, Acrobot = 3 }

    public enum GymnetGatherMode { None = 0, Direct = 1, Rccl = 2 }

    [Flags]
    public enum GymnetFlags : uint {
        None = 0, AutoReset = 0x01, ValidateActions = 0x02, DoneList = 0x04, EpisodeStats = 0x08, FinalObs = 0x10,
        DoubleBuffer = 0x20, F64 = 0x40, CompactRecordsOnly = 0x80,
        Resident = 0x100      // ABI 5, num_envs <= 64: host-boundary Step / Reset served by a resident single-wave kernel + a host-memory mailbox
    }

    public enum GymnetDtype { F32 = 0, F64 = 1 }

    public enum GymnetArrayId {
        Reward = 0, Done = 1, StepsBeyondDone = 2, EpisodeReturn = 3, EpisodeLength = 4, FinishedReturn = 5, FinishedLength = 6,
        FinalObs = 7, LaneSeeds = 8
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct GymnetConfig {
        public uint struct_size; public int env_id; public long num_envs; public long lane_offset;
        public int device; public uint flags; public ulong seed; public IntPtr stream;
        public IntPtr d_ext_obs; public long ext_obs_stride; public int max_episode_steps; public int reserved;
        public IntPtr d_ext_obs_alt;
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct GymnetEnvInfo {
        public uint struct_size; public int env_id; public fixed byte name[32];
        public int state_dim; public int obs_dim; public int obs_aliases_state; public int action_is_box; public int action_n;
        public float action_low; public float action_high;
        public fixed float obs_low[8]; public fixed float obs_high[8];
        public float reward_low; public float reward_high; public int algorithmic_bytes_per_step;
        public int traffic_bytes_per_step; public fixed int state_row_in_obs[8];
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct GymnetDeviceView {
        public uint struct_size; public int state_dim; public int obs_dim; public int obs_aliases_state;
        public long num_envs; public long state_stride; public long obs_stride;
        public IntPtr d_state; public IntPtr d_obs; public IntPtr d_reward; public IntPtr d_done;
        public IntPtr d_steps_beyond_done; public IntPtr d_final_obs; public IntPtr d_done_list;
        public IntPtr d_episode_return; public IntPtr d_episode_length; public IntPtr d_finished_return; public IntPtr d_finished_length;
        public IntPtr stream; public int obs_buffer; public int state_dtype; public IntPtr d_obs_alt;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct GymnetCounters {
        public uint struct_size; public uint reserved; public ulong tick; public ulong lane_steps; public ulong stepped_after_done;
        public long last_done_count;
    }

    /// Launch configuration of the step kernel; every field -1 = leave as it is (gymnet_vecenv_set_launch_policy).
    [StructLayout(LayoutKind.Sequential)]
    public struct GymnetLaunchPolicy {
        public uint struct_size; public int vec; public int block; public int nt; public int sequential_lanes; public int reset_form;
        public int lds_pipe; public int occupancy_lds_bytes; public int graph;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct GymnetRolloutBuffers { public IntPtr d_obs; public IntPtr d_reward; public IntPtr d_done; }

    /// gymnet_rollout_spec.record_flags: NoOverflow = the 8 % faster records variant that may drop records of very unevenly finishing lanes
    public static class GymnetRecordFlags { public const int None = 0, NoOverflow = 1; }
    /// gymnet_vecenv_rollout_fused_ex_device (ABI 5): action source (0 ring, 1 ActionSpace.Sample() drawn in the kernel, 2 epsilon-greedy
    /// over the ring as the policy's actions), dense recording, and the compact records of the episodes that end during the rollout.
    [StructLayout(LayoutKind.Sequential)]
    public struct GymnetRolloutSpec {
        public uint struct_size; public int action_source; public IntPtr d_actions; public long steps; public long action_stride; public long ring;
        public ulong action_seed; public ulong action_tick0; public float epsilon; public int record_flags;
        public IntPtr d_rec_obs; public IntPtr d_rec_reward; public IntPtr d_rec_done; public IntPtr d_rec_actions;
        public IntPtr d_ep_step; public IntPtr d_ep_lane; public IntPtr d_ep_return; public IntPtr d_ep_length; public long ep_capacity; public IntPtr d_ep_count;
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct GymnetIpcHandle { public fixed byte bytes[64]; }

    [StructLayout(LayoutKind.Sequential)]
    public struct GymnetGroupConfig {
        public uint struct_size; public int env_id; public long global_num_envs; public int num_members; public uint flags;
        public ulong seed; public IntPtr devices; public int gather; public int max_episode_steps;
    }

    internal static unsafe class Native {
        private const string Lib = "gymnet_amd";   // libgymnet_amd.so on the loader path

        // ---- library level
        [DllImport(Lib)] public static extern int gymnet_abi_version();
        [DllImport(Lib)] public static extern IntPtr gymnet_status_string(int status);
        [DllImport(Lib)] public static extern IntPtr gymnet_last_error();
        [DllImport(Lib)] public static extern int gymnet_device_count(out int count);
        [DllImport(Lib)] public static extern int gymnet_env_describe(int env_id, out GymnetEnvInfo info);

        // ---- lifecycle
        [DllImport(Lib)] public static extern int gymnet_vecenv_create(ref GymnetConfig cfg, out IntPtr handle);
        [DllImport(Lib)] public static extern int gymnet_vecenv_destroy(IntPtr h);
        [DllImport(Lib)] public static extern int gymnet_vecenv_seed(IntPtr h, ulong seed);
        [DllImport(Lib)] public static extern int gymnet_vecenv_seed_lanes(IntPtr h, ulong[] seeds, long count);

        // ---- host-boundary path (obs_out: float32 [N, D]; float64 for a handle created with GymnetFlags.F64 — hence void*)
        [DllImport(Lib)] public static extern int gymnet_vecenv_reset(IntPtr h, void* obs_out);
        [DllImport(Lib)] public static extern int gymnet_vecenv_reset_where(IntPtr h, byte* mask, void* obs_out);
        [DllImport(Lib)] public static extern int gymnet_vecenv_step(IntPtr h, void* actions, void* obs_out, float* reward_out, byte* done_out);
        [DllImport(Lib)] public static extern int gymnet_vecenv_step_broadcast(IntPtr h, int action, void* obs_out, float* reward_out, byte* done_out);
        [DllImport(Lib)] public static extern int gymnet_vecenv_step_async(IntPtr h, void* actions);
        [DllImport(Lib)] public static extern int gymnet_vecenv_step_wait(IntPtr h, void* obs_out, float* reward_out, byte* done_out);
        [DllImport(Lib)] public static extern int gymnet_vecenv_read(IntPtr h, void* obs_out, float* reward_out, byte* done_out);

        // ---- device-resident path
        [DllImport(Lib)] public static extern int gymnet_vecenv_reset_device(IntPtr h);
        [DllImport(Lib)] public static extern int gymnet_vecenv_reset_where_device(IntPtr h, IntPtr d_mask);
        [DllImport(Lib)] public static extern int gymnet_vecenv_step_device(IntPtr h, IntPtr d_actions);
        [DllImport(Lib)] public static extern int gymnet_vecenv_rollout_device(IntPtr h, IntPtr d_actions, long steps, long action_stride, long ring);
        [DllImport(Lib)] public static extern int gymnet_vecenv_rollout_fused_device(IntPtr h, IntPtr d_actions, long steps, long action_stride, long ring, ref GymnetRolloutBuffers rec);
        [DllImport(Lib)] public static extern int gymnet_vecenv_rollout_fused_device(IntPtr h, IntPtr d_actions, long steps, long action_stride, long ring, IntPtr rec_null);   // rec = NULL: record nothing
        [DllImport(Lib)] public static extern int gymnet_vecenv_rollout_fused_ex_device(IntPtr h, ref GymnetRolloutSpec spec);
        [DllImport(Lib)] public static extern int gymnet_vecenv_pack_obs_device(IntPtr h, IntPtr d_obs_rowmajor);
        [DllImport(Lib)] public static extern int gymnet_vecenv_sync(IntPtr h);
        [DllImport(Lib)] public static extern int gymnet_vecenv_device_view(IntPtr h, out GymnetDeviceView view);
        [DllImport(Lib)] public static extern int gymnet_vecenv_launch_policy(IntPtr h, out int vec, out int block, out int nt, out int sequential_lanes);
        [DllImport(Lib)] public static extern int gymnet_vecenv_set_launch_policy(IntPtr h, ref GymnetLaunchPolicy policy);
        [DllImport(Lib)] public static extern int gymnet_vecenv_get_launch_policy(IntPtr h, out GymnetLaunchPolicy policy);
        [DllImport(Lib)] public static extern int gymnet_vecenv_kernel_name(IntPtr h, byte[] buf, int capacity);
        [DllImport(Lib)] public static extern int gymnet_vecenv_host_buffers(IntPtr h, out IntPtr actions, out IntPtr obs, out IntPtr reward, out IntPtr done);

        // ---- state access / bookkeeping
        [DllImport(Lib)] public static extern int gymnet_vecenv_get_state(IntPtr h, void* state_soa);
        [DllImport(Lib)] public static extern int gymnet_vecenv_set_state(IntPtr h, void* state_soa);
        [DllImport(Lib)] public static extern int gymnet_vecenv_get_steps_beyond_done(IntPtr h, int* out_sbd);
        [DllImport(Lib)] public static extern int gymnet_vecenv_set_steps_beyond_done(IntPtr h, int* in_sbd);
        [DllImport(Lib)] public static extern int gymnet_vecenv_get_tick(IntPtr h, out ulong tick);
        [DllImport(Lib)] public static extern int gymnet_vecenv_set_tick(IntPtr h, ulong tick);
        [DllImport(Lib)] public static extern int gymnet_vecenv_counters(IntPtr h, out GymnetCounters counters);
        [DllImport(Lib)] public static extern int gymnet_vecenv_get_array(IntPtr h, int which, void* out_array, long bytes);
        [DllImport(Lib)] public static extern int gymnet_vecenv_set_array(IntPtr h, int which, void* in_array, long bytes);
        [DllImport(Lib)] public static extern int gymnet_vecenv_get_seed(IntPtr h, out ulong seed, out int per_lane);
        [DllImport(Lib)] public static extern int gymnet_vecenv_done_lanes(IntPtr h, int* lanes_out, long capacity, out long count);
        [DllImport(Lib)] public static extern int gymnet_vecenv_done_lanes_device(IntPtr h, IntPtr d_lanes_out, IntPtr d_count_out);
        [DllImport(Lib)] public static extern int gymnet_vecenv_episode_stats(IntPtr h, float* finished_return, int* finished_length);
        [DllImport(Lib)] public static extern int gymnet_vecenv_final_obs(IntPtr h, float* final_obs_out);
        [DllImport(Lib)] public static extern int gymnet_vecenv_done_records(IntPtr h, int* lanes_out, float* return_out, int* length_out, float* final_obs_out, long capacity, out long count);
        [DllImport(Lib)] public static extern int gymnet_vecenv_done_records_device(IntPtr h, IntPtr d_lanes, IntPtr d_return, IntPtr d_length, IntPtr d_final_obs, long capacity, IntPtr d_count);

        // ---- batched space sampling
        [DllImport(Lib)] public static extern int gymnet_sample_discrete_device(int device, IntPtr stream, IntPtr d_out, long count, int n, int start, ulong seed, ulong lane_offset, ulong tick);
        [DllImport(Lib)] public static extern int gymnet_sample_discrete_masked_device(int device, IntPtr stream, IntPtr d_out, long count, int n, int start, IntPtr d_mask, long mask_stride, ulong seed, ulong lane_offset, ulong tick);
        [DllImport(Lib)] public static extern int gymnet_sample_box_device(int device, IntPtr stream, IntPtr d_out, long count, float low, float high, ulong seed, ulong lane_offset, ulong tick);
        [DllImport(Lib)] public static extern int gymnet_sample_box_elementwise_device(int device, IntPtr stream, IntPtr d_out, long count, int dim, IntPtr d_low, IntPtr d_high, ulong seed, ulong lane_offset, ulong tick);
        [DllImport(Lib)] public static extern int gymnet_vecenv_sample_actions_device(IntPtr h, IntPtr d_actions, ulong seed, ulong tick);
        [DllImport(Lib)] public static extern int gymnet_vecenv_sample_actions_masked_device(IntPtr h, IntPtr d_actions, IntPtr d_mask, long mask_stride, ulong seed, ulong tick);
        [DllImport(Lib)] public static extern int gymnet_vecenv_sample_actions(IntPtr h, void* actions_out, ulong seed, ulong tick);
        [DllImport(Lib)] public static extern int gymnet_vecenv_compose_actions_device(IntPtr h, IntPtr d_policy_actions, float epsilon, IntPtr d_actions_out, ulong seed, ulong tick);

        // ---- multi-GPU group (one process, G members)
        [DllImport(Lib)] public static extern int gymnet_group_create(ref GymnetGroupConfig cfg, out IntPtr group);
        [DllImport(Lib)] public static extern int gymnet_group_destroy(IntPtr g);
        [DllImport(Lib)] public static extern int gymnet_group_size(IntPtr g, out int num_members, out long lanes_per_member);
        [DllImport(Lib)] public static extern int gymnet_group_member(IntPtr g, int member, out IntPtr handle);
        [DllImport(Lib)] public static extern int gymnet_group_seed(IntPtr g, ulong seed);
        [DllImport(Lib)] public static extern int gymnet_group_reset_device(IntPtr g);
        [DllImport(Lib)] public static extern int gymnet_group_step_device(IntPtr g, IntPtr[] d_actions);
        [DllImport(Lib)] public static extern int gymnet_group_rollout_device(IntPtr g, IntPtr[] d_actions, long steps, long action_stride, long ring);
        [DllImport(Lib)] public static extern int gymnet_group_allgather_obs(IntPtr g);
        [DllImport(Lib)] public static extern int gymnet_group_wait_gather(IntPtr g);
        [DllImport(Lib)] public static extern int gymnet_group_global_obs(IntPtr g, int member, out IntPtr d_obs_all);
        [DllImport(Lib)] public static extern int gymnet_group_read_replica(IntPtr g, int member, float* replica_out);
        [DllImport(Lib)] public static extern int gymnet_group_sync(IntPtr g);
        [DllImport(Lib)] public static extern int gymnet_group_reset(IntPtr g, float* obs_out);
        [DllImport(Lib)] public static extern int gymnet_group_step(IntPtr g, void* actions, float* obs_out, float* reward_out, byte* done_out);

        // ---- peer buffers (one process per GPU hosts: direct all-gather over HIP IPC)
        [DllImport(Lib)] public static extern int gymnet_peer_buffer_create(int device, long bytes, out IntPtr d_ptr, out GymnetIpcHandle handle);
        [DllImport(Lib)] public static extern int gymnet_peer_buffer_open(int device, ref GymnetIpcHandle handle, out IntPtr d_ptr);
        [DllImport(Lib)] public static extern int gymnet_peer_buffer_close(int device, IntPtr d_ptr);
        [DllImport(Lib)] public static extern int gymnet_peer_buffer_destroy(int device, IntPtr d_ptr);
        [DllImport(Lib)] public static extern int gymnet_push_obs_device(int device, IntPtr stream, IntPtr d_src, IntPtr[] d_dst, int npeers, long count);

        /// Maps a status to the exception the reference throws for the same condition.
        public static void Check(int status) {
            if (status == 0) return;
            string msg = Marshal.PtrToStringAnsi(gymnet_last_error());
            if (string.IsNullOrEmpty(msg)) msg = Marshal.PtrToStringAnsi(gymnet_status_string(status));
            switch ((GymnetStatus) status) {
                case GymnetStatus.InvalidArg: throw new ArgumentException(msg);                       // VecEnv.cs:49
                case GymnetStatus.InvalidAction: throw new Gym.Exceptions.InvalidActionError(msg);    // InvalidActionError.cs:7-10
                case GymnetStatus.AlreadyStepping: throw new Gym.Exceptions.AlreadySteppingError();   // AlreadySteppingError.cs:8-10
                case GymnetStatus.NotStepping: throw new Gym.Exceptions.NotSteppingError();
                case GymnetStatus.Oom: throw new OutOfMemoryException(msg);
                case GymnetStatus.Unsupported: throw new NotSupportedException(msg);
                default: throw new InvalidOperationException($"gymnet_amd: {msg} (status {status})");
            }
        }
    }
}
