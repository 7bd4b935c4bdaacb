/* abi_manifest.c — prints sizeof / offsetof of every struct field of include/gymnet_amd.h as JSON.
 * Compiled with plain gcc by tests/test_host_api.py: the ctypes binding (gym.net_amd/_capi.py) and the C# binding
 * (gym.net_amd/csharp/Native.cs, [StructLayout(LayoutKind.Sequential)]) are checked against THIS, not against numbers
 * typed into a test. */
#include <stddef.h>
#include <stdio.h>

#include "../include/gymnet_amd.h"

#define BEGIN(T) printf("%s\"%s\": {\"size\": %zu, \"fields\": [", first_struct ? "" : ",\n ", #T, sizeof(T)); first_struct = 0; first = 1
#define F(T, f) printf("%s[\"%s\", %zu, %zu]", first ? "" : ", ", #f, offsetof(T, f), sizeof(((T *)0)->f)); first = 0
#define END() printf("]}")

int main(void) {
    int first = 1, first_struct = 1;
    printf("{");
    BEGIN(gymnet_config);
    F(gymnet_config, struct_size); F(gymnet_config, env_id); F(gymnet_config, num_envs); F(gymnet_config, lane_offset);
    F(gymnet_config, device); F(gymnet_config, flags); F(gymnet_config, seed); F(gymnet_config, stream);
    F(gymnet_config, d_ext_obs); F(gymnet_config, ext_obs_stride); F(gymnet_config, max_episode_steps);
    F(gymnet_config, reserved); F(gymnet_config, d_ext_obs_alt);
    END();
    BEGIN(gymnet_env_info);
    F(gymnet_env_info, struct_size); F(gymnet_env_info, env_id); F(gymnet_env_info, name); F(gymnet_env_info, state_dim);
    F(gymnet_env_info, obs_dim); F(gymnet_env_info, obs_aliases_state); F(gymnet_env_info, action_is_box);
    F(gymnet_env_info, action_n); F(gymnet_env_info, action_low); F(gymnet_env_info, action_high);
    F(gymnet_env_info, obs_low); F(gymnet_env_info, obs_high); F(gymnet_env_info, reward_low); F(gymnet_env_info, reward_high);
    F(gymnet_env_info, algorithmic_bytes_per_step); F(gymnet_env_info, traffic_bytes_per_step); F(gymnet_env_info, state_row_in_obs);
    END();
    BEGIN(gymnet_device_view);
    F(gymnet_device_view, struct_size); F(gymnet_device_view, state_dim); F(gymnet_device_view, obs_dim);
    F(gymnet_device_view, obs_aliases_state); F(gymnet_device_view, num_envs); F(gymnet_device_view, state_stride);
    F(gymnet_device_view, obs_stride); F(gymnet_device_view, d_state); F(gymnet_device_view, d_obs);
    F(gymnet_device_view, d_reward); F(gymnet_device_view, d_done); F(gymnet_device_view, d_steps_beyond_done);
    F(gymnet_device_view, d_final_obs); F(gymnet_device_view, d_done_list); F(gymnet_device_view, d_episode_return);
    F(gymnet_device_view, d_episode_length); F(gymnet_device_view, d_finished_return); F(gymnet_device_view, d_finished_length);
    F(gymnet_device_view, stream); F(gymnet_device_view, obs_buffer); F(gymnet_device_view, state_dtype); F(gymnet_device_view, d_obs_alt);
    END();
    BEGIN(gymnet_counters);
    F(gymnet_counters, struct_size); F(gymnet_counters, reserved); F(gymnet_counters, tick); F(gymnet_counters, lane_steps);
    F(gymnet_counters, stepped_after_done); F(gymnet_counters, last_done_count);
    END();
    BEGIN(gymnet_rollout_buffers);
    F(gymnet_rollout_buffers, d_obs); F(gymnet_rollout_buffers, d_reward); F(gymnet_rollout_buffers, d_done);
    END();
    BEGIN(gymnet_rollout_spec);
    F(gymnet_rollout_spec, struct_size); F(gymnet_rollout_spec, action_source); F(gymnet_rollout_spec, d_actions); F(gymnet_rollout_spec, steps);
    F(gymnet_rollout_spec, action_stride); F(gymnet_rollout_spec, ring); F(gymnet_rollout_spec, action_seed); F(gymnet_rollout_spec, action_tick0);
    F(gymnet_rollout_spec, epsilon); F(gymnet_rollout_spec, record_flags); F(gymnet_rollout_spec, d_rec_obs); F(gymnet_rollout_spec, d_rec_reward);
    F(gymnet_rollout_spec, d_rec_done); F(gymnet_rollout_spec, d_rec_actions); F(gymnet_rollout_spec, d_ep_step); F(gymnet_rollout_spec, d_ep_lane);
    F(gymnet_rollout_spec, d_ep_return); F(gymnet_rollout_spec, d_ep_length); F(gymnet_rollout_spec, ep_capacity); F(gymnet_rollout_spec, d_ep_count);
    END();
    BEGIN(gymnet_group_config);
    F(gymnet_group_config, struct_size); F(gymnet_group_config, env_id); F(gymnet_group_config, global_num_envs);
    F(gymnet_group_config, num_members); F(gymnet_group_config, flags); F(gymnet_group_config, seed);
    F(gymnet_group_config, devices); F(gymnet_group_config, gather); F(gymnet_group_config, max_episode_steps);
    END();
    BEGIN(gymnet_launch_policy);
    F(gymnet_launch_policy, struct_size); F(gymnet_launch_policy, vec); F(gymnet_launch_policy, block); F(gymnet_launch_policy, nt);
    F(gymnet_launch_policy, sequential_lanes); F(gymnet_launch_policy, reset_form); F(gymnet_launch_policy, lds_pipe);
    F(gymnet_launch_policy, occupancy_lds_bytes); F(gymnet_launch_policy, graph);
    END();
    BEGIN(gymnet_ipc_handle);
    F(gymnet_ipc_handle, bytes);
    END();
    printf(",\n \"abi_version\": %d}\n", GYMNET_ABI_VERSION);
    return 0;
}
