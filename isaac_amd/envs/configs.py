"""Task / training configuration surface: nested-class configs with the reference's names and values.

Drop-in for the reference's config classes -- same attribute paths, same defaults -- so user scripts
that do `env_cfg.env.num_envs = ...` or subclass `HectorCfg` keep working:
  BaseConfig          humanoid/envs/base/base_config.py:34-56   (recursively instantiates member classes)
  LeggedRobotCfg/PPO  humanoid/envs/base/legged_robot_config.py:34-237
  HectorCfg/PPO       humanoid/envs/custom/hector_config.py:4-234
tests/test_configs.py checks every leaf value against the reference's `class_to_dict` output captured in
tests/golden/configs.json.

No deliberate differences: hector's default `terrain.mesh_type` is the reference's 'trimesh' (procedural tile map,
isaac_amd/envs/terrain.py); 'heightfield' and 'plane' are accepted as in the reference (hector_env.py:120-131).
"""
import inspect


class BaseConfig:
    def __init__(self):
        self.init_member_classes(self)

    @staticmethod
    def init_member_classes(obj):
        for name in dir(obj):
            if name == "__class__":
                continue
            member = getattr(obj, name)
            if inspect.isclass(member):
                inst = member()
                setattr(obj, name, inst)
                BaseConfig.init_member_classes(inst)


class LeggedRobotCfg(BaseConfig):
    class env:
        num_envs = 4096
        num_observations = 235
        num_privileged_obs = None
        num_actions = 12
        env_spacing = 3.
        send_timeouts = True
        episode_length_s = 20

    class terrain:
        mesh_type = 'trimesh'
        horizontal_scale = 0.1
        vertical_scale = 0.005
        border_size = 25
        curriculum = True
        static_friction = 1.0
        dynamic_friction = 1.0
        restitution = 0.
        measure_heights = True
        measured_points_x = [-0.8, -0.7, -0.6, -0.5, -0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
        measured_points_y = [-0.5, -0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4, 0.5]
        selected = False
        terrain_kwargs = None
        max_init_terrain_level = 5
        terrain_length = 8.
        terrain_width = 8.
        num_rows = 10
        num_cols = 20
        terrain_proportions = [0.1, 0.1, 0.35, 0.25, 0.2]
        slope_treshold = 0.75

    class commands:
        curriculum = False
        max_curriculum = 1.
        num_commands = 4
        resampling_time = 10.
        heading_command = True

        class ranges:
            lin_vel_x = [-1.0, 1.0]
            lin_vel_y = [-1.0, 1.0]
            ang_vel_yaw = [-1, 1]
            heading = [-3.14, 3.14]

    class init_state:
        pos = [0.0, 0.0, 1.]
        rot = [0.0, 0.0, 0.0, 1.0]
        lin_vel = [0.0, 0.0, 0.0]
        ang_vel = [0.0, 0.0, 0.0]
        default_joint_angles = {"joint_a": 0., "joint_b": 0.}

    class control:
        stiffness = {'joint_a': 10.0, 'joint_b': 15.}
        damping = {'joint_a': 1.0, 'joint_b': 1.5}
        action_scale = 0.5
        decimation = 4

    class asset:
        file = ""
        name = "legged_robot"
        foot_name = "None"
        penalize_contacts_on = []
        terminate_after_contacts_on = []
        disable_gravity = False
        collapse_fixed_joints = True
        fix_base_link = False
        default_dof_drive_mode = 3
        self_collisions = 0
        replace_cylinder_with_capsule = True
        flip_visual_attachments = True
        density = 0.001
        angular_damping = 0.
        linear_damping = 0.
        max_angular_velocity = 1000.
        max_linear_velocity = 1000.
        armature = 0.
        thickness = 0.01

    class domain_rand:
        randomize_friction = True
        friction_range = [0.5, 1.25]
        randomize_base_mass = False
        added_mass_range = [-1., 1.]
        push_robots = True
        push_interval_s = 15
        max_push_vel_xy = 1.

    class rewards:
        class scales:
            termination = -0.0
            tracking_lin_vel = 1.0
            tracking_ang_vel = 0.5
            lin_vel_z = -2.0
            ang_vel_xy = -0.05
            orientation = -0.
            torques = -0.00001
            dof_vel = -0.
            dof_acc = -2.5e-7
            base_height = -0.
            feet_air_time = 1.0
            collision = -1.
            feet_stumble = -0.0
            action_rate = -0.
            stand_still = -0.

        only_positive_rewards = True
        tracking_sigma = 0.25
        max_contact_force = 100.

    class normalization:
        class obs_scales:
            lin_vel = 2.0
            ang_vel = 0.25
            dof_pos = 1.0
            dof_vel = 0.05
            height_measurements = 5.0
        clip_observations = 100.
        clip_actions = 0.

    class noise:
        add_noise = True
        noise_level = 1.0

        class noise_scales:
            dof_pos = 0.01
            dof_vel = 1.5
            lin_vel = 0.1
            ang_vel = 0.2
            gravity = 0.05
            height_measurements = 0.1

    class viewer:
        ref_env = 0
        pos = [10, 0, 6]
        lookat = [11., 5, 3.]

    class sim:
        dt = 0.005
        substeps = 1
        gravity = [0., 0., -9.81]
        up_axis = 1

        class physx:
            num_threads = 10
            solver_type = 1
            num_position_iterations = 4
            num_velocity_iterations = 0
            contact_offset = 0.01
            rest_offset = 0.0
            bounce_threshold_velocity = 0.5
            max_depenetration_velocity = 1.0
            max_gpu_contact_pairs = 2 ** 23
            default_buffer_size_multiplier = 5
            contact_collection = 2


class LeggedRobotCfgPPO(BaseConfig):
    seed = 1
    runner_class_name = 'OnPolicyRunner'

    class policy:
        init_noise_std = 1.0
        actor_hidden_dims = [512, 256, 128]
        critic_hidden_dims = [512, 256, 128]

    class algorithm:
        value_loss_coef = 1.0
        use_clipped_value_loss = True
        clip_param = 0.2
        entropy_coef = 0.01
        num_learning_epochs = 5
        num_mini_batches = 4
        learning_rate = 1.e-3
        schedule = 'adaptive'
        gamma = 0.99
        lam = 0.95
        desired_kl = 0.01
        max_grad_norm = 1.

    class runner:
        policy_class_name = 'ActorCritic'
        algorithm_class_name = 'PPO'
        num_steps_per_env = 24
        max_iterations = 1500
        save_interval = 100
        experiment_name = 'test'
        run_name = ''
        resume = False
        load_run = -1
        checkpoint = -1
        resume_path = None


class HectorCfg(LeggedRobotCfg):
    class env(LeggedRobotCfg.env):
        frame_stack = 15
        c_frame_stack = 15
        num_single_obs = 41
        num_observations = int(frame_stack * num_single_obs)
        single_num_privileged_obs = 70
        num_privileged_obs = int(c_frame_stack * single_num_privileged_obs)
        num_actions = 10
        num_envs = 4096
        episode_length_s = 24
        use_ref_actions = False

    class safety:
        pos_limit = 0.8
        vel_limit = 0.5
        torque_limit = 0.85

    class asset(LeggedRobotCfg.asset):
        file = '{LEGGED_GYM_ROOT_DIR}/resources/robots/hector_v2/xacro/robot.urdf'
        name = "hector"
        foot_name = "toe"
        knee_name = "calf"
        terminate_after_contacts_on = ['base', 'thigh']
        penalize_contacts_on = ["base", "thigh"]
        self_collisions = 1
        flip_visual_attachments = False
        replace_cylinder_with_capsule = False
        fix_base_link = False

    class terrain(LeggedRobotCfg.terrain):
        mesh_type = 'trimesh'
        curriculum = False
        measure_heights = False
        static_friction = 0.6
        dynamic_friction = 0.6
        terrain_length = 8.
        terrain_width = 8.
        num_rows = 20
        num_cols = 20
        max_init_terrain_level = 10
        terrain_proportions = [0.1, 0.1, 0.2, 0.1, 0.1, 0.2, 0.2]
        restitution = 0.

    class noise:
        add_noise = True
        noise_level = 0.6

        class noise_scales:
            dof_pos = 0.05
            dof_vel = 0.5
            ang_vel = 0.1
            lin_vel = 0.05
            quat = 0.03
            height_measurements = 0.1

    class init_state(LeggedRobotCfg.init_state):
        pos = [0.0, 0.0, 0.55]
        default_joint_angles = {
            'L_hip_joint': 0., 'L_hip_roll_joint': 0.0, 'L_thigh_joint': 0.785, 'L_calf_joint': -1.578, 'L_toe_joint': 0.785,
            'R_hip_joint': 0., 'R_hip_roll_joint': 0., 'R_thigh_joint': 0.785, 'R_calf_joint': -1.578, 'R_toe_joint': 0.785,
        }

    class control(LeggedRobotCfg.control):
        stiffness = {'hip_joint': 40.0, 'hip_roll': 40.0, 'thigh': 60.0, 'calf': 120.0, 'toe': 20.0}
        damping = {'hip_joint': 3.0, 'hip_roll': 3.0, 'thigh': 5.0, 'calf': 4.0, 'toe': 1.0}
        action_scale = 0.25
        decimation = 10

    class sim(LeggedRobotCfg.sim):
        dt = 0.001
        substeps = 1
        up_axis = 1

        class physx(LeggedRobotCfg.sim.physx):
            num_threads = 10
            solver_type = 1
            num_position_iterations = 4
            num_velocity_iterations = 0
            contact_offset = 0.01
            rest_offset = 0.0
            bounce_threshold_velocity = 0.1
            max_depenetration_velocity = 1.0
            max_gpu_contact_pairs = 2 ** 23
            default_buffer_size_multiplier = 5
            contact_collection = 2

    class domain_rand:
        randomize_friction = True
        friction_range = [0.1, 1]
        randomize_base_mass = True
        added_mass_range = [-2., 4.]
        push_robots = True
        push_interval_s = 4
        max_push_vel_xy = 0.3
        max_push_ang_vel = 0.4
        action_delay = 0.0
        action_noise = 0.02

    class commands(LeggedRobotCfg.commands):
        num_commands = 4
        resampling_time = 8.
        heading_command = True

        class ranges:
            lin_vel_x = [-0.6, 0.6]
            lin_vel_y = [-0.3, 0.3]
            ang_vel_yaw = [-0.3, 0.3]
            heading = [-3.14, 3.14]

    class rewards:
        base_height_target = 0.55
        min_dist = 0.1
        max_dist = 0.5
        target_joint_pos_scale = 0.17
        target_feet_height = 0.06
        cycle_time = 0.64
        only_positive_rewards = True
        tracking_sigma = 5
        max_contact_force = 180

        class scales:
            joint_pos = 0.0
            feet_clearance = 1.5
            feet_contact_number = 2.5
            feet_air_time = 2.0
            foot_slip = -0.05
            feet_distance = 0.2
            knee_distance = 0.2
            feet_contact_forces = -0.05
            tracking_lin_vel = 2.5
            tracking_ang_vel = 1.5
            vel_mismatch_exp = 0.0
            low_speed = 0.0
            track_vel_hard = 0.0
            default_joint_pos = 1.7
            orientation = 2
            base_height = 1.0
            base_acc = 0.3
            action_smoothness = -0.008
            torques = -1e-5
            dof_vel = -1e-4
            dof_acc = -1e-6
            collision = -0.5

    class normalization:
        class obs_scales:
            lin_vel = 2.
            ang_vel = 1.
            dof_pos = 1.
            dof_vel = 0.05
            quat = 1.
            height_measurements = 5.0
        clip_observations = 100
        clip_actions = 100


class HectorCfgPPO(LeggedRobotCfgPPO):
    seed = 5
    runner_class_name = 'OnPolicyRunner'

    class policy:
        init_noise_std = 1.0
        actor_hidden_dims = [512, 256, 128]
        critic_hidden_dims = [768, 256, 128]

    class algorithm(LeggedRobotCfgPPO.algorithm):
        entropy_coef = 0.001
        learning_rate = 1e-5
        num_learning_epochs = 2
        gamma = 0.994
        lam = 0.9
        num_mini_batches = 4

    class runner:
        policy_class_name = 'ActorCritic'
        algorithm_class_name = 'PPO'
        num_steps_per_env = 60
        max_iterations = 10001
        save_interval = 100
        experiment_name = 'hector'
        run_name = ''
        resume = False
        load_run = -1
        checkpoint = -1
        resume_path = None


# ------------------------------------------------------------------------------------------------------------------
# hector_full (reference humanoid/envs/custom/hector_w_arm_config.py): the same biped with its two 4-joint arms freed,
# 18 DoF.  Written as what differs from HectorCfg; every value is pinned by tests/golden/configs.json.  The learner
# runs this task's network shapes (tests/test_gpu_ppo_shapes.py), the oracle reproduces its env glue
# (tests/golden/env_rollout_g.npz) and the env-step kernel's 18-DoF instantiation replays that fixture
# (tests/test_gpu_sim.py); registered as task `hector_full`.
class HectorFullCfg(HectorCfg):
    class env(HectorCfg.env):
        num_single_obs = 65
        num_observations = int(HectorCfg.env.frame_stack * num_single_obs)
        single_num_privileged_obs = 94
        num_privileged_obs = int(HectorCfg.env.c_frame_stack * single_num_privileged_obs)
        num_actions = 18

    class asset(HectorCfg.asset):
        file = '{LEGGED_GYM_ROOT_DIR}/resources/robots/hector_v2/xacro/robot_w_arm.urdf'
        terminate_after_contacts_on = ['base', 'thigh', 'shoulder', 'twist', 'roll']

    class terrain(HectorCfg.terrain):
        mesh_type = 'plane'
        terrain_proportions = [0.2, 0.2, 0.4, 0.1, 0.1, 0, 0]

    class init_state(HectorCfg.init_state):
        default_joint_angles = dict(HectorCfg.init_state.default_joint_angles, **{
            'L_shoulder_yaw_joint': 0., 'L_shoulder_pitch_joint': 0., 'L_shoulder_roll_joint': 0., 'L_elbow_joint': -0.785,
            'R_shoulder_yaw_joint': 0., 'R_shoulder_pitch_joint': 0., 'R_shoulder_roll_joint': 0., 'R_elbow_joint': -0.785})

    class control(HectorCfg.control):
        stiffness = {'hip_joint': 80.0, 'hip_roll': 80.0, 'thigh': 80.0, 'calf': 80.0, 'toe': 60.0,
                     'shoulder_yaw': 30.0, 'shoulder_pitch': 30.0, 'shoulder_roll': 30.0, 'elbow': 30.0}
        damping = {'hip_joint': 5.0, 'hip_roll': 5.0, 'thigh': 5.0, 'calf': 5.0, 'toe': 3.0,
                   'shoulder_yaw': 3.0, 'shoulder_pitch': 3.0, 'shoulder_roll': 3.0, 'elbow': 3.0}

    class sim(HectorCfg.sim):
        class physx(HectorCfg.sim.physx):
            num_velocity_iterations = 1

    class domain_rand(HectorCfg.domain_rand):
        friction_range = [0.1, 2.0]
        added_mass_range = [-1., 4.]
        max_push_vel_xy = 0.5

    class commands(HectorCfg.commands):
        class ranges(HectorCfg.commands.ranges):
            lin_vel_x = [-0.6, 0.8]

    class rewards(HectorCfg.rewards):
        min_dist = 0.2
        max_contact_force = 200

        class scales(HectorCfg.rewards.scales):
            feet_clearance = 1.2
            feet_contact_number = 1.5
            feet_air_time = 1.5
            feet_contact_forces = -0.02
            tracking_lin_vel = 1.2
            tracking_ang_vel = 1.1
            vel_mismatch_exp = 0.5
            low_speed = 0.2
            track_vel_hard = 0.5
            default_joint_pos = 1.2
            orientation = 1.
            base_height = 0.8
            base_acc = 0.22
            action_smoothness = -0.002
            dof_vel = -1e-3
            collision = -1.


class HectorFullCfgPPO(HectorCfgPPO):
    class policy(HectorCfgPPO.policy):
        actor_hidden_dims = [768, 512, 128]
        critic_hidden_dims = [768, 768, 768]

    class algorithm(HectorCfgPPO.algorithm):
        entropy_coef = 0.01
        num_learning_epochs = 5
        learning_rate = 1.e-3
        gamma = 0.99
        lam = 0.95

    class runner(HectorCfgPPO.runner):
        experiment_name = 'hector_arm'


# ---- sibling task `humanoid_ppo` (reference humanoid/envs/custom/humanoid_config.py: XBot-L, 12 DoF, 47 x 15 observations,
# 73 x 3 privileged).  Configs only: written as what differs from HectorCfg (which the reference derived from this
# one), every value pinned by tests/golden/configs.json.  The learner runs its network shapes
# (tests/test_gpu_ppo_shapes.py); its env step is not built, so the task is not registered.
class XBotLCfg(HectorCfg):
    class env(HectorCfg.env):
        num_single_obs = 47
        num_observations = int(HectorCfg.env.frame_stack * num_single_obs)
        single_num_privileged_obs = 73
        c_frame_stack = 3
        num_privileged_obs = int(c_frame_stack * single_num_privileged_obs)
        num_actions = 12

    class safety(HectorCfg.safety):
        pos_limit = 1.0
        vel_limit = 1.0

    class asset(HectorCfg.asset):
        file = '{LEGGED_GYM_ROOT_DIR}/resources/robots/XBot/urdf/XBot-L.urdf'
        name = "XBot-L"
        foot_name = "ankle_roll"
        knee_name = "knee"
        terminate_after_contacts_on = ['base_link']
        penalize_contacts_on = ["base_link"]
        self_collisions = 0

    class terrain(HectorCfg.terrain):
        terrain_proportions = [0.2, 0.2, 0.4, 0.1, 0.1, 0, 0]

    class init_state(HectorCfg.init_state):
        pos = [0.0, 0.0, 0.95]
        default_joint_angles = {f'{side}_{j}_joint': 0. for side in ('left', 'right')
                                for j in ('leg_roll', 'leg_yaw', 'leg_pitch', 'knee', 'ankle_pitch', 'ankle_roll')}

    class control(HectorCfg.control):
        stiffness = {'leg_roll': 200.0, 'leg_yaw': 200.0, 'leg_pitch': 350.0, 'knee': 350.0, 'ankle': 15}
        damping = {'leg_roll': 10, 'leg_yaw': 10, 'leg_pitch': 10, 'knee': 10, 'ankle': 10}

    class sim(HectorCfg.sim):
        class physx(HectorCfg.sim.physx):
            num_velocity_iterations = 1

    class domain_rand(HectorCfg.domain_rand):
        friction_range = [0.1, 2.0]
        added_mass_range = [-5., 5.]
        max_push_vel_xy = 0.2
        action_delay = 0.5

    class commands(HectorCfg.commands):
        class ranges(HectorCfg.commands.ranges):
            lin_vel_x = [-0.3, 0.6]

    class rewards(HectorCfg.rewards):
        base_height_target = 0.89
        min_dist = 0.2
        max_contact_force = 700

        class scales(HectorCfg.rewards.scales):
            joint_pos = 1.6
            feet_clearance = 1.
            feet_contact_number = 1.2
            feet_air_time = 1.
            feet_contact_forces = -0.01
            tracking_lin_vel = 1.2
            tracking_ang_vel = 1.1
            vel_mismatch_exp = 0.5
            low_speed = 0.2
            track_vel_hard = 0.5
            default_joint_pos = 0.5
            orientation = 1.
            base_height = 0.2
            base_acc = 0.2
            action_smoothness = -0.002
            dof_vel = -5e-4
            dof_acc = -1e-7
            collision = -1.

    class normalization(HectorCfg.normalization):
        clip_observations = 18.
        clip_actions = 18.


class XBotLCfgPPO(HectorCfgPPO):
    class runner(HectorCfgPPO.runner):
        experiment_name = 'XBot_ppo'
        max_iterations = 3001
