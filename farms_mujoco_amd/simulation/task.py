"""Task — counterpart of reference farms_mujoco/simulation/task.py (same hooks, batched tensors)."""
from typing import Dict, List

import numpy as np
import torch

from ..control import ControlType
from ..data import AnimatData
from ..units import SimulationUnitScaling as SimulationUnits
from .physics import get_sensor_maps, get_physics2data_maps, physics2data


def duration2nit(duration: float, timestep: float) -> int:
    """Number of iterations from duration"""
    return int(duration/timestep)


class TaskCallback:
    """Task callback (reference task.py:415-446): identical hooks; ``physics`` is the batched physics."""

    writes_state = False     # set True in a callback that edits physics.data.qpos / qvel itself (ExperimentTask.rows_ahead_ok)
    writes_ctrl = False      # set True in a callback that writes physics.data.ctrl itself: the step then takes ctrl from there (ExperimentTask.controller_in_step)

    def __init__(self, substep=False):
        self.substep = substep

    def initialize_episode(self, task, physics): """Initialize episode"""
    def before_step(self, task, action, physics): """Before step"""
    def after_step(self, task, physics): """After step"""
    def action_spec(self, task, physics): """Action specifications"""
    def step_spec(self, task, physics): """Timestep specifications"""
    def get_observation(self, task, physics): """Environment observation"""
    def get_reward(self, task, physics): """Reward"""
    def get_termination(self, task, physics): """Return final discount if episode should end, else None"""
    def observation_spec(self, task, physics): """Observation specifications"""


class SwimmingCallback(TaskCallback):
    """The swimming callback the reference leaves to external packages: constructs the
    SwimmingHandler at episode start and, before every step, computes drag from the links row and writes
    ``physics.data.xfrc_applied`` (world frame, unit-scaled)."""
    fusable = True

    def __init__(self, animat_options, arena_options, substep=False):
        super().__init__(substep=substep)
        self.animat_options = animat_options
        self.arena_options = arena_options
        self.handler = None

    def initialize_episode(self, task, physics):
        from ..swimming.drag import SwimmingHandler
        self.handler = SwimmingHandler(task.data, self.animat_options, self.arena_options, task.units, physics)

    def before_step(self, task, action, physics):
        self.handler.step(task.iteration % task.buffer_size)


class ExperimentTask:
    """FARMS experiment (reference task.py:37-412)."""

    def __init__(self, base_link: str, n_iterations: int, timestep: float, **kwargs):
        self.iteration: int = 0
        self.timestep: float = timestep
        self.n_iterations: int = n_iterations
        self.base_link: str = base_link
        self.data: AnimatData = kwargs.pop('data', None)
        self._controller = kwargs.pop('controller', None)
        self.animat_options = kwargs.pop('animat_options', None)
        self.external_force: float = kwargs.pop('external_force', 0.2)
        self._app = None
        self._restart: bool = kwargs.pop('restart', False)
        self._callbacks: List[TaskCallback] = kwargs.pop('callbacks', [])
        self._extras: Dict = {'hfield': kwargs.pop('hfield', None)}
        self.units: SimulationUnits = kwargs.pop('units', SimulationUnits())
        self.substeps = max(1, kwargs.pop('substeps', 1))
        self.buffer_size = max(1, kwargs.pop('buffer_size', 1))
        self.substeps_links = any(cb.substep for cb in self._callbacks)
        self.sim_iteration = 0
        self.sim_iterations = self.n_iterations*self.substeps
        self.sim_timestep = self.timestep/self.substeps
        self.maps: Dict = {'sensors': {}, 'ctrl': {}, 'xpos': {}, 'qpos': {}, 'geoms': {}, 'links': {},
                           'joints': {}, 'contacts': {}, 'xfrc': {}, 'muscles': {}}
        assert not kwargs, kwargs

    # ---- episode ---------------------------------------------------------------------------------
    def initialize_episode(self, physics):
        """Sets the state of the environment at the start of each episode (reference task.py:87-154)."""
        if self._restart:                      # reference task.py:91-94
            assert self._app is not None, 'Simulation can not be restarted without application interface'
        self.iteration = 0
        self.sim_iteration = 0
        self.initialize_maps(physics)
        if self.data is None:
            self.initialize_data(physics)
        self.data.sensors.links.masses = np.array(
            [physics.model.body_mass[physics.model.body_names.index(n)] for n in self.data.sensors.links.names],
            dtype=float)/self.units.kilograms
        self.initialize_sensors(physics)
        if self._controller is not None:
            self.initialize_control(physics)
        physics.reset(keyframe_id=0)
        for callback in self._callbacks:
            callback.initialize_episode(task=self, physics=physics)

    def initialize_maps(self, physics):
        m = physics.model
        self.maps['xpos']['names'] = m.body_names[1:]
        self.maps['qpos']['names'] = m.hinge_joint_names()
        self.maps['xfrc']['names'] = m.body_names[1:]

    def initialize_data(self, physics):
        self.data = AnimatData.from_sensors_names(timestep=self.timestep, buffer_size=self.buffer_size,
                                                  links=self.maps['xpos']['names'], joints=self.maps['qpos']['names'],
                                                  n_envs=physics.n_envs, device=physics.device)

    def initialize_sensors(self, physics):
        self.maps['sensors'] = get_sensor_maps(physics)
        get_physics2data_maps(physics=physics, sensor_data=self.data.sensors, sensor_maps=self.maps['sensors'])

    def initialize_control(self, physics):
        """ctrl index maps (reference task.py:227-286)."""
        m = physics.model
        names = m.actuator_names
        dev = physics.device
        jn = self._controller.joints_names
        # a device controller that integrates its own state advances once per ITERATION, by task.timestep - not by the model's
        # timestep, which is timestep / num_sub_steps (the reference hands task.timestep to controller.step, task.py:293-300)
        ct = getattr(self._controller, 'timestep', None)
        if ct is not None and abs(ct - self.timestep) > 1e-9*self.timestep:
            if getattr(self._controller, '_timestep_given', True):
                raise ValueError(f'controller.timestep = {ct} but the task steps it once per iteration of {self.timestep} s '
                                 f'({self.substeps} sub-steps of {self.sim_timestep} s)')
            self._controller.timestep = float(self.timestep)
        self.maps['ctrl']['pos'] = torch.as_tensor([names.index(f'actuator_position_{j}') for j in jn[ControlType.POSITION]], device=dev)
        self.maps['ctrl']['vel'] = torch.as_tensor([names.index(f'actuator_velocity_{j}') for j in jn[ControlType.VELOCITY]], device=dev)
        self.maps['ctrl']['trq'] = torch.as_tensor([names.index(f'actuator_torque_{j}') for j in jn[ControlType.TORQUE]], device=dev)
        self.maps['ctrl']['springref'] = {j: int(m.jnt_qposadr[m.joint_names.index(j)]) for j in m.hinge_joint_names()}
        # actuator kind from its bias parameters (reference task.py:258-271, including its quirk that a position
        # actuator with kp = 0 reads as 'trq': SURVEY Appendix C.10)
        jntname2actid = {name: {} for name in m.joint_names}
        for act_i in range(m.nu):
            bias = m.actuator_bias[act_i]
            act_type = 'pos' if bias[1] != 0 else 'vel' if bias[2] != 0 else 'trq'
            jntname2actid[m.joint_names[int(m.actuator_jntid[act_i])]][act_type] = act_i
        self.maps['ctrl']['jntname2actid'] = jntname2actid
        # Actuator limits (reference task.py:273-286): motors that are not position-controlled get their position and
        # velocity actuators switched off by a zero force range, rewritten in the model at run time
        if self.animat_options is not None:
            limited = np.array(m.actuator_forcelimited, np.int32).copy()
            frange = np.array(m.actuator_forcerange, float).reshape(m.nu, 2).copy()
            changed = False
            for mtr_opts in self.animat_options.control.motors:
                if 'position' not in mtr_opts.control_types:
                    for act_type in ('pos', 'vel'):
                        if act_type in jntname2actid.get(mtr_opts.joint_name, {}):
                            a = jntname2actid[mtr_opts.joint_name][act_type]
                            limited[a] = 1
                            frange[a] = (0.0, 0.0)
                            changed = True
            if changed:
                physics.set_actuator_forcerange(limited, frange)

    # ---- per step -----------------------------------------------------------------------------------
    def update_sensors(self, physics, links_only=False, swimming=None):
        index = self.iteration % self.buffer_size
        physics2data(physics=physics, iteration=index, data=self.data, maps=self.maps, units=self.units,
                     links_only=links_only, swimming=swimming)

    def rows_ahead_ok(self, physics):
        """True when the step's launch can also write the NEXT iteration's rows (one launch per iteration with host callbacks): a device
        controller evaluated in the step, no sub-steps, the swimming callback (if any) first among the callbacks, no contact sensors
        and the two-env unconstrained kernel (fmj_step_fused refuses rows_ahead otherwise).  A callback that edits qpos / qvel itself
        sets ``writes_state = True``: the rows of its iteration are then read after it ran, as the reference reads them."""
        if not self.controller_in_step() or physics.has_constraints or physics.kernel_info()['threads_per_env'] != 32:
            return False
        if any(getattr(cb, 'writes_state', False) for cb in self._callbacks) or self.data.sensors.contacts.names:
            return False
        swims = [i for i, cb in enumerate(self._callbacks) if isinstance(cb, SwimmingCallback)]
        return swims in ([], [0])

    def controller_in_step(self):
        """True when the controller is evaluated by the step launch itself (Simulation._env_step -> fmj_step_fused of one step):
        a device controller (``fusable``) in a run without sub-steps whose host callbacks do not write ``physics.data.ctrl``
        (``TaskCallback.writes_ctrl``).  Same semantics as the fused path: actuators the controller does not drive get ctrl 0."""
        c = self._controller
        return (c is not None and getattr(c, 'fusable', False) and self.substeps == 1 and not getattr(self, 'host_step_only', False)
                and not any(getattr(cb, 'writes_ctrl', False) for cb in self._callbacks))

    def before_step(self, action, physics, rows_written=False):
        """Operations before physics step (reference task.py:168-186).  ``rows_written``: the sensors' rows and the swimming callback's
        drag of this iteration are already there - the previous step's launch wrote them (fmj_fused_args::rows_ahead)."""
        # the reference asserts iteration < n_iterations here; with sub-steps its own counter reaches n_iterations one sub-step
        # before the run ends (task.py:352-355) and only dm_control's reset-on-first-step, which costs the run its last
        # environment step, keeps that assert from firing.  run() here advances all n_iterations * substeps
        # steps, so the bound is stated on the counter that does not run ahead - and the sub-steps the reference never executes
        # (task.iteration == n_iterations) write no rows and call no sub-step callback: their ring index would be row 0 of a full log
        assert self.sim_iteration < self.sim_iterations
        full_step = not self.sim_iteration % self.substeps
        in_run = full_step or self.iteration < self.n_iterations
        callbacks = [cb for cb in self._callbacks if full_step or (cb.substep and in_run)]
        sensors = (full_step or self.substeps_links) and in_run
        # a swimming callback that comes first has its drag computed by the sensors' own launch (fmj_before_step)
        swim = callbacks[0] if sensors and callbacks and isinstance(callbacks[0], SwimmingCallback) and callbacks[0].handler is not None else None
        if sensors and not rows_written:
            self.update_sensors(physics=physics, links_only=not full_step, swimming=None if swim is None else swim.handler)
        for callback in callbacks:
            if callback is not swim:
                callback.before_step(task=self, action=action, physics=physics)
        if full_step and self._controller is not None and not self.controller_in_step():
            self.step_control(physics)

    def step_control(self, physics):
        """Step control (reference task.py:288-346): ctrl[pos idx] = positions, ctrl[trq idx] =
        torques*units.torques, qpos_spring[j] = springref."""
        current_time = self.iteration*self.timestep
        index = self.iteration % self.buffer_size
        c = self._controller
        c.step(iteration=index, time=current_time, timestep=self.timestep)
        if c.joints_names[ControlType.POSITION]:
            physics.data.ctrl[:, self.maps['ctrl']['pos']] = c.positions(iteration=index, time=current_time, timestep=self.timestep)
        if c.joints_names[ControlType.TORQUE]:
            physics.data.ctrl[:, self.maps['ctrl']['trq']] = c.torques(iteration=index, time=current_time, timestep=self.timestep)*self.units.torques
            springrefs = c.springrefs(iteration=index, time=current_time, timestep=self.timestep)
            if springrefs:
                for joint, value in springrefs.items():
                    physics.data.qpos_spring[:, self.maps['ctrl']['springref'][joint]] = value

    def after_step(self, physics):
        """Operations after physics step (reference task.py:348-369, including its sub-step counter quirk)."""
        self.sim_iteration += 1
        fullstep = not (self.sim_iteration + 1) % self.substeps
        if fullstep:
            self.iteration += 1
        assert self.iteration <= self.n_iterations
        if self.iteration == self.n_iterations and self._app is not None and not self._restart:
            self._app.close()                  # reference task.py:358-364 (a viewer; None in headless batch runs)
        if fullstep:
            for callback in self._callbacks:
                callback.after_step(task=self, physics=physics)

    def set_app(self, app):
        """Simulation application (reference task.py:82-85); None in headless batch runs."""
        self._app = app

    def get_reward(self, physics):
        reward = 0
        for callback in self._callbacks:
            r = callback.get_reward(task=self, physics=physics)
            if r is not None:
                reward += r
        return reward

    def get_termination(self, physics):
        terminate = None
        for callback in self._callbacks:
            if callback.get_termination(task=self, physics=physics):
                terminate = 1
        if self.iteration >= self.n_iterations:
            terminate = 1
        return terminate

    def action_spec(self, physics):
        """Action specifications (reference task.py:371-378)."""
        specs = []
        for callback in self._callbacks:
            spec = callback.action_spec(task=self, physics=physics)
            if spec is not None:
                specs += spec
        return specs

    def step_spec(self, physics):
        """Timestep specifications (reference task.py:382-385)."""
        for callback in self._callbacks:
            callback.step_spec(task=self, physics=physics)

    def get_observation(self, physics):
        """Environment observation (reference task.py:387-390)."""
        for callback in self._callbacks:
            callback.get_observation(task=self, physics=physics)

    def observation_spec(self, physics):
        """Observation specifications (reference task.py:410-412)."""
        for callback in self._callbacks:
            callback.observation_spec(task=self, physics=physics)

    # ---- fused fast path ------------------------------------------------------------------------------
    def fusable(self):
        """True when every per-step hook has a device implementation, so the whole before_step + mj_step
        sequence can run inside one launch (fmj_step_fused) - sub-steps included: the kernel sequences full steps and
        sub-steps, links-only rows and the iteration counter as before_step / after_step do (include/fmj.h)."""
        cbs_ok = all(getattr(cb, 'fusable', False) for cb in self._callbacks)
        ctl_ok = self._controller is None or getattr(self._controller, 'fusable', False)
        return cbs_ok and ctl_ok and not getattr(self, 'host_step_only', False)
