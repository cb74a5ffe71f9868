"""Simulation — counterpart of reference farms_mujoco/simulation/simulation.py."""
import ctypes
import os
import traceback
from typing import Dict, List

import numpy as np
import torch

from .. import _lib
from ..model import Model
from ..options import AnimatOptions, ArenaOptions, SimulationOptions
from ..physics import BatchedPhysics, PhysicsError
from .task import ExperimentTask, SwimmingCallback

import logging
pylog = logging.getLogger('farms_mujoco_amd')


def extract_sub_dict(dictionary: Dict, keys: List[str]) -> Dict:
    """Extract sub-dictionary"""
    return {key: dictionary.pop(key) for key in keys if key in dictionary}


class Simulation:
    """Batched simulation with the reference's API shape (reference simulation.py:33-213).

    ``mjcf_model`` is a compiled :class:`~farms_mujoco_amd.model.Model` (the dm_control MJCF element tree is
    replaced).  Extra kwargs: ``n_envs``, ``device``.  ``legacy_step`` is accepted for signature
    compatibility; the step is always the full mj_step (legacy_step=False semantics, simulation.py:36-37).
    Unlike dm_control's Environment, whose first ``step()`` only resets, ``run()`` here
    advances exactly ``n_iterations * substeps`` physics steps after an explicit :meth:`reset`.
    """

    def __init__(self, mjcf_model: Model, base_link: str, simulation_options: SimulationOptions,
                 legacy_step: bool = False, **kwargs):
        from .mjcf import check_supported_options
        check_supported_options(simulation_options)
        self._mjcf_model = mjcf_model
        self.options = simulation_options
        self.pause = not self.options.play
        n_envs = kwargs.pop('n_envs', 1)
        device = kwargs.pop('device', 'cuda:0')
        self.physics = BatchedPhysics(mjcf_model, n_envs, device)
        self.handle_exceptions = kwargs.pop('handle_exceptions', False)
        # dm_control raises PhysicsError inside the offending step; here the device freezes the offending env at that
        # step (include/fmj.h) and the host looks at the status words every `check_every` steps (one sync each)
        self.check_every = int(kwargs.pop('check_every', 100))
        self.order_by_contacts = bool(kwargs.pop('order_by_contacts', True))
        extract_sub_dict(kwargs, ('control_timestep', 'n_sub_steps', 'flat_observation'))
        self.task = ExperimentTask(base_link=base_link, n_iterations=self.options.n_iterations,
                                   timestep=self.options.timestep, units=self.options.units,
                                   substeps=self.options.num_sub_steps, **kwargs)
        # RK4 (four forward launches per step, fmj_step): no fused launch, the controller is evaluated on the host path (task.step_control)
        self.task.host_step_only = self.physics.rk4
        self._needs_reset = True

    @property
    def iteration(self):
        """Iteration"""
        return self.task.iteration

    @classmethod
    def from_sdf(cls, simulation_options, animat_options, arena_options, **kwargs):
        """From SDF (reference simulation.py:96-124): compiles ``animat_options.sdf`` with
        :func:`~farms_mujoco_amd.simulation.mjcf.setup_model`, or takes a pre-built ``model=``."""
        model = kwargs.pop('model', None)
        save_mjcf = kwargs.pop('save_mjcf', False)
        show_mjcf = kwargs.pop('show_mjcf', False)
        extract_sub_dict(kwargs, ('spawn_position', 'spawn_rotation', 'use_particles'))
        if model is None:       # setup_mjcf_xml role (reference simulation.py:105-116)
            from .mjcf import setup_model
            model = setup_model(simulation_options, animat_options, arena_options)
        callbacks = kwargs.pop('callbacks', [])
        water = getattr(arena_options, 'water', None)
        if water is not None and water.height is not None and (water.drag or water.sph) \
                and not any(isinstance(cb, SwimmingCallback) for cb in callbacks):
            callbacks = [SwimmingCallback(animat_options, arena_options)] + list(callbacks)
        if save_mjcf or show_mjcf:                   # reference mjcf.py:1503-1509
            from .mjcf import model2mjcf_xml
            xml = model2mjcf_xml(model)
            if show_mjcf:
                pylog.info(xml)
            if save_mjcf:
                with open(save_mjcf if isinstance(save_mjcf, str) else 'simulation_mjcf.xml', 'w+', encoding='utf-8') as f:
                    f.write(xml)
        return cls(mjcf_model=model, base_link=model.body_names[1], simulation_options=simulation_options,
                   animat_options=animat_options, callbacks=callbacks, **kwargs)

    def reset(self):
        self.task.initialize_episode(self.physics)
        self._needs_reset = False
        self._rows_for, self._ahead_ok = -1, None

    # ---- stepping ---------------------------------------------------------------------------------------
    def _env_step(self):
        """Environment.step(action=None): before_step -> physics.step -> after_step.  With host callbacks an iteration is
        ONE launch where the model allows it (swimming, no constraints: the step's launch also writes the next iteration's rows, drag
        and xfrc_applied - fmj_fused_args::rows_ahead - so before_step finds them done) and two otherwise (fmj_before_step, then the
        step, which also evaluates a device controller); round 4 took four and a handful of torch kernels for the ctrl write."""
        task, phys = self.task, self.physics
        if getattr(self, '_ahead_ok', None) is None:
            self._ahead_ok = task.rows_ahead_ok(phys)
        have_rows = getattr(self, '_rows_for', -1) == task.sim_iteration       # the previous launch wrote this iteration's rows
        task.before_step(None, phys, rows_written=have_rows)
        if task.controller_in_step() and task.sim_iteration % task.substeps == 0:
            # rows ahead: only when this iteration's own rows are in place (the first iteration gets them from before_step) and another
            # iteration follows (the row after the run's last would be ring index n_iterations % buffer_size: row 0 of a full log)
            ahead = self._ahead_ok and task.iteration + 1 < task.n_iterations
            self._step_with_controller(rows_ahead=ahead)
            self._rows_for = task.sim_iteration + 1 if ahead else -1
        else:
            phys.step(1)
            self._rows_for = -1
        task.after_step(phys)

    def _step_with_controller(self, rows_ahead=False):
        """One mj_step whose launch evaluates the device controller first (fmj_step_fused of one step).  Without ``rows_ahead`` no rows
        and no drag: before_step wrote them, xfrc_applied and qpos_spring are read from physics.data as fmj_step does.  With it the
        launch also writes what before_step would write for the NEXT iteration."""
        task, phys = self.task, self.physics
        a = getattr(self, '_step_args', None)
        if a is None:
            a = self._step_args = _lib.CFusedArgs()
            a.n_steps, a.buffer_size, a.substeps = 1, task.buffer_size, 1
            a.units = task.units.as_c()
        a.iteration0 = task.iteration
        a.rows_ahead = int(rows_ahead)
        a.do_readout = a.do_drag = 0
        if rows_ahead:
            sens = task.data.sensors
            a.do_readout = 1
            a.rows_base.links, a.rows_base.joints, a.rows_base.xfrc = sens.links.array.data_ptr(), sens.joints.array.data_ptr(), sens.xfrc.array.data_ptr()
            a.row_stride_links, a.row_stride_joints, a.row_stride_xfrc = sens.links.array.stride(0), sens.joints.array.stride(0), sens.xfrc.array.stride(0)
            swim = [cb for cb in task._callbacks if isinstance(cb, SwimmingCallback)]
            a.do_drag = int(bool(swim) and swim[0].handler.drag)
            if swim:
                a.water = swim[0].handler.water.as_c(use_buoyancy=swim[0].handler.buoyancy)
        c = task._controller
        cd = phys._cdata()
        if getattr(c, 'tape', False):
            tape = c.ctrl_tape(1)
            a.controller, a.ctrl_step_stride = 0, tape.stride(0)
            cd.ctrl = tape.data_ptr()
        else:
            a.controller = 1
            a.wave.amplitude, a.wave.phase_lag, a.wave.env_phase = c.amplitude.data_ptr(), c.phase_lag.data_ptr(), c.env_phase.data_ptr()
            a.wave.frequency = c.frequency
            a.ctrl_out = phys.data.ctrl.data_ptr()
        _lib.check(phys._lib.fmj_step_fused(phys._ctx, ctypes.byref(cd), ctypes.byref(a),
                                            ctypes.c_void_p(torch.cuda.current_stream(phys.device).cuda_stream)))

    def step_fused(self, n_steps: int):
        """Run ``n_steps`` full iterations (``substeps`` physics steps each) inside ONE launch (fmj_step_fused): ring-buffer
        readout, drag, xfrc glue, controller and mj_step, with state resident in LDS/registers between steps."""
        task, phys = self.task, self.physics
        assert task.fusable()
        assert task.sim_iteration % task.substeps == 0, 'a fused launch starts on a full step'
        n_steps = min(n_steps, task.n_iterations - task.sim_iteration//task.substeps)
        if n_steps <= 0:
            return 0
        a = _lib.CFusedArgs()
        a.n_steps, a.iteration0, a.buffer_size = n_steps, task.sim_iteration//task.substeps, task.buffer_size
        a.substeps, a.substep_links = task.substeps, int(task.substeps_links)
        a.n_iterations = task.n_iterations if task.n_iterations < (1 << 30) else 0
        a.do_readout = 1
        sens = task.data.sensors
        a.rows_base.links = sens.links.array.data_ptr()
        a.rows_base.joints = sens.joints.array.data_ptr()
        a.rows_base.xfrc = sens.xfrc.array.data_ptr()
        a.row_stride_links = sens.links.array.stride(0)
        a.row_stride_joints = sens.joints.array.stride(0)
        a.row_stride_xfrc = sens.xfrc.array.stride(0)
        if sens.contacts.names:
            a.rows_base.contacts = sens.contacts.array.data_ptr()
            a.row_stride_contacts = sens.contacts.array.stride(0)
        swim = [cb for cb in task._callbacks if isinstance(cb, SwimmingCallback)]
        a.do_drag = int(bool(swim) and swim[0].handler.drag)
        if swim:
            h = swim[0].handler
            a.water = h.water.as_c(use_buoyancy=h.buoyancy)
        a.units = task.units.as_c()
        c = task._controller
        cd = phys._cdata()
        if c is not None and getattr(c, 'tape', False):        # device controller that hands over a ctrl tape
            tape = c.ctrl_tape(n_steps)
            a.controller = 0
            a.ctrl_step_stride = tape.stride(0)
            cd.ctrl = tape.data_ptr()
        elif c is not None:
            a.controller = 1
            a.wave.amplitude, a.wave.phase_lag, a.wave.env_phase = (c.amplitude.data_ptr(), c.phase_lag.data_ptr(),
                                                                     c.env_phase.data_ptr())
            a.wave.frequency = c.frequency
            a.ctrl_out = phys.data.ctrl.data_ptr()       # callbacks reading physics.data.ctrl see the last step's command
        if phys.has_constraints and self.order_by_contacts:
            # envs with the most contacts (the slowest to step) are launched first instead of wherever they sit; the two-env constraint
            # kernel steps entries 2b and 2b + 1 of the order in one wave: neighbours have similar row counts (one PGS sweep length
            # for both; pairing the heaviest with the lightest instead was measured: no difference in the launch time)
            self._env_order = torch.argsort(phys.data.ncon, descending=True, stable=True).to(torch.int32)
            a.env_order = self._env_order.data_ptr()
        _lib.check(phys._lib.fmj_step_fused(phys._ctx, ctypes.byref(cd), ctypes.byref(a),
                                            ctypes.c_void_p(torch.cuda.current_stream(phys.device).cuda_stream)))
        # the counters as n_steps * substeps calls of after_step leave them (task.py:351-355: with sub-steps the iteration is
        # advanced one sub-step early, except that nothing follows the launch's last full step yet)
        task.sim_iteration += n_steps*task.substeps
        task.iteration = (task.sim_iteration + 1)//task.substeps if task.substeps > 1 else task.sim_iteration
        self._rows_for = -1
        return n_steps

    def run(self, fused=None, chunk=None):
        """Run simulation (headless loop of reference simulation.py:148-161)."""
        if self._needs_reset:
            self.reset()
        task = self.task
        fused = task.fusable() if fused is None else fused
        try:
            if fused:
                chunk = chunk or task.buffer_size
                while task.sim_iteration < task.sim_iterations:
                    self.step_fused(min(chunk, task.buffer_size))
                    self.physics.check_invalid_state()      # per chunk: a bad env is already frozen on the device
            else:
                for _ in range(task.sim_iterations - task.sim_iteration):
                    self._env_step()
                    if self.check_every and task.sim_iteration % self.check_every == 0:
                        self.physics.check_invalid_state()
            self.physics.check_invalid_state()
        except PhysicsError as err:
            pylog.error(traceback.format_exc())
            if self.handle_exceptions:
                return
            raise err
        pylog.info('Closing simulation')

    def iterator(self, show_progress: bool = True, verbose: bool = True):
        """Run simulation, yielding each iteration to the caller first (reference simulation.py:164-179)."""
        if self._needs_reset:
            self.reset()
        try:
            for iteration in range(self.task.n_iterations):
                yield iteration
                for _ in range(self.task.substeps):
                    self._env_step()
                if self.check_every and (iteration + 1) % self.check_every == 0:
                    self.physics.check_invalid_state()
            self.physics.check_invalid_state()
        except PhysicsError as err:
            if verbose:
                pylog.error(traceback.format_exc())
            raise err

    # ---- checkpoint / resume ---------------------------------------------------------
    def save_state(self, path: str, include_log: bool = True):
        """Write everything a later ``load_state`` needs to continue this run bit for bit: mjData (``physics.get_state``), the
        task's counters, the device controller's state (``state_dict()`` if it has one) and - ``include_log`` - the ring buffer
        rows written so far.  One ``.npz`` file; call it between steps / launches."""
        task = self.task
        out = {f'physics/{k}': v for k, v in self.physics.get_state().items()}
        out['task/counters'] = np.array([task.iteration, task.sim_iteration, task.n_iterations, task.substeps, task.buffer_size,
                                         self.physics.n_envs], np.int64)
        c = task._controller
        if c is not None and hasattr(c, 'state_dict'):
            out.update({f'controller/{k}': v for k, v in c.state_dict().items()})
        if include_log and task.data is not None:
            sens = task.data.sensors
            for name in ('links', 'joints', 'xfrc', 'contacts'):
                arr = getattr(sens, name).array
                if arr is not None and arr.numel():
                    out[f'log/{name}'] = arr.detach().cpu().numpy()
        with open(path, 'wb') as f:
            np.savez(f, **out)
        return path

    def load_state(self, path: str):
        """Continue from a ``save_state`` file: same model, batch size, sub-steps and ring length (checked)."""
        if self._needs_reset:
            self.reset()
        task = self.task
        with np.load(path) as z:
            it, sim_it, n_it, substeps, buffer_size, n_envs = (int(x) for x in z['task/counters'])
            if (substeps, buffer_size, n_envs) != (task.substeps, task.buffer_size, self.physics.n_envs):
                raise ValueError(f'checkpoint of {n_envs} envs, {substeps} sub-steps, ring of {buffer_size}; this simulation has '
                                 f'{self.physics.n_envs}, {task.substeps}, {task.buffer_size}')
            self.physics.set_state({k[len('physics/'):]: z[k] for k in z.files if k.startswith('physics/')})
            task.iteration, task.sim_iteration = it, sim_it
            self._rows_for = -1
            c = task._controller
            cs = {k[len('controller/'):]: z[k] for k in z.files if k.startswith('controller/')}
            if cs:
                c.load_state_dict(cs)
            sens = task.data.sensors
            for name in ('links', 'joints', 'xfrc', 'contacts'):
                if f'log/{name}' in z.files:
                    arr = getattr(sens, name).array
                    arr.copy_(torch.as_tensor(z[f'log/{name}'], dtype=arr.dtype))

    def postprocess(self, iteration: int, log_path: str = '', plot: bool = False, **kwargs):
        """Postprocessing after simulation (reference simulation.py:181-213)."""
        if log_path:
            pylog.info('Saving data to %s', log_path)
            self.task.data.to_file(os.path.join(log_path, 'simulation.hdf5'), iteration)   # .npz when h5py is absent
            self.options.save(os.path.join(log_path, 'simulation_options.yaml'))
            if self.task.animat_options is not None:
                self.task.animat_options.save(os.path.join(log_path, 'animat_options.yaml'))

    def save_mjcf_xml(self, path: str, verbose: bool = False):
        """Save the compiled model as MJCF XML (reference simulation.py:215-225)."""
        from .mjcf import model2mjcf_xml
        xml = model2mjcf_xml(self.physics.model)
        with open(path, 'w') as f:
            f.write(xml)
        if verbose:
            pylog.info(xml)
        return path
