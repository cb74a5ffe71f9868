"""Controller interface stand-ins (farms_core.model.control is not in the reference tree).

The reference's ExperimentTask calls ``controller.step/positions/torques/springrefs`` and reads
``controller.joints_names[ControlType.*]`` every control step (reference task.py:229-252,292-346).
Here the same interface returns batched device tensors instead of per-joint dicts.
"""
import enum
import math

import numpy as np
import torch


class ControlType(enum.IntEnum):
    POSITION = 0
    VELOCITY = 1
    TORQUE = 2


class AnimatController:
    """Base: subclasses return ``[n_envs, n_joints_of_that_type]`` tensors ordered like ``joints_names``."""

    def __init__(self, joints_names=None, muscles_names=None):
        self.joints_names = joints_names or {ControlType.POSITION: [], ControlType.VELOCITY: [], ControlType.TORQUE: []}
        self.muscles_names = muscles_names or []

    def step(self, iteration, time, timestep):
        """Advance the controller state (called once per control step, task.py:292-296)."""

    def positions(self, iteration, time, timestep):
        raise NotImplementedError

    def torques(self, iteration, time, timestep):
        raise NotImplementedError

    def springrefs(self, iteration, time, timestep):
        return None


class WaveController(AnimatController):
    """Travelling-wave position controller of the benchmark configs:
    ``ctrl_j(t, e) = A_j sin(2 pi f t - phi_j + psi_e)``.  ``fusable``: the fused HIP loop evaluates the
    same expression on the device (include/fmj.h: fmj_wave_controller), so no per-step host work remains."""
    fusable = True

    def __init__(self, model, env_phase, frequency=1.0, amplitude=0.3, n_wave=1.0, device='cuda:0'):
        from .model import wave_controller_params
        names = [model.joint_names[model.actuator_jntid[a]] for a in range(model.nu)
                 if model.actuator_tags[a] == 'position']
        super().__init__({ControlType.POSITION: names, ControlType.VELOCITY: [], ControlType.TORQUE: []})
        amp, lag = wave_controller_params(model, amplitude, n_wave)
        self.frequency = float(frequency)
        self.amplitude = torch.as_tensor(amp, dtype=torch.float32, device=device)      # [nu]
        self.phase_lag = torch.as_tensor(lag, dtype=torch.float32, device=device)      # [nu]
        self.env_phase = torch.as_tensor(np.asarray(env_phase), dtype=torch.float32, device=device)  # [n_envs]
        self._pos_idx = torch.as_tensor([a for a in range(model.nu) if model.actuator_tags[a] == 'position'],
                                        device=device)

    def positions(self, iteration, time, timestep):
        cyc = (self.frequency*time) % 1.0
        arg = (2*math.pi*cyc) + self.env_phase[:, None] - self.phase_lag[None, self._pos_idx]
        return self.amplitude[None, self._pos_idx]*torch.sin(arg)
