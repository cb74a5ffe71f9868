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


class OscillatorNetwork:
    """Host description of a network of amplitude-controlled phase oscillators (include/fmj.h: fmj_cpg_desc).
    ``connections`` = rows (to, from, weight, phase_bias); ``outputs`` = per actuator (osc_a, osc_b, gain, offset),
    ``ctrl_u = gain (r_a (1 + cos theta_a) - r_b (1 + cos theta_b)) + offset``."""

    def __init__(self, frequency, rate, amplitude, connections, outputs, initial_phase=None):
        self.initial_phase = None if initial_phase is None else np.ascontiguousarray(initial_phase, np.float64)
        self.frequency = np.ascontiguousarray(frequency, np.float64)
        self.rate = np.ascontiguousarray(rate, np.float64)
        self.amplitude = np.ascontiguousarray(amplitude, np.float64)
        con = np.asarray(connections, np.float64).reshape(-1, 4)
        self.conn_to = np.ascontiguousarray(con[:, 0], np.int32)
        self.conn_from = np.ascontiguousarray(con[:, 1], np.int32)
        self.conn_weight = np.ascontiguousarray(con[:, 2])
        self.conn_bias = np.ascontiguousarray(con[:, 3])
        out = np.asarray(outputs, np.float64).reshape(-1, 4)
        self.out_a = np.ascontiguousarray(out[:, 0], np.int32)
        self.out_b = np.ascontiguousarray(out[:, 1], np.int32)
        self.out_gain = np.ascontiguousarray(out[:, 2])
        self.out_offset = np.ascontiguousarray(out[:, 3])
        self.n_osc, self.n_conn, self.nu = len(self.frequency), len(self.conn_to), len(self.out_a)

    def as_c(self, struct):
        """Fill a ctypes mirror of fmj_cpg_desc (``struct`` = its class); arrays stay owned by self."""
        import ctypes
        I, D = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double)
        d = struct()
        d.n_osc, d.n_conn, d.nu = self.n_osc, self.n_conn, self.nu
        for k in ('frequency', 'rate', 'amplitude', 'conn_weight', 'conn_bias', 'out_gain', 'out_offset'):
            setattr(d, k, getattr(self, k).ctypes.data_as(D))
        for k in ('conn_to', 'conn_from', 'out_a', 'out_b'):
            setattr(d, k, getattr(self, k).ctypes.data_as(I))
        return d


def salamander_network(model, frequency=1.0, amplitude=0.15, n_wave=1.0, limb_amplitude=0.15, rate=20.0,
                       weight=30.0):
    """Double-chain salamander CPG on the axial joints (left/right oscillator per ``joint_body_*``: ascending and
    descending nearest-neighbour couplings with the travelling-wave phase bias, antiphase contralateral couplings)
    plus one oscillator pair per limb coupled to the axial oscillator at its girdle and to the diagonal limb.
    Axial outputs drive the position actuators with r_L (1 + cos th_L) - r_R (1 + cos th_R); each limb drives the
    position actuator of its ``_0`` joint.  Amplitude 0.15 gives a 0.3 rad joint swing."""
    axial = [j for j, n in enumerate(model.joint_names) if n.startswith('joint_body_')]
    na = len(axial)
    limbs = sorted({n.rsplit('_', 1)[0] for n in model.joint_names if n.startswith('joint_leg_')})
    n_osc = 2*na + 2*len(limbs)
    freq = np.full(n_osc, float(frequency)); rt = np.full(n_osc, float(rate))
    amp = np.concatenate([np.full(2*na, float(amplitude)), np.full(2*len(limbs), float(limb_amplitude))])
    con = []
    bias = 2*np.pi*n_wave/na
    for k in range(na):
        for side in (0, 1):
            i = 2*k + side
            con.append((i, 2*k + (1 - side), weight, np.pi))                    # contralateral, antiphase
            if k + 1 < na:
                con.append((2*(k + 1) + side, i, weight, bias))                 # descending: the next segment lags
                con.append((i, 2*(k + 1) + side, weight, -bias))                # ascending
    for li, name in enumerate(limbs):
        a, b = 2*na + 2*li, 2*na + 2*li + 1
        con.append((a, b, weight, np.pi)); con.append((b, a, weight, np.pi))
        girdle = 0 if 'front' in name else min(na - 1, na//2)
        side = 0 if name.endswith('_L') else 1
        con.append((a, 2*girdle + side, weight, np.pi))                         # limb swings against its body side
    out = []
    for u in range(model.nu):
        jn = model.joint_names[model.actuator_jntid[u]]
        if model.actuator_tags[u] != 'position':
            out.append((-1, -1, 0.0, 0.0))
        elif jn.startswith('joint_body_'):
            k = axial.index(int(model.actuator_jntid[u]))
            out.append((2*k, 2*k + 1, 1.0, 0.0))
        elif jn.endswith('_0') and jn.rsplit('_', 1)[0] in limbs:
            li = limbs.index(jn.rsplit('_', 1)[0])
            out.append((2*na + 2*li, 2*na + 2*li + 1, 1.0, 0.0))
        else:
            out.append((-1, -1, 0.0, 0.0))
    ph0 = np.zeros(n_osc)                     # the limit cycle: head-to-tail lag, sides and limb pairs in antiphase
    for k in range(na):
        ph0[2*k] = -k*bias; ph0[2*k + 1] = -k*bias + np.pi
    for li, name in enumerate(limbs):
        girdle = 0 if 'front' in name else min(na - 1, na//2)
        side = 0 if name.endswith('_L') else 1
        ph0[2*na + 2*li] = ph0[2*girdle + side] + np.pi; ph0[2*na + 2*li + 1] = ph0[2*girdle + side]
    return OscillatorNetwork(freq, rt, amp, con, out, initial_phase=(ph0 + np.pi) % (2*np.pi) - np.pi)


class NetworkController(AnimatController):
    """AnimatController (reference task.py:292-346) whose joint position commands come from an OscillatorNetwork
    integrated on the device by ``fmj_cpg_tape``.  ``fusable``: a chunk of the fused loop asks for its
    ctrl tape up front (``ctrl_tape``), so no host work or host->device copy remains per step.  ``drive`` ([n_envs],
    optional) scales the intrinsic frequencies per env."""
    fusable = True
    tape = True

    def __init__(self, model, network, n_envs, env_phase=None, drive=None, device='cuda:0', timestep=None):
        from . import _lib
        names = [model.joint_names[model.actuator_jntid[a]] for a in range(model.nu)
                 if model.actuator_tags[a] == 'position']
        super().__init__({ControlType.POSITION: names, ControlType.VELOCITY: [], ControlType.TORQUE: []})
        assert network.nu == model.nu
        self.network, self.n_envs, self.nu, self.device = network, n_envs, model.nu, torch.device(device)
        # the network advances once per ITERATION (task.py:288-346 calls the controller on full steps only): with sub-steps pass
        # simulation_options.timestep, the model's own timestep being timestep / num_sub_steps (mjcf.py:1187-1192)
        self.timestep = float(model.timestep if timestep is None else timestep)
        self._timestep_given = timestep is not None      # ExperimentTask.initialize_control sets / validates it against task.timestep
        self._lib = _lib.load()
        import ctypes
        self._ctx = ctypes.c_void_p()
        self._desc = network.as_c(_lib.CCpgDesc)
        _lib.check(self._lib.fmj_cpg_create(ctypes.byref(self._desc), self.device.index or 0, ctypes.byref(self._ctx)))
        ph = np.zeros((n_envs, network.n_osc))
        if network.initial_phase is not None:
            ph += network.initial_phase[None, :]
        if env_phase is not None:
            ph += np.asarray(env_phase, np.float64).reshape(n_envs, 1)
        ph = (ph + np.pi) % (2*np.pi) - np.pi
        self.phase = torch.as_tensor(ph, dtype=torch.float32, device=self.device).contiguous()
        self.amp = torch.zeros(n_envs, network.n_osc, dtype=torch.float32, device=self.device)
        self.damp = torch.zeros_like(self.amp)
        self.drive = None if drive is None else torch.as_tensor(np.asarray(drive), dtype=torch.float32,
                                                                device=self.device).contiguous()
        self._pos_idx = torch.as_tensor([a for a in range(model.nu) if model.actuator_tags[a] == 'position'],
                                        device=self.device)
        self._tape = None

    def __del__(self):
        try:
            if getattr(self, '_ctx', None):
                self._lib.fmj_cpg_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    def state_dict(self):
        """Oscillator state for Simulation.save_state (phases, amplitudes and their rates, per env)."""
        return {k: getattr(self, k).detach().cpu().numpy().copy() for k in ('phase', 'amp', 'damp')}

    def load_state_dict(self, state):
        for k in ('phase', 'amp', 'damp'):
            t = getattr(self, k)
            t.copy_(torch.as_tensor(np.asarray(state[k]), dtype=t.dtype))

    def ctrl_tape(self, n_steps):
        """Advance the network ``n_steps`` and return ctrl[n_steps, n_envs, nu] (device)."""
        import ctypes
        from . import _lib
        if self._tape is None or self._tape.shape[0] < n_steps:
            self._tape = torch.empty(n_steps, self.n_envs, self.nu, dtype=torch.float32, device=self.device)
        tape = self._tape[:n_steps]
        _lib.check(self._lib.fmj_cpg_tape(self._ctx, self.n_envs, n_steps, self.timestep, self.phase.data_ptr(),
                                          self.amp.data_ptr(), self.damp.data_ptr(),
                                          None if self.drive is None else self.drive.data_ptr(), tape.data_ptr(),
                                          ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return tape

    def positions(self, iteration, time, timestep):
        return self.ctrl_tape(1)[0][:, self._pos_idx]
