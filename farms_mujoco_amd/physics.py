"""Batched physics: the object behind ``Simulation.physics``.

Replaces ``dm_control.mjcf.Physics`` (reference farms_mujoco/simulation/simulation.py:53) for a
batch of independent environments.  ``physics.data.<field>`` are PyTorch tensors on the GPU, fp32,
batch-first (``[n_envs, ...]``); ``physics.model`` is the shared :class:`~farms_mujoco_amd.model.Model`.
All arithmetic happens in the HIP library (``_lib``); this file only owns memory and marshals pointers.
"""
from __future__ import annotations

import ctypes
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from . import _lib
from .model import Model, JNT_FREE


class PhysicsError(RuntimeError):
    """Bad simulation state (dm_control.rl.control.PhysicsError role, reference simulation.py:13,157-161)."""


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _MjData(SimpleNamespace):
    """``physics.data``: rebinding a field (``data.qpos = tensor``) is seen by the cached pointer struct of the per-step host path;
    in-place writes (``data.qpos[:] = ...``) never move a tensor."""

    def __setattr__(self, name, value):
        object.__setattr__(self, '_cdata_cache', None)
        object.__setattr__(self, name, value)


class BatchedPhysics:
    """Device-resident mjData for ``n_envs`` copies of one model + the HIP step context."""

    def __init__(self, model: Model, n_envs: int, device='cuda:0'):
        if not torch.cuda.is_available():
            raise _lib.FmjError('BatchedPhysics needs a GPU (torch.cuda.is_available() is False); '
                                'there is no CPU fallback for the product path.')
        self.model = model
        self.n_envs = int(n_envs)
        self.device = torch.device(device)
        self._lib = _lib.load()
        self._cmodel = model.as_c()
        ctx = ctypes.c_void_p()
        _lib.check(self._lib.fmj_create(ctypes.byref(self._cmodel), self.n_envs, self.device.index or 0,
                                        ctypes.byref(ctx)))
        self._ctx = ctx
        lay = _lib.CSensorLayout()
        _lib.check(self._lib.fmj_get_sensor_layout(self._ctx, ctypes.byref(lay)))
        self.sensor_layout = lay
        assert lay.nsensordata == model.nsensordata
        n, m, dev = self.n_envs, model, self.device
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=dev)
        self.data = _MjData(
            qpos=z(n, m.nq), qvel=z(n, m.nv), ctrl=z(n, m.nu), qpos_spring=z(n, m.nq),
            xfrc_applied=z(n, m.nbody, 6), xpos=z(n, m.nbody, 3), xquat=z(n, m.nbody, 4), xipos=z(n, m.nbody, 3),
            sensordata=z(n, m.nsensordata), qacc=z(n, m.nv), time=z(n), status=z(n, dtype=torch.int32))
        self.has_constraints = bool(np.any(m.jnt_limited)) or m.ngeom > 0
        self.rk4 = int(getattr(m, 'integrator', 0)) == 1      # FMJ_INT_RK4: fmj_step runs four forward launches per step, fmj_step_fused refuses
        self.max_contacts = max(int(m.max_contacts), 1)
        self.data.qacc_warmstart = z(n, m.nv)
        self.data.contact = z(n, self.max_contacts, 16)
        self.data.ncon = z(n, dtype=torch.int32)
        self.data.xquat[:, :, 0] = 1.0
        self.data.qpos_spring[:] = torch.as_tensor(m.qpos_spring, dtype=torch.float32)
        # the solver that actually runs (fmj_solver_info): fmj_create swaps a primal solver for the dual one on near-frictionless contacts
        rq, ef, it = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        if hasattr(self._lib, 'fmj_solver_info') and self._lib.fmj_solver_info.argtypes:     # absent only from an A/B base build (FMJ_SO)
            _lib.check(self._lib.fmj_solver_info(self._ctx, ctypes.byref(rq), ctypes.byref(ef), ctypes.byref(it)))
        names = {0: 'PGS', 1: 'CG', 2: 'Newton'}
        self.solver_requested, self.solver_effective, self.solver_budget = names[rq.value], names[ef.value], it.value
        if rq.value != ef.value:
            import warnings
            why = ('the noslip post-pass works on the dual matrices, which a primal solver never forms' if int(getattr(model, 'noslip_iterations', 0)) > 0 else
                   'explicit pairs under the elliptic cone have the dual block update only' if (int(getattr(model, 'cone', 0)) == 1 and int(getattr(model, 'npair', 0)) > 0) else
                   'some ground contact has friction / sqrt(impratio) < 1e-3 (rows an fp32 primal iteration cannot resolve)')
            warnings.warn(f'solver={self.solver_requested!r} was requested, but {why}: the model is solved on the dual problem instead - {self.solver_effective} to the '
                          f'solver tolerance, up to {self.solver_budget} sweeps per step (same minimiser; include/fmj.h: fmj_solver_info)', stacklevel=2)
        self.links_body = np.arange(1, m.nbody, dtype=np.int32)
        self.joints_jnt = np.nonzero(m.jnt_type != JNT_FREE)[0].astype(np.int32)
        self.reset()

    def __del__(self):
        ctx = getattr(self, '_ctx', None)
        if ctx:
            self._lib.fmj_destroy(ctx)
            self._ctx = None

    # ---- marshalling ----------------------------------------------------------------------------
    def _cdata(self, ctrl: Optional[torch.Tensor] = None, use_xfrc: bool = True) -> _lib.CData:
        d = self.data
        cached = getattr(d, '_cdata_cache', None)
        if cached is not None:                  # a private copy: callers edit single fields (ctrl tapes)
            c = _lib.CData.from_buffer_copy(cached)
            if ctrl is not None:
                c.ctrl = ctrl.data_ptr()
            if not use_xfrc:
                c.xfrc_applied = None
            return c
        c = self._cdata_build()
        object.__setattr__(d, '_cdata_cache', _lib.CData.from_buffer_copy(c))
        return self._cdata(ctrl, use_xfrc)

    def _cdata_build(self) -> _lib.CData:
        d = self.data
        ctrl, use_xfrc = None, True
        c = _lib.CData()
        c.qpos, c.qvel = d.qpos.data_ptr(), d.qvel.data_ptr()
        c.ctrl = (d.ctrl if ctrl is None else ctrl).data_ptr()
        c.qpos_spring = d.qpos_spring.data_ptr()
        c.xfrc_applied = d.xfrc_applied.data_ptr() if use_xfrc else None
        c.xpos, c.xquat, c.xipos = d.xpos.data_ptr(), d.xquat.data_ptr(), d.xipos.data_ptr()
        c.sensordata, c.qacc, c.time, c.status = (d.sensordata.data_ptr(), d.qacc.data_ptr(), d.time.data_ptr(),
                                                  d.status.data_ptr())
        c.qacc_warmstart, c.contact, c.ncon = d.qacc_warmstart.data_ptr(), d.contact.data_ptr(), d.ncon.data_ptr()
        return c

    # ---- dm_control Physics surface -----------------------------------------------------------------
    def reset(self, keyframe_id: int = 0):
        """physics.reset(keyframe_id=0) (reference task.py:137): keyframe state, then mj_forward with
        actuation disabled."""
        m, d = self.model, self.data
        d.qpos[:] = torch.as_tensor(m.key_qpos, dtype=torch.float32)
        d.qvel[:] = torch.as_tensor(m.key_qvel, dtype=torch.float32)
        d.ctrl.zero_(); d.xfrc_applied.zero_(); d.time.zero_(); d.status.zero_(); d.qacc.zero_()
        d.sensordata.zero_(); d.qacc_warmstart.zero_(); d.contact.zero_(); d.ncon.zero_()
        self.forward(disable_actuation=True)

    def forward(self, disable_actuation: bool = False):
        c = self._cdata()
        _lib.check(self._lib.fmj_forward(self._ctx, ctypes.byref(c), int(disable_actuation), _stream_ptr(self.device)))

    def step(self, nstep: int = 1, ctrl_tape: Optional[torch.Tensor] = None):
        """mujoco.mj_step x nstep for every env (reference simulation.py:156 via Physics.step).
        ``ctrl_tape`` [nstep, n_envs, nu] supplies a different ctrl row per step."""
        if ctrl_tape is not None:
            assert ctrl_tape.shape == (nstep, self.n_envs, self.model.nu) and ctrl_tape.is_contiguous()
            c = self._cdata(ctrl=ctrl_tape)
            stride = self.n_envs*self.model.nu
        else:
            c = self._cdata()
            stride = 0
        _lib.check(self._lib.fmj_step(self._ctx, ctypes.byref(c), int(nstep), stride, _stream_ptr(self.device)))

    def step_debug(self, want_pgs: bool = True):
        """One mj_step through ``fmj_step_debug``: returns ``(efc_rows [n_envs, maxefc, 8], pgs_improvement [n_envs,
        solver_iterations] or None)`` next to the usual state update (tests of the constraint solve; not the product path)."""
        me, mc, it = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _lib.check(self._lib.fmj_constraint_info(self._ctx, ctypes.byref(me), ctypes.byref(mc), ctypes.byref(it)))
        rows = torch.zeros(self.n_envs, max(me.value, 1), 8, device=self.device)
        imp = torch.full((self.n_envs, max(it.value, 1)), float('nan'), device=self.device) if want_pgs else None
        c = self._cdata()
        _lib.check(self._lib.fmj_step_debug(self._ctx, ctypes.byref(c), _ptr(rows), _ptr(imp), _stream_ptr(self.device)))
        return rows, imp

    # ---- checkpoint / resume ---------------------------------------------------------
    STATE_FIELDS = ('qpos', 'qvel', 'ctrl', 'qpos_spring', 'xfrc_applied', 'xpos', 'xquat', 'xipos', 'sensordata', 'qacc',
                    'time', 'status', 'qacc_warmstart', 'contact', 'ncon')

    def get_state(self):
        """Every mjData field as a host array: the integrated state (qpos, qvel, qpos_spring, time, status), the solver's warm
        start, and the derived fields of the last step (poses, sensordata, contacts) that the NEXT step's row readout reports
        (mj_step lag).  ``set_state`` of this dict on a context of the same model and batch resumes bitwise."""
        return {k: getattr(self.data, k).detach().cpu().numpy().copy() for k in self.STATE_FIELDS}

    def set_state(self, state):
        for k in self.STATE_FIELDS:
            t = getattr(self.data, k)
            a = np.asarray(state[k])
            if tuple(a.shape) != tuple(t.shape):
                raise ValueError(f'state field {k}: shape {a.shape}, this context holds {tuple(t.shape)}')
            t.copy_(torch.as_tensor(a, dtype=t.dtype))

    def check_invalid_state(self):
        """Raise PhysicsError if any env reported a bad-state warning (lazy, one sync)."""
        bad = torch.nonzero(self.data.status).flatten()
        if bad.numel():
            e = int(bad[0])
            raise PhysicsError(f'bad simulation state in {bad.numel()} env(s); first env {e} '
                               f'status bits {int(self.data.status[e])}')

    def timestep(self):
        return self.model.timestep

    # ---- operators the API layer calls ---------------------------------------------------------------
    def set_readout_maps(self, links_body, joints_jnt):
        self.links_body = np.ascontiguousarray(links_body, np.int32)
        self.joints_jnt = np.ascontiguousarray(joints_jnt, np.int32)
        I = ctypes.POINTER(ctypes.c_int32)
        _lib.check(self._lib.fmj_set_readout_maps(self._ctx, len(self.links_body), self.links_body.ctypes.data_as(I),
                                                  len(self.joints_jnt), self.joints_jnt.ctypes.data_as(I)))

    def set_swimming(self, n_xfrc_rows, links_index, xfrc_index, body_index, coefficients, masses, heights, densities):
        """``n_xfrc_rows`` = len(data.sensors.xfrc.names): the per-env row count of the xfrc tensor."""
        I, D = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double)
        a = [np.ascontiguousarray(x, np.int32) for x in (links_index, xfrc_index, body_index)]
        b = [np.ascontiguousarray(x, np.float64) for x in (coefficients, masses, heights, densities)]
        _lib.check(self._lib.fmj_set_swimming(self._ctx, len(a[0]), int(n_xfrc_rows), *[x.ctypes.data_as(I) for x in a],
                                              *[x.ctypes.data_as(D) for x in b]))

    def set_actuator_forcerange(self, forcelimited, forcerange):
        """Rewrite ``model.actuator_forcelimited`` / ``actuator_forcerange`` and refresh the device tables (the
        run-time edit of reference task.py:279-286)."""
        m = self.model
        m.actuator_forcelimited = np.ascontiguousarray(forcelimited, np.int32).reshape(m.nu)
        m.actuator_forcerange = np.ascontiguousarray(forcerange, np.float64).reshape(m.nu, 2)
        _lib.check(self._lib.fmj_set_actuator_forcerange(
            self._ctx, m.nu, m.actuator_forcelimited.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
            m.actuator_forcerange.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))

    def set_contact_maps(self, n_rows, geom_sensor, pairs=()):
        """geompair2data (reference physics.py:360-382): geom -> contact sensor row for keys (geom, -1), plus explicit
        (geom1, geom2, row) pairs."""
        I = ctypes.POINTER(ctypes.c_int32)
        gs = np.ascontiguousarray(geom_sensor, np.int32)
        pr = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 3))
        _lib.check(self._lib.fmj_set_contact_maps(self._ctx, int(n_rows), gs.ctypes.data_as(I), len(pr),
                                                  pr.ctypes.data_as(I) if len(pr) else None))

    def kernel_info(self):
        lds, thr = ctypes.c_int32(), ctypes.c_int32()
        _lib.check(self._lib.fmj_kernel_info(self._ctx, ctypes.byref(lds), ctypes.byref(thr)))
        return dict(lds_bytes_per_env=lds.value, threads_per_env=thr.value)
