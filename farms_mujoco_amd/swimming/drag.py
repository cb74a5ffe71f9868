"""Drag forces — HIP counterpart of reference farms_mujoco/swimming/drag.pyx.

Same class and method names; ``SwimmingHandler.step`` launches the HIP drag operator for every
environment (C-ABI ``fmj_drag``) instead of looping over links in Cython (drag.pyx:389-411).
"""
import ctypes

import numpy as np
import torch

from .. import _lib


class WaterProperties:
    """Water properties (reference drag.pyx:271-306)."""

    def __init__(self, surface, density, velocity, viscosity):
        self._surface = float(surface)
        self._density = float(density)
        self._velocity = np.array(velocity, dtype=float)
        self._viscosity = float(viscosity)

    def surface(self, x=0.0, y=0.0): return self._surface
    def density(self, x=0.0, y=0.0, z=0.0): return self._density
    def velocity(self, x=0.0, y=0.0, z=0.0): return self._velocity
    def viscosity(self, x=0.0, y=0.0, z=0.0): return self._viscosity

    def set_velocity(self, vx, vy, vz):
        self._velocity[:] = (vx, vy, vz)

    def as_c(self, gravity=-9.81, use_buoyancy=True):
        w = _lib.CWater()
        w.surface, w.density, w.viscosity = self._surface, self._density, self._viscosity
        w.velocity = (ctypes.c_float*3)(*self._velocity)
        w.gravity = gravity          # hard-coded -9.81 at reference drag.pyx:409
        w.use_buoyancy = int(use_buoyancy)
        return w


class SwimmingHandler:
    """Swimming handler (reference drag.pyx:309-419)."""

    def __init__(self, data, animat_options, arena_options, units, physics):
        self.animat_options = animat_options
        self.links = data.sensors.links
        self.xfrc = data.sensors.xfrc
        water_options = arena_options.water
        self.drag = bool(water_options.drag)
        self.sph = getattr(water_options, 'sph', False)
        self.buoyancy = bool(water_options.buoyancy)
        self.units = units
        self.water = WaterProperties(surface=float(water_options.height), density=float(water_options.density),
                                     velocity=np.array(water_options.velocity, dtype=float),
                                     viscosity=float(water_options.viscosity))
        links = [link for link in animat_options.morphology.links if link.swimming]
        self.n_links = len(links)
        m = physics.model
        body_of = {n: i for i, n in enumerate(m.body_names)}
        self.body_index = np.array([body_of[l.name] for l in links], np.int32)
        self.masses = np.array([m.body_mass[b] for b in self.body_index], float)/units.kilograms   # drag.pyx:360-363
        self.heights = np.array([l.height for l in links], float)/units.meters                    # drag.pyx:364-372
        self.densities = np.array([l.density for l in links], float)
        self.xfrc_indices = np.array([self.xfrc.names.index(l.name) for l in links], np.int32)
        self.links_indices = np.array([self.links.names.index(l.name) for l in links], np.int32)
        self.links_coefficients = np.array([np.array(l.drag_coefficients) for l in links], float).reshape(-1, 2, 3)
        if self.sph:
            self.water._surface = 1e8                                                              # drag.pyx:386-387
        self.physics = physics
        physics.set_swimming(len(self.xfrc.names), self.links_indices, self.xfrc_indices, self.body_index,
                             self.links_coefficients, self.masses, self.heights, self.densities)

    def swim_dict(self):
        """Arrays in the layout the oracle wrapper takes (tests only)."""
        return dict(links_index=self.links_indices, xfrc_index=self.xfrc_indices, body_index=self.body_index,
                    coefficients=self.links_coefficients, masses=self.masses, heights=self.heights,
                    densities=self.densities)

    def step(self, iteration, write_xfrc_applied=True):
        """Swimming step for every env: links row ``iteration`` -> xfrc row ``iteration`` (+ the
        xfrc_applied glue the reference leaves to an external callback)."""
        if not (self.drag or self.sph) or not self.drag:
            return
        phys = self.physics
        rows = _lib.CRows()
        rows.links = self.links.array[iteration].data_ptr()
        rows.xfrc = self.xfrc.array[iteration].data_ptr()
        water = self.water.as_c(use_buoyancy=self.buoyancy)
        units = self.units.as_c()
        xa = phys.data.xfrc_applied.data_ptr() if write_xfrc_applied else None
        _lib.check(phys._lib.fmj_drag(phys._ctx, ctypes.byref(rows), ctypes.byref(water), ctypes.byref(units),
                                      ctypes.c_void_p(xa), ctypes.c_void_p(torch.cuda.current_stream(phys.device).cuda_stream)))

    def set_water_velocity(self, velocity):
        self.water.set_velocity(vx=velocity[0], vy=velocity[1], vz=velocity[2])


def drag_forces(iteration, data_links, links_index, data_xfrc, xfrc_index, coefficients, z3=None, z4=None, water=None,
                mass=0.0, height=1.0, density=1000.0, gravity=-9.81, use_buoyancy=True):
    """Drag swimming of ONE link (reference drag.pyx:152-268), for every env at once -> C-ABI ``fmj_drag_link``.

    ``data_links`` / ``data_xfrc`` are the ``links`` / ``xfrc`` sensor arrays of an :class:`AnimatData`
    (``[buffer_size, n_envs, n, width]`` device tensors), ``water`` a :class:`WaterProperties`; ``z3`` / ``z4`` are the
    reference's scratch vectors and are ignored.  Returns a bool tensor ``[n_envs]``: the reference's return value
    (False where the link is above the surface and its xfrc row was left untouched, drag.pyx:192-194)."""
    links = data_links.array if hasattr(data_links, 'array') else data_links
    xfrc = data_xfrc.array if hasattr(data_xfrc, 'array') else data_xfrc
    for name, t in (('links', links), ('xfrc', xfrc)):     # the kernel reads raw fp32 rows with a unit column stride
        if t.dtype != torch.float32 or t.stride(-1) != 1 or not t.is_cuda:
            raise TypeError(f'drag_forces: data_{name}.array must be a float32 device tensor with a unit last stride '
                            f'(got {t.dtype}, stride {tuple(t.stride())}, {t.device})')
    lrow = links[iteration, :, links_index]           # [n_envs, 20] view, env stride = links.stride(1)
    xrow = xfrc[iteration, :, xfrc_index]
    n_envs = lrow.shape[0]
    hydro = torch.zeros(n_envs, dtype=torch.int32, device=links.device)
    co = np.ascontiguousarray(coefficients, np.float64).reshape(2, 3)
    w = water.as_c(gravity=gravity, use_buoyancy=use_buoyancy)
    lib = _lib.load()
    _lib.check(lib.fmj_drag_link(n_envs, links.device.index or 0, lrow.data_ptr(), links.stride(1), xrow.data_ptr(),
                                 xfrc.stride(1), co.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), float(mass),
                                 float(height), float(density), ctypes.byref(w), hydro.data_ptr(),
                                 ctypes.c_void_p(torch.cuda.current_stream(links.device).cuda_stream)))
    return hydro.bool()
