"""Physics readout — counterpart of reference farms_mujoco/simulation/physics.py."""
import ctypes

import numpy as np
import torch

from .. import _lib
from ..model import JNT_FREE


def get_sensor_maps(physics, verbose=False):
    """Sensor-name -> sensordata index maps (reference physics.py:64-185), by the same name prefixes."""
    names = physics.model.sensor_names()
    widths = [3 if n.startswith(('framelinvel', 'frameangvel')) else 1 for n in names]
    adr = np.concatenate([[0], np.cumsum(widths)])
    sensors = ['framepos', 'framequat', 'framelinvel', 'frameangvel', 'jointpos', 'jointvel', 'jointlimitfrc',
               'force', 'torque', 'actuatorfrc_position', 'actuatorfrc_velocity', 'actuatorfrc_torque', 'touch']
    maps = {}
    for s in sensors:
        idx = [i for i, n in enumerate(names) if n.startswith(s)]
        maps[s] = {'names': [names[i] for i in idx],
                   'indices': np.array([np.arange(adr[i], adr[i + 1]) for i in idx])}
    return maps


def get_physics2data_maps(physics, sensor_data, sensor_maps):
    """Row maps from AnimatData names to bodies / joints (reference physics.py:188-393); uploads them to
    the HIP context (fmj_set_readout_maps)."""
    m = physics.model
    links_body = np.array([m.body_names.index(n) for n in sensor_data.links.names], np.int32)
    joints_jnt = np.array([m.joint_names.index(n) for n in sensor_data.joints.names], np.int32)
    assert all(m.jnt_type[j] != JNT_FREE for j in joints_jnt), 'joint rows must be hinge/slide joints'
    sensor_maps['xpos2data'] = sensor_maps['xquat2data'] = sensor_maps['xipos2data'] = links_body
    sensor_maps['qpos2data'] = m.jnt_qposadr[joints_jnt]
    sensor_maps['qvel2data'] = m.jnt_dofadr[joints_jnt]
    sensor_maps['datalinks2xfrc'] = links_body
    sensor_maps['data2xfrc'] = np.array([m.body_names.index(n) for n in sensor_data.xfrc.names], np.int32)
    physics.set_readout_maps(links_body, joints_jnt)
    # Contact sensors.  Contract of reference physics.py:360-382: ``geompair2data`` maps (geom, -1) to the row of the sensor named
    # (body of geom, '') and (geom1, geom2) to the row of the sensor named (body1, body2), every geom of a body counting for its body.
    # Built from a body -> geoms table: a sensor row fans out to its body's geoms (or the product of two bodies' geoms).
    sensor_names = list(sensor_data.contacts.names)      # (errors are AssertionError, as the reference's asserts raise)
    for name in sensor_names:
        if isinstance(name, str) or len(name) != 2:
            raise AssertionError(f'contact sensor {name!r}: expected a (body, body-or-empty) pair of strings')
    body_index = {name: b for b, name in enumerate(m.body_names)}
    geom_body = np.asarray(m.geom_bodyid[:m.ngeom], np.int64)
    geoms_of = {b: np.nonzero(geom_body == b)[0] for b in np.unique(geom_body)}
    geompair2data = {}
    geom_sensor = np.full(max(m.ngeom, 1), -1, np.int32)
    pair_rows = []
    if len(set(map(tuple, sensor_names))) != len(sensor_names):      # the reference's "Missing pair" assertion fires on a repeated name too
        raise AssertionError(f'contact sensor names must be unique: {sensor_names}')
    for row, (first, second) in enumerate(sensor_names):
        g1 = geoms_of.get(body_index.get(first, -1), ())
        if second == '':
            for g in g1:
                geompair2data[(int(g), -1)] = row
                geom_sensor[g] = row
            covered = len(g1) > 0
        else:
            g2 = geoms_of.get(body_index.get(second, -1), ())
            for a_ in g1:
                for b_ in g2:
                    geompair2data[(int(a_), int(b_))] = row
                    pair_rows.append((int(a_), int(b_), row))
            covered = len(g1) > 0 and len(g2) > 0
        if not covered:
            raise AssertionError(f'contact sensor {(first, second)!r} matches no collision geom (bodies: {m.body_names})')
    sensor_maps['geompair2data'] = geompair2data
    if sensor_names:
        physics.set_contact_maps(len(sensor_names), geom_sensor, pair_rows)
    return sensor_maps


def physics2data(physics, iteration, data, maps, units, links_only=False, swimming=None):
    """Sensors data collection for every env (reference physics.py:527-545) in ONE launch (C-ABI fmj_before_step): links and joints
    rows, the contact rows of ``cycontacts2data`` when the data has contact sensors, and - ``swimming`` = the SwimmingHandler of a
    swimming callback that comes first among the task's callbacks - its drag on the links row just written (reference
    task.py:176-182: sensors, then the callbacks in order)."""
    rows = _lib.CRows()
    rows.links = data.sensors.links.row_ptr(iteration)
    rows.joints = data.sensors.joints.row_ptr(iteration)
    flags = _lib.BEFORE_ROWS | (_lib.BEFORE_LINKS_ONLY if links_only else 0)
    if not links_only and data.sensors.contacts.names:
        rows.contacts = data.sensors.contacts.row_ptr(iteration)
        flags |= _lib.BEFORE_CONTACTS
    water, xa = None, None
    if swimming is not None and swimming.drag:
        rows.xfrc = swimming.xfrc.row_ptr(iteration)
        cwater = swimming.water.as_c(use_buoyancy=swimming.buoyancy)      # (kept alive until the call returns)
        water = ctypes.byref(cwater)
        xa = physics.data.xfrc_applied.data_ptr()
        flags |= _lib.BEFORE_DRAG
    c = physics._cdata()
    u = getattr(units, '_c_cache', None)
    if u is None:
        u = units._c_cache = units.as_c()       # (SimulationUnitScaling is a set of constants for the life of a task)
    _lib.check(physics._lib.fmj_before_step(physics._ctx, ctypes.byref(c), ctypes.byref(rows), water, ctypes.byref(u), flags,
                                            ctypes.c_void_p(xa), ctypes.c_void_p(torch.cuda.current_stream(physics.device).cuda_stream)))
