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
    # Contacts (reference physics.py:360-382): a sensor named (body, '') takes every geom of that body as key
    # (geom, -1); a sensor named (body1, body2) takes every ordered geom pair (geom of body1, geom of body2)
    contacts_pairs = list(sensor_data.contacts.names)
    body_names = m.body_names
    geompair2data = {
        (geom_id, -1): contacts_pairs.index((body_names[body_id], ''))
        for geom_id, body_id in enumerate(m.geom_bodyid[:m.ngeom])
        if (body_names[body_id], '') in contacts_pairs
    }
    if any(pair[1] != '' for pair in contacts_pairs if not isinstance(pair, str) and len(pair) == 2):
        geompair2data.update({
            (geom_id1, geom_id2): contacts_pairs.index((body_names[body_id1], body_names[body_id2]))
            for geom_id1, body_id1 in enumerate(m.geom_bodyid[:m.ngeom])
            for geom_id2, body_id2 in enumerate(m.geom_bodyid[:m.ngeom])
            if (body_names[body_id1], body_names[body_id2]) in contacts_pairs
        })
    sensor_maps['geompair2data'] = geompair2data
    geompair2data_values = geompair2data.values()
    for pair_i, pair in enumerate(contacts_pairs):
        assert not isinstance(pair, str) and len(pair) == 2, f'Contact "{pair}" should be a pair of strings'
        assert pair_i in geompair2data_values, f'Missing pair: {pair} ({body_names=})'
    if contacts_pairs:
        geom_sensor = -np.ones(max(m.ngeom, 1), np.int32)
        pairs = []
        for (g1, g2), row in geompair2data.items():
            if g2 < 0:
                geom_sensor[g1] = row
            else:
                pairs.append((g1, g2, row))
        physics.set_contact_maps(len(contacts_pairs), geom_sensor, pairs)
    return sensor_maps


def physics2data(physics, iteration, data, maps, units, links_only=False):
    """Sensors data collection for every env (reference physics.py:527-545) -> C-ABI fmj_physics2data."""
    rows = _lib.CRows()
    rows.links = data.sensors.links.array[iteration].data_ptr()
    rows.joints = data.sensors.joints.array[iteration].data_ptr()
    c = physics._cdata()
    u = units.as_c()
    _lib.check(physics._lib.fmj_physics2data(physics._ctx, ctypes.byref(c), ctypes.byref(rows), ctypes.byref(u),
                                             int(links_only),
                                             ctypes.c_void_p(torch.cuda.current_stream(physics.device).cuda_stream)))
    if not links_only and data.sensors.contacts.names:
        from ..sensors.sensors import cycontacts2data
        cycontacts2data(physics=physics, iteration=iteration, data=data.sensors.contacts,
                        geompair2data=maps['sensors'].get('geompair2data', {}), meters=units.meters, newtons=units.newtons)
