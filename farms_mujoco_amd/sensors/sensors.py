"""Contact sensors - HIP counterpart of reference farms_mujoco/sensors/sensors.pyx (contacts part)."""
import ctypes

import torch

from .. import _lib


def cycontacts2data(physics, iteration, data, geompair2data, meters, newtons):
    """Contacts to data (reference sensors.pyx:140-190) for every env -> C-ABI fmj_contacts2data.

    ``data`` is ``AnimatData.sensors.contacts``; ``geompair2data`` is the dict built by
    ``get_physics2data_maps`` and already uploaded with ``physics.set_contact_maps`` (kept for signature
    compatibility).  The row of ring index ``iteration`` is overwritten (the reference accumulates with ``+=``
    into a fresh row)."""
    rows = _lib.CRows()
    rows.contacts = data.array[iteration].data_ptr()
    u = _lib.CUnits(meters, newtons, 1.0, 1.0, 1.0, 1.0)
    c = physics._cdata()
    _lib.check(physics._lib.fmj_contacts2data(physics._ctx, ctypes.byref(c), ctypes.byref(rows), ctypes.byref(u),
                                              ctypes.c_void_p(torch.cuda.current_stream(physics.device).cuda_stream)))


def cymusclesensors2data(*args, **kwargs):
    """Muscle sensors (reference sensors.pyx:193-297) need farms_muscle, absent from the reference tree: out of scope."""
    raise NotImplementedError('muscle sensors are out of scope (farms_muscle is not part of the reference tree)')
