"""AnimatData stand-in (farms_core.model.data.AnimatData is not in the reference tree).

Ring-buffer sensor log the reference writes every iteration (reference task.py:62,158,208-216):
``data.sensors.{links,joints,xfrc,contacts}.array`` with shape ``[buffer_size, n_envs, n, width]`` —
the reference's ``[buffer_size, n, width]`` with an env axis inserted after the ring index, so
``array[index]`` is one contiguous slab written by one launch.  Column integers come from the HIP
library (``fmj_sc``), never hard-coded here.
"""
from types import SimpleNamespace

import torch

from . import _lib


class _SC:
    """Column convention namespace (farms_core ``sc``; reference physics.py:427-523)."""
    _names = {
        'link_com_position_x': ('LINK_COM_POS', 0), 'link_com_position_y': ('LINK_COM_POS', 1),
        'link_com_position_z': ('LINK_COM_POS', 2),
        'link_com_orientation_x': ('LINK_COM_QUAT', 0), 'link_com_orientation_w': ('LINK_COM_QUAT', 3),
        'link_urdf_position_x': ('LINK_URDF_POS', 0), 'link_urdf_position_z': ('LINK_URDF_POS', 2),
        'link_urdf_orientation_x': ('LINK_URDF_QUAT', 0), 'link_urdf_orientation_w': ('LINK_URDF_QUAT', 3),
        'link_com_velocity_lin_x': ('LINK_COM_LINVEL', 0), 'link_com_velocity_lin_z': ('LINK_COM_LINVEL', 2),
        'link_com_velocity_ang_x': ('LINK_COM_ANGVEL', 0), 'link_com_velocity_ang_z': ('LINK_COM_ANGVEL', 2),
        'link_size': ('LINK_SIZE', 0),
        'joint_position': ('JOINT_POSITION', 0), 'joint_velocity': ('JOINT_VELOCITY', 0),
        'joint_force_x': ('JOINT_FORCE', 0), 'joint_force_z': ('JOINT_FORCE', 2),
        'joint_torque_x': ('JOINT_TORQUE3', 0), 'joint_torque_z': ('JOINT_TORQUE3', 2),
        'joint_torque': ('JOINT_TORQUE', 0), 'joint_limit_force': ('JOINT_LIMIT_FORCE', 0),
        'joint_size': ('JOINT_SIZE', 0),
        'contact_reaction_x': ('CONTACT_REACTION', 0), 'contact_friction_x': ('CONTACT_FRICTION', 0),
        'contact_total_x': ('CONTACT_TOTAL', 0), 'contact_position_x': ('CONTACT_POSITION', 0),
        'contact_size': ('CONTACT_SIZE', 0),
        'xfrc_force_x': ('XFRC_FORCE', 0), 'xfrc_torque_x': ('XFRC_TORQUE', 0), 'xfrc_size': ('XFRC_SIZE', 0),
    }

    def __getattr__(self, name):
        try:
            base, off = self._names[name]
        except KeyError as e:
            raise AttributeError(name) from e
        return _lib.sc(base) + off


sc = _SC()


class SensorArray:
    def __init__(self, names, array):
        self.names = list(names)
        self.array = array

    @property
    def array(self):
        return self._array

    @array.setter
    def array(self, value):
        self._array = value
        self._base = None

    def row_ptr(self, index):
        """Device address of ring row ``index`` (= ``array[index].data_ptr()`` without building a view: the per-step host path)."""
        if self._base is None:
            a = self._array
            self._base = (a.data_ptr(), a.stride(0)*a.element_size())
        return self._base[0] + index*self._base[1]


class AnimatData:
    def __init__(self, timestep, buffer_size, n_envs, links, joints, xfrc=None, contacts=(), device='cuda:0'):
        self.timestep = timestep
        self.buffer_size = buffer_size
        self.n_envs = n_envs
        z = lambda n, w: torch.zeros(buffer_size, n_envs, max(n, 1), w, dtype=torch.float32, device=device)[:, :, :n]
        xfrc = list(links) if xfrc is None else list(xfrc)
        self.sensors = SimpleNamespace(
            links=SensorArray(links, z(len(links), sc.link_size)),
            joints=SensorArray(joints, z(len(joints), sc.joint_size)),
            xfrc=SensorArray(xfrc, z(len(xfrc), sc.xfrc_size)),
            contacts=SensorArray(contacts, z(len(contacts), sc.contact_size)),
            muscles=SensorArray([], z(0, 1)),
        )
        self.sensors.links.masses = None

    @classmethod
    def from_sensors_names(cls, timestep, buffer_size, links, joints, n_envs=1, device='cuda:0', **kwargs):
        """reference task.py:208-216 (``muscles`` accepted and ignored: out of scope)."""
        kwargs.pop('muscles', None)
        return cls(timestep, buffer_size, n_envs, links, joints, device=device, **kwargs)

    def to_file(self, path, iteration=None):
        """Post-hoc log (reference simulation.py:200-203: ``data.to_file('simulation.hdf5', iteration)``).
        ``*.hdf5`` / ``*.h5`` are written with h5py when it is importable (one dataset per sensor array under
        ``sensors/``, names as string attributes); otherwise, and for any other suffix, a compressed ``.npz`` with the
        same keys.  Arrays are [iteration, n_envs, n, width]."""
        import numpy as np
        n = self.buffer_size if iteration is None else min(iteration, self.buffer_size)
        arrays = {k: getattr(self.sensors, k).array[:n].cpu().numpy() for k in ('links', 'joints', 'xfrc', 'contacts')}
        names = {k: [str(x) for x in getattr(self.sensors, k).names] for k in arrays}
        if str(path).endswith(('.hdf5', '.h5')):
            try:
                import h5py
            except ImportError:
                path = str(path).rsplit('.', 1)[0] + '.npz'
            else:
                with h5py.File(path, 'w') as f:
                    f.attrs['timestep'] = self.timestep
                    g = f.create_group('sensors')
                    for k, a in arrays.items():
                        ds = g.create_dataset(k, data=a, compression='gzip')
                        ds.attrs['names'] = [str(x) for x in names[k]]
                return path
        np.savez_compressed(path, timestep=self.timestep, **arrays,
                            **{f'{k}_names': np.array([str(x) for x in v]) for k, v in names.items()})
        return path

    @classmethod
    def from_file(cls, path, device='cpu'):
        """Load a log written by ``to_file`` (either container)."""
        import numpy as np
        if str(path).endswith(('.hdf5', '.h5')):
            import h5py
            with h5py.File(path, 'r') as f:
                ts = float(f.attrs['timestep'])
                arrays = {k: np.asarray(f['sensors'][k]) for k in f['sensors']}
                names = {k: [n.decode() if isinstance(n, bytes) else str(n) for n in f['sensors'][k].attrs['names']] for k in f['sensors']}
        else:
            z = np.load(path, allow_pickle=False)
            ts = float(z['timestep'])
            arrays = {k: z[k] for k in ('links', 'joints', 'xfrc', 'contacts')}
            names = {k: [str(n) for n in z[f'{k}_names']] for k in arrays}
        n_it, n_envs = arrays['links'].shape[:2]
        contacts = [tuple(c.strip("()' ").replace("'", '').split(', ')) if c.startswith('(') else c for c in names['contacts']]
        out = cls(ts, n_it, n_envs, names['links'], names['joints'], xfrc=names['xfrc'], contacts=contacts, device=device)
        for k, a in arrays.items():
            getattr(out.sensors, k).array.copy_(torch.as_tensor(a))
        return out


if __name__ == '__main__':      # python -m farms_mujoco_amd.data in.npz out.hdf5  (on a machine that has h5py)
    import sys
    if len(sys.argv) != 3:
        sys.exit('usage: python -m farms_mujoco_amd.data <log written by to_file> <output .npz | .hdf5>')
    out = AnimatData.from_file(sys.argv[1]).to_file(sys.argv[2])
    print(out)
