"""Option stand-ins exposing the attributes the reference consumes; farms_core's
option classes are not part of the reference tree.  Same ``kwargs.pop`` + ``assert not kwargs``
strictness as the reference (task.py:53-73)."""
from types import SimpleNamespace

from .units import SimulationUnitScaling


class SimulationOptions:
    """farms_core.simulation.options.SimulationOptions surface (reference simulation.py:52,63,76-79,86)."""

    def __init__(self, **kwargs):
        self.timestep = kwargs.pop('timestep', 1e-3)
        self.n_iterations = kwargs.pop('n_iterations', 1000)
        self.num_sub_steps = kwargs.pop('num_sub_steps', 1)
        self.units = kwargs.pop('units', SimulationUnitScaling())
        self.play = kwargs.pop('play', True)
        self.headless = kwargs.pop('headless', True)
        self.fast = kwargs.pop('fast', True)
        self.show_progress = kwargs.pop('show_progress', False)
        self.gravity = kwargs.pop('gravity', [0.0, 0.0, -9.81])
        self.integrator = kwargs.pop('integrator', 'Euler')
        self.cone = kwargs.pop('cone', 'pyramidal')
        self.solver = kwargs.pop('solver', 'PGS')
        self.n_solver_iters = kwargs.pop('n_solver_iters', 50)
        self.impratio = kwargs.pop('impratio', 1)
        # forwarded to the option block by the reference (mjcf.py:1366-1403); ccd / mpr settings drive MuJoCo's general convex
        # collider, which this path does not use (its narrow phase is analytic), so they are accepted and ignored
        self.noslip_iterations = kwargs.pop('noslip_iterations', 0)
        self.noslip_tolerance = kwargs.pop('noslip_tolerance', 1e-6)
        self.ccd_iterations = kwargs.pop('ccd_iterations', 1000)
        self.ccd_tolerance = kwargs.pop('ccd_tolerance', 1e-6)
        self.mpr_iterations = kwargs.pop('mpr_iterations', 1000)
        self.mpr_tolerance = kwargs.pop('mpr_tolerance', 1e-6)
        assert not kwargs, kwargs

    def save(self, path):
        import yaml
        with open(path, 'w', encoding='utf-8') as f:
            yaml.safe_dump({k: (v if not isinstance(v, SimulationUnitScaling) else
                                dict(meters=v.meters, seconds=v.seconds, kilograms=v.kilograms))
                            for k, v in vars(self).items()}, f)


class WaterOptions:
    """ArenaOptions.water (reference drag.pyx:338-350, mjcf.py:1208-1224)."""

    def __init__(self, **kwargs):
        self.height = kwargs.pop('height', 0.0)
        self.drag = kwargs.pop('drag', True)
        self.sph = kwargs.pop('sph', False)
        self.buoyancy = kwargs.pop('buoyancy', True)
        self.density = kwargs.pop('density', 1000.0)
        self.velocity = kwargs.pop('velocity', [0.0, 0.0, 0.0])
        self.viscosity = kwargs.pop('viscosity', 1.0)
        assert not kwargs, kwargs


class ArenaOptions:
    """farms_core ArenaOptions surface (reference mjcf.py:1195-1225): ``sdf`` (arena model: a plane or a heightmap),
    ``spawn.pose``, ``ground_height``, ``water``."""

    def __init__(self, **kwargs):
        self.water = kwargs.pop('water', WaterOptions())
        self.ground_height = kwargs.pop('ground_height', None)
        self.sdf = kwargs.pop('sdf', None)
        self.spawn = SimpleNamespace(pose=list(kwargs.pop('spawn_pose', [0, 0, 0, 0, 0, 0])))
        assert not kwargs, kwargs


class AnimatOptions:
    """The slice of farms_core AnimatOptions the hot path reads: ``morphology.links[*].{name, swimming,
    density, drag_coefficients}`` (reference drag.pyx:353-385) and ``control.motors`` (task.py:274-286)."""

    def __init__(self, name='animat', links=(), motors=(), joints=(), sdf=None, spawn_pose=(0, 0, 0, 0, 0, 0),
                 spawn_velocity=(0, 0, 0, 0, 0, 0), mujoco=None):
        self.name = name
        self.sdf = sdf
        self.spawn = SimpleNamespace(pose=list(spawn_pose), velocity=list(spawn_velocity))
        self.morphology = SimpleNamespace(links=list(links), joints=list(joints), self_collisions=[])
        motors = list(motors)
        self.control = SimpleNamespace(motors=motors, joints_names=lambda: [m_.joint_name for m_ in motors])
        self.mujoco = dict(mujoco or {})

    @staticmethod
    def link(name, swimming=False, density=1000.0, drag_coefficients=((0, 0, 0), (0, 0, 0)), friction=(0, 0, 0), height=None):
        return SimpleNamespace(name=name, swimming=swimming, density=density, drag_coefficients=drag_coefficients,
                               friction=list(friction), height=height)

    @staticmethod
    def joint(name, initial=(0.0, 0.0), stiffness=0.0, damping=0.0, extras=None):
        return SimpleNamespace(name=name, initial=list(initial), stiffness=stiffness, damping=damping, extras=extras or {})

    @staticmethod
    def motor(joint_name, control_types=('position',), gains=(0.0, 0.0), limits_torque=None, passive=None):
        return SimpleNamespace(joint_name=joint_name, control_types=list(control_types), gains=list(gains),
                               limits_torque=limits_torque,
                               passive=passive or SimpleNamespace(is_passive=False, stiffness_coefficient=0.0, damping_coefficient=0.0))

    @classmethod
    def from_model(cls, model):
        links = []
        swim = {s['name']: s for s in model.swimming}
        for b in range(1, model.nbody):
            n = model.body_names[b]
            s = swim.get(n)
            links.append(SimpleNamespace(name=n, swimming=s is not None,
                                         density=s['density'] if s else 1000.0,
                                         drag_coefficients=s['drag_coefficients'] if s else [[0]*3, [0]*3],
                                         height=s['height'] if s else None, friction=[0, 0, 0]))
        return cls(name=model.name, links=links)

    def save(self, path):
        import yaml
        with open(path, 'w', encoding='utf-8') as f:
            yaml.safe_dump(dict(name=self.name, links=[l.name for l in self.morphology.links]), f)
