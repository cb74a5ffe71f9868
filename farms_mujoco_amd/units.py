"""Unit scaling stand-in for farms_core.units.SimulationUnitScaling (absent from the reference tree;
attributes consumed at reference physics.py:428-524, mjcf.py:169,567,582, task.py:331, drag.pyx:342-344)."""


class SimulationUnitScaling:
    def __init__(self, meters: float = 1.0, seconds: float = 1.0, kilograms: float = 1.0):
        self.meters = float(meters)
        self.seconds = float(seconds)
        self.kilograms = float(kilograms)

    @property
    def hertz(self): return 1.0/self.seconds
    @property
    def velocity(self): return self.meters/self.seconds
    @property
    def angular_velocity(self): return 1.0/self.seconds
    @property
    def acceleration(self): return self.velocity/self.seconds
    @property
    def newtons(self): return self.kilograms*self.acceleration
    @property
    def torques(self): return self.newtons*self.meters
    @property
    def inertia(self): return self.kilograms*self.meters**2
    @property
    def angular_stiffness(self): return self.torques
    @property
    def angular_damping(self): return self.torques/self.angular_velocity

    def as_c(self):
        from ._lib import CUnits
        return CUnits(self.meters, self.newtons, self.torques, self.velocity, self.angular_velocity, self.kilograms)

    def as_array(self):
        """(meters, newtons, torques, velocity, angular_velocity) — order used by the oracle wrapper."""
        return (self.meters, self.newtons, self.torques, self.velocity, self.angular_velocity)
