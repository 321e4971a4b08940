"""Block / vehicle parameter spaces and their sampling rule.

Restates metadrive/component/pg_space.py: BoxSpace is declared namedtuple("BoxSpace", "max min")
(:14) so BoxSpace(750, 850) means max=750, min=850 and the sampler then draws uniform(low=850,
high=750) -- kept on purpose (hard part 5).  Sampling (BaseRunnable.sample_parameters,
base_class/base_runnable.py:82-96 + Dict.seed/Box.sample, pg_space.py:113-119,448-476): one integer
`random_seed = np_random.randint(0, 1e6)` is drawn from the OWNER's stream, every parameter's own
RandomState is re-seeded with that same integer, and each parameter draws one uniform.  Values are
rounded through float32 (Box dtype) -- or floored to int64 for DiscreteSpace.
"""
from collections import namedtuple

import numpy as np

from metadrive_ped_amd.rng import get_np_random

BoxSpace = namedtuple("BoxSpace", "max min")
DiscreteSpace = namedtuple("DiscreteSpace", "max min")
ConstantSpace = namedtuple("ConstantSpace", "value")


def sample_space(space, random_seed):
    """One draw of one parameter whose RandomState was seeded with `random_seed`."""
    rng = get_np_random(random_seed)
    if isinstance(space, ConstantSpace):
        low = np.float32(space.value)
        v = rng.uniform(low=np.full((1, ), low, np.float32), high=np.full((1, ), low, np.float32), size=(1, ))
        return float(np.float32(v[0]))
    if isinstance(space, BoxSpace):
        low = np.full((1, ), space.min, np.float32)
        high = np.full((1, ), space.max, np.float32)
        v = rng.uniform(low=low, high=high, size=(1, ))
        return float(np.float32(v[0]))
    if isinstance(space, DiscreteSpace):
        low = np.full((1, ), space.min, np.int64)
        high = np.full((1, ), space.max, np.int64) + 1
        v = rng.uniform(low=low, high=high, size=(1, ))
        return int(np.floor(v[0]))
    raise TypeError(space)


def sample_parameters(owner_rng, spaces):
    """BaseRunnable.sample_parameters: returns {name: value}; consumes ONE randint from owner_rng."""
    random_seed = int(owner_rng.randint(low=0, high=int(1e6)))
    return {k: sample_space(v, random_seed) for k, v in sorted(spaces.items())}


class Parameter:
    length = "length"
    radius = "radius"
    angle = "angle"
    dir = "dir"
    radius_inner = "inner_radius"
    radius_exit = "exit_radius"


class BlockParameterSpace:
    """pg_space.py:275-326 (subset built so far)"""
    INTERSECTION = {
        "radius": ConstantSpace(10),
        "change_lane_num": DiscreteSpace(min=0, max=1),
        "decrease_increase": DiscreteSpace(min=0, max=1),
    }
    T_INTERSECTION = {
        "radius": ConstantSpace(10),
        "t_type": DiscreteSpace(min=0, max=2),
        "change_lane_num": DiscreteSpace(min=0, max=1),
        "decrease_increase": DiscreteSpace(min=0, max=1),
    }
    RAMP_PARAMETER = {"length": BoxSpace(min=20, max=40)}
    ROUNDABOUT = {
        "exit_radius": BoxSpace(min=5, max=15),
        "inner_radius": BoxSpace(min=15, max=45),
        "angle": ConstantSpace(60),
    }
    BOTTLENECK = {
        Parameter.length: BoxSpace(min=20, max=50),
        "lane_num": DiscreteSpace(min=1, max=2),
        "bottle_len": ConstantSpace(20),
        "solid_center_line": ConstantSpace(0),
    }
    STRAIGHT = {Parameter.length: BoxSpace(min=40.0, max=80.0)}
    BIDIRECTION = {Parameter.length: BoxSpace(min=40.0, max=80.0)}
    PARKING_LOT = {"one_side_vehicle_number": DiscreteSpace(min=2, max=10), Parameter.radius: ConstantSpace(4),
                   Parameter.length: ConstantSpace(8)}
    CURVE = {
        Parameter.length: BoxSpace(min=40.0, max=80.0),
        Parameter.radius: BoxSpace(min=25.0, max=60.0),
        Parameter.angle: BoxSpace(min=45, max=135),
        Parameter.dir: DiscreteSpace(min=0, max=1),
    }


class VehicleParameterSpace:
    """pg_space.py:226-272"""
    STATIC_DEFAULT_VEHICLE = dict(
        wheel_friction=ConstantSpace(0.9), max_engine_force=ConstantSpace(800), max_brake_force=ConstantSpace(150),
        max_steering=ConstantSpace(40), max_speed_km_h=ConstantSpace(80))
    DEFAULT_VEHICLE = dict(
        wheel_friction=ConstantSpace(0.9), max_engine_force=BoxSpace(750, 850), max_brake_force=BoxSpace(80, 180),
        max_steering=ConstantSpace(40), max_speed_km_h=ConstantSpace(80))
    S_VEHICLE = dict(
        wheel_friction=ConstantSpace(0.9), max_engine_force=BoxSpace(350, 550), max_brake_force=BoxSpace(35, 80),
        max_steering=ConstantSpace(50), max_speed_km_h=ConstantSpace(80))
    M_VEHICLE = dict(
        wheel_friction=ConstantSpace(0.75), max_engine_force=BoxSpace(650, 850), max_brake_force=BoxSpace(60, 150),
        max_steering=ConstantSpace(45), max_speed_km_h=ConstantSpace(80))
    L_VEHICLE = dict(
        wheel_friction=ConstantSpace(0.8), max_engine_force=BoxSpace(450, 650), max_brake_force=BoxSpace(60, 120),
        max_steering=ConstantSpace(40), max_speed_km_h=ConstantSpace(80))
    XL_VEHICLE = dict(
        wheel_friction=ConstantSpace(0.7), max_engine_force=BoxSpace(500, 700), max_brake_force=BoxSpace(50, 100),
        max_steering=ConstantSpace(35), max_speed_km_h=ConstantSpace(80))


# (length, width, height, mass, front wheelbase, rear wheelbase, parameter space) per vehicle type
# component/vehicle/vehicle_type.py:8-165
VEHICLE_TYPES = {
    "default": dict(length=4.515, width=1.852, height=1.19, mass=1100, lf=1.05234, lr=1.4166,
                    space=VehicleParameterSpace.DEFAULT_VEHICLE),
    # VaryingDynamicsVehicle (vehicle_type.py:168-187): the default vehicle whose dynamics come from its config
    "varying_dynamics": dict(length=4.515, width=1.852, height=1.19, mass=1100, lf=1.05234, lr=1.4166,
                             space=VehicleParameterSpace.DEFAULT_VEHICLE),
    "static_default": dict(length=4.515, width=1.852, height=1.19, mass=1100, lf=1.05234, lr=1.4166,
                           space=VehicleParameterSpace.STATIC_DEFAULT_VEHICLE),
    "xl": dict(length=5.74, width=2.3, height=2.8, mass=1600, lf=1.726, lr=1.075, space=VehicleParameterSpace.XL_VEHICLE),
    "l": dict(length=4.87, width=2.046, height=1.85, mass=1300, lf=1.5301, lr=1.218261,
              space=VehicleParameterSpace.L_VEHICLE),
    "m": dict(length=4.6, width=1.85, height=1.37, mass=1200, lf=1.285, lr=1.203, space=VehicleParameterSpace.M_VEHICLE),
    "s": dict(length=4.3, width=1.70, height=1.70, mass=800, lf=1.385, lr=1.11, space=VehicleParameterSpace.S_VEHICLE),
}
