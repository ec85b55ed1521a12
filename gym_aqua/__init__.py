"""gym_aqua: same import name and env ids as the reference package (gym_aqua/__init__.py:4-41), backed
by the batched HIP implementation in aquaticgymenv_amd.

Six ids: Aqua{,Continuous}Env-v0 (no obstacles unless asked), -v1 (the default five obstacles),
-v2 (the six "difficult" obstacles).  They are registered with classic `gym` when it is importable (the reference's
target; the classes speak its reset()/step() protocol, not gymnasium's); `gym_aqua.make(id, **kwargs)` works either way.
"""
from aquaticgymenv_amd import presets

_VERSIONS = {"v0": {}, "v1": {"obstacles": True}, "v2": {"obstacles": presets.as_reference_list(presets.DIFFICULT6)}}
REGISTRY = {}
for _cls in ("AquaEnv", "AquaContinuousEnv"):
    for _ver, _kw in _VERSIONS.items():
        REGISTRY["%s-%s" % (_cls, _ver)] = ("gym_aqua.envs:%s" % _cls, dict(_kw))

difficult_obstacles = _VERSIONS["v2"]["obstacles"]


def _registered_ids(module):
    """ids the gym-like module's registry holds: gym <= 0.21 keeps `registry.env_specs` (a dict), later versions make
    `registry` itself the dict.  None when no registry can be read."""
    reg = getattr(module.envs.registration, "registry", None)
    specs = getattr(reg, "env_specs", reg)
    try:
        return set(specs.keys())
    except Exception:
        return None


def _register_with(module):
    """register the six ids with a gym-like module (anything with envs.registration.register(id=, entry_point=, kwargs=)),
    as the reference's gym_aqua/__init__.py:4-41 does.  An id the module's registry already holds (the package imported
    twice) keeps its first registration -- decided by looking the id up, not by reading an exception's text; every
    failure of register() itself is the caller's to see."""
    register = module.envs.registration.register
    have = _registered_ids(module)
    for env_id, (entry, kwargs) in REGISTRY.items():
        if have is not None and env_id in have:
            continue
        register(id=env_id, entry_point=entry, kwargs=kwargs)


# Classic `gym` only: the classes implement the reference's API (gym 0.17: reset() -> obs, step() -> 4-tuple, no
# seed/options keywords), which gymnasium's make() wrappers and env checker reject.
try:
    import gym as _gym_module
    import gym.envs.registration  # noqa: F401
except Exception:                  # gym is optional (absent from the build image): gym_aqua.make() works without it
    _gym_module = None
if _gym_module is not None:
    _register_with(_gym_module)


def make(env_id, **kwargs):
    """gym.make() look-alike for hosts without gym: make('AquaEnv-v1', num_envs=4096, obstacles=...)."""
    from gym_aqua import envs
    if env_id not in REGISTRY:
        raise KeyError("unknown environment id %r (known: %s)" % (env_id, ", ".join(sorted(REGISTRY))))
    entry, base = REGISTRY[env_id]
    merged = dict(base)
    merged.update(kwargs)
    return getattr(envs, entry.split(":")[1])(**merged)


from gym_aqua.envs.aqua import AquaEnv, AquaContinuousEnv  # noqa: E402,F401
