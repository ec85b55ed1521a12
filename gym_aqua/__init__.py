"""gym_aqua: same import name and env ids as the reference package (gym_aqua/__init__.py:4-41), backed
by the batched HIP implementation in aquaticgymenv_amd.

Six ids: Aqua{,Continuous}Env-v0 (no obstacles unless asked), -v1 (the default five obstacles),
-v2 (the six "difficult" obstacles).  They are registered with gym or gymnasium when one of them is
importable; `gym_aqua.make(id, **kwargs)` works either way.
"""
from aquaticgymenv_amd import presets

_VERSIONS = {"v0": {}, "v1": {"obstacles": True}, "v2": {"obstacles": presets.as_reference_list(presets.DIFFICULT6)}}
REGISTRY = {}
for _cls in ("AquaEnv", "AquaContinuousEnv"):
    for _ver, _kw in _VERSIONS.items():
        REGISTRY["%s-%s" % (_cls, _ver)] = ("gym_aqua.envs:%s" % _cls, dict(_kw))

difficult_obstacles = _VERSIONS["v2"]["obstacles"]


def _register_with(module):
    from_registration = getattr(module.envs.registration, "register")
    for env_id, (entry, kwargs) in REGISTRY.items():
        try:
            from_registration(id=env_id, entry_point=entry, kwargs=kwargs)
        except Exception:          # already registered (re-import) -> keep the first registration
            pass


for _name in ("gym", "gymnasium"):
    try:
        _mod = __import__(_name)
        __import__(_name + ".envs.registration")
        _register_with(_mod)
    except Exception:
        pass


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
