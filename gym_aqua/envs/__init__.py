from gym_aqua.envs.aqua import AquaEnv, AquaContinuousEnv  # noqa: F401
